// af_tower_bf16.hip — bf16 residual tower for gfx950 (C ABI: include/af_tower_bf16.h).
//
// One kernel, af_tower_conv<PROJ>, is a 3x3 convolution 128 -> 128 over 11x11 boards as an implicit GEMM
//   out[cout][pixel] = sum_{tap, cin} W[cout][cin][tap] * in[cin][pixel + tap offset]
// on v_mfma_f32_32x32x16_bf16 (M = 32 couts, N = 32 pixels, K = 16 cin of one tap), plus, for the second
// convolution of a block (PROJ), 8 more k-steps that fold the block's 1x1 projection of the block input in.
//
// Work split (weight-stationary, persistent): a workgroup = 4 waves = the 4 cout tiles of 32; each wave keeps
// ALL its weights — 32 couts x (9*128 [+128]) k = 72 [80] A fragments of 8 bf16 = 288 [320] registers — resident
// for the whole launch (one wave per SIMD, 512-register budget) and the workgroup walks over positions with
// stride gridDim.x.  A position's activations (36 KB in the C8 layout, include/af_tower_bf16.h) are copied
// global -> LDS by LDS-DMA (global_load_lds_dwordx4: contiguous, no staging registers), double buffered:
// position i+1 lands while position i multiplies.  In C8, a B fragment (32 pixels x 16 cin) is one
// ds_read_b128 per lane at lane_base + compile-time offset (pixel units are 16 bytes, consecutive pixels are
// consecutive units => conflict-free; for a row's missing left/right neighbours the edge lanes read an all-zero LDS
// region at the same bank offset instead), so the inner loop is ds_read_b128 + MFMA only.
// Epilogue: bias, ELU, round to bf16, two 16-byte stores per pixel tile (the MFMA rows are permuted at pack time so a
// lane holds 16 consecutive couts); the 32 lanes of a k half cover 512 contiguous bytes.
//
// Roofline: 2*128*128*9*121 = 35.7 MFLOP per position and convolution on 2.5 PFLOP/s dense bf16 MFMA;
// per position a workgroup reads 36 KB (+36 KB PROJ) and writes 31 KB of HBM: 8192 positions x 16 convolutions
// = 4.7 TFLOP and ~11 GB per tower pass, i.e. MFMA-bound (1.9 ms) over HBM (1.4 ms at 8 TB/s) by a small margin.
#include <hip/hip_runtime.h>

#include <type_traits>
#include <stdint.h>
#include <string.h>

#include <vector>

#include "af_tower_bf16.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define TW_HIP_OK(expr)                                      \
    do {                                                     \
        hipError_t err_ = (expr);                            \
        if (err_ != hipSuccess) return AF_TOWER_ERR_HIP;     \
    } while (0)

// Build-time variants (same experiments as af_conv_f16s.hip, r3; profiles/r3_18_tower_ab.txt): nt hint on the LDS-DMA loads: 5.29 -> 5.17 ms per
// 8192-position tower pass; lanes past the board read zeros: within noise (5.16-5.17 with both)
#ifndef AF_TOWER_NT_LOAD
#define AF_TOWER_NT_LOAD 1
#endif
#ifndef AF_TOWER_ZPAD
#define AF_TOWER_ZPAD 1
#endif
#ifndef AF_TOWER_VSLACK
#define AF_TOWER_VSLACK 88       // architectural VGPRs left to everything that is not a weight fragment or the B-fragment ring
#endif
#ifndef AF_TOWER_EPI_SCALAR
#define AF_TOWER_EPI_SCALAR 1    // af_tower_conv3: epilogue stages of two single-issue VALU instructions per MFMA gap (r5); 0 = r4's 2-vector stages
#endif
#ifndef AF_TOWER_ABL_NOLDS
#define AF_TOWER_ABL_NOLDS 0     // profiling build: af_tower_conv without its B-fragment reads (results wrong by design)
#endif

// LDS-DMA: 16 bytes per lane from global memory into LDS at (wave-uniform lds_dst) + lane*16; counted on vmcnt.
__device__ __forceinline__ void glds16(const void* gsrc, uint32_t lds_dst) {
    unsigned keep;
#if AF_TOWER_NT_LOAD
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
#else
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
#endif
}

// ELU as max(x, min(exp(x), 1) - 1): x > 0 ? x : exp(x) - 1 with one v_max_f32 instead of compare + select and the clamp riding on
// v_exp_f32 (r3_44).  exp(x) - 1 > x for x < 0 and the clamped exponential is exactly 1 for x >= 0; only for -3e-4 < x < 0 can the
// rounding of v_exp_f32 put exp(x) - 1 below x, and the max then returns x, within 5e-8 of the true value (see af_conv_f16s.hip)
__device__ __forceinline__ float elu1(float x) {
    const float e = __builtin_amdgcn_exp2f(x * 1.44269504088896341f);
    return fmaxf(x, fminf(fmaxf(e, 0.0f), 1.0f) - 1.0f);
}
// ... and two at a time on 2-vectors, which keeps the bias add in front and the "- 1" as v_pk_add_f32 (hipcc's SLP vectoriser pairs them
// in the compare / select form but not in the max form): 3 + v_exp_f32 instead of 4 + v_exp_f32 issue slots per element
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 elu2(f32x2 x) {
    const f32x2 t = x * 1.44269504088896341f;
    f32x2 e;
    e.x = fminf(fmaxf(__builtin_amdgcn_exp2f(t.x), 0.0f), 1.0f);
    e.y = fminf(fmaxf(__builtin_amdgcn_exp2f(t.y), 0.0f), 1.0f);
    const f32x2 em1 = e - 1.0f;
    return f32x2{fmaxf(x.x, em1.x), fmaxf(x.y, em1.y)};
}

// First position of a persistent workgroup (r4, as in af_conv_f16s.hip): consecutive workgroup ids sit on consecutive XCDs, so with
// pos = blockIdx.x the 32 workgroups of XCD k would walk the positions = k (mod 8) only; XCD k takes the contiguous positions
// 32k..32k+31 (+ gridDim.x per pass) instead.  A position's result does not depend on the workgroup that computes it.
#ifndef AF_TOWER_XCD_SWIZZLE
#define AF_TOWER_XCD_SWIZZLE 1
#endif
__device__ __forceinline__ int first_position() {
    if (AF_TOWER_XCD_SWIZZLE && (gridDim.x & 7) == 0) return (int)(blockIdx.x >> 3) + (int)(gridDim.x >> 3) * (int)(blockIdx.x & 7);
    return (int)blockIdx.x;
}

struct TowerArgs {
    const char* in;      // C8 bf16 [batch][16][PIX][8]: input of the 3x3 convolution
    const char* in2;     // PROJ: block input (1x1 projection), same layout
    const uint4* w;      // [4 waves][NS][64 lanes] A fragments (8 bf16)
    const float* bias;   // [128]
    char* out;           // C8 bf16
    int batch;
    int abl;             // profiling: bit 0 = stage only the first position, bit 1 = no stores, bit 2 = no ELU
    char* dump;          // af_tower_conv3: 4 KB the lanes past the board store to (an unconditional store keeps the epilogue out of a branch)
};

constexpr int kS = 11, kPIX = 144, kNPIX = kS * kS;                    // 13 rows x 11 pixels (+1): pixel n sits at unit n + 11
constexpr uint32_t kPlaneB = 16u * kPIX * 16u;                          // 36,864 bytes per position
constexpr int kRounds = kPlaneB / (256 * 16);                           // 9 LDS-DMA rounds of 256 lanes x 16 B
constexpr uint32_t kLds0 = 256u;                                        // planes start here: unit -1 of a plane stays inside LDS
constexpr uint32_t kZeroB = 33024u;                                     // all-zero LDS region the edge lanes read instead of a wrapped neighbour

#ifdef AF_TOWER_TIMING
// profiling build only (tools/probe_tower_timing.py): per wave and position, cycles of MFMA loop | wait for the next planes | barrier | epilogue
__device__ unsigned long long g_tower_cyc[2][256][4][5];
#define TT(v) const unsigned long long v = __builtin_readcyclecounter()
#else
#define TT(v)
#endif
// One convolution layer over this workgroup's positions.
template <bool PROJ, int DEPTH>
__device__ __forceinline__ void tower_layer(const TowerArgs& A, char* smem) {
    constexpr int NS = PROJ ? 80 : 72;
    const int lane = threadIdx.x & 63, kg = lane >> 5, nn = lane & 31;
    const uint32_t wv = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t lds0 = (uint32_t)(uintptr_t)smem + kLds0;

    auto stage_round = [&](const char* src, int pos, uint32_t off, int r) {   // 4 KB piece r of a position's plane -> LDS
        const char* gp = src + (size_t)pos * kPlaneB + threadIdx.x * 16u + r * 4096;
        const uint32_t ld = (uint32_t)__builtin_amdgcn_readfirstlane((int)(lds0 + off + wv * 1024u + r * 4096u));
        glds16(gp, ld);
    };
    auto stage = [&](const char* src, int pos, uint32_t off) {          // one position's plane -> LDS at byte offset off
#pragma unroll
        for (int r = 0; r < kRounds; ++r) stage_round(src, pos, off, r);
    };
    int pos = first_position();
    if (pos >= A.batch) return;
    constexpr uint32_t kZoff = kLds0 + (PROJ ? 3u : 2u) * kPlaneB;      // the zero region follows the planes
    for (uint32_t u = threadIdx.x; u < kZeroB / 16; u += 256) *reinterpret_cast<uint4*>(smem + kZoff + u * 16) = uint4{0, 0, 0, 0};
    stage(A.in, pos, 0u);
    if (PROJ) stage(A.in2, pos, 2u * kPlaneB);

    // resident weights and bias
    bf16x8 W[NS];
    {
        const uint4* wp = A.w + ((size_t)wv * NS * 64 + lane);
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const uint4 v = wp[s * 64];
            __builtin_memcpy(&W[s], &v, 16);
        }
    }
    float bias_r[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) bias_r[r] = A.bias[32 * wv + 16 * kg + r];   // MFMA rows are permuted at pack time: see pack_tower

    // the lane's pixel in each of the 4 pixel tiles: LDS byte base of tap (0,0) (one row up, one pixel left).  Rows are
    // stored back to back (no pad columns: consecutive lanes read consecutive 16-byte units, which is what keeps
    // ds_read_b128 conflict-free); for the left / right taps of the first / last pixel of a row the lane reads zeros
    // from a dedicated LDS region (a different base register, no per-read VALU).
    uint32_t lb[4], ob[4], zb[4];
    bool ok[4], edgeL[4], edgeR[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int n = 32 * j + nn;
        ok[j] = n < kNPIX;
        const int nc = ok[j] ? n : 0;
        const int x = nc % kS;
        edgeL[j] = x == 0; edgeR[j] = x == kS - 1;
        lb[j] = kLds0 + (uint32_t)(kg * kPIX + nc + kS - (kS + 1)) * 16u;
        ob[j] = (uint32_t)(nc + kS) * 16u;                              // output unit of the pixel
        zb[j] = kZoff + (lb[j] & 255u);                                 // same banks as the lane's own unit: stays conflict-free
    }
    // Pin the weight / bias loads' completion HERE: hipcc otherwise sinks its counted vmcnt waits to the first use of
    // each fragment inside the position loop, where they would also wait for the (to hipcc invisible) LDS-DMA of the
    // next position that is issued at the top of every iteration.
    constexpr int kNVfit = (256 - 4 * DEPTH - AF_TOWER_VSLACK) / 4, kNVmin = NS - (256 - 64) / 4;   // fragments kept in VGPRs
    constexpr int NV = kNVfit > kNVmin ? kNVfit : kNVmin;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        // 288 [320] weight registers do not fit the 256 architectural VGPRs: the upper half is pinned into accumulator registers,
        // which the MFMA reads as its A operand directly ("+v" for all of them made hipcc park them there anyway and copy each
        // fragment back with four v_accvgpr_read before use: one VALU instruction per MFMA in the hot loop)
        if (s < NV) asm volatile("" : "+v"(W[s]));
        else asm volatile("" : "+a"(W[s]));
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) asm volatile("" : "+v"(bias_r[r]));
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // first planes landed, zero region written
    __builtin_amdgcn_s_barrier();

#ifdef AF_TOWER_TIMING
    unsigned long long tacc[5] = {0, 0, 0, 0, 0};
#endif
    for (int it = 0; pos < A.batch; pos += gridDim.x, ++it) {
        const uint32_t gcur = (it & 1) ? kPlaneB : 0u, gnxt = kPlaneB - gcur;
        const int nxt = pos + (int)gridDim.x;
        const bool more = nxt < A.batch && !(A.abl & 1);
        uint32_t bC[4], bL[4], bR[4];                                   // read bases for the centre / left / right tap columns
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            bC[j] = (AF_TOWER_ZPAD && !ok[j]) ? zb[j] : gcur + lb[j];
            bL[j] = edgeL[j] ? zb[j] : bC[j];
            bR[j] = edgeR[j] ? zb[j] : bC[j];
        }
        TT(t0);
        f32x16 acc[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;
        // MFMA i of the position: pixel tile j = i % 4 of k-step i / 4; PROJ runs its 8 projection steps (block-input
        // plane) first.  B fragments ride a DEPTH-deep register ring: the ds_read_b128 of MFMA i + DEPTH is issued
        // right after MFMA i, so an LDS read has DEPTH MFMAs (32 cycles each) to land.
        constexpr int NM = NS * 4, NPJ = PROJ ? 32 : 0;
        auto rd = [&](int i) -> bf16x8 {
            const int j = i & 3;
            if (i < NPJ) {
                const int cc = i >> 2;
                return *reinterpret_cast<const bf16x8*>(smem + 2u * kPlaneB + lb[j] + (uint32_t)(2 * cc * kPIX + kS + 1) * 16u);
            }
            const int s_ = (i - NPJ) >> 2, t = s_ >> 3, cc = s_ & 7;
            const uint32_t base = t % 3 == 0 ? bL[j] : (t % 3 == 2 ? bR[j] : bC[j]);
            return *reinterpret_cast<const bf16x8*>(smem + base + (uint32_t)(2 * cc * kPIX + (t / 3) * kS + (t % 3)) * 16u);
        };
        bf16x8 ring[DEPTH];
#pragma unroll
        for (int i = 0; i < DEPTH; ++i) ring[i] = rd(i);
#pragma clang loop unroll(full)
        for (int i = 0; i < NM; ++i) {
            const int ws = i < NPJ ? 72 + (i >> 2) : (i - NPJ) >> 2;
            acc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W[ws], ring[i % DEPTH], acc[i & 3], 0, 0, 0);
            // (no run-time condition here: a profiling bit tested at this point cost a branch and four register moves per MFMA,
            // r2 -> r3_37; the "no LDS reads" ablation is the build-time macro AF_TOWER_ABL_NOLDS)
            if (i + DEPTH < NM && !AF_TOWER_ABL_NOLDS) ring[i % DEPTH] = rd(i + DEPTH);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // MFMA
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // DS read
            // the next position's planes are requested one 4 KB piece every 16 MFMAs (an LDS-DMA costs ~5 issue slots:
            // in one burst they would stall the matrix pipe for the whole burst)
            if (i % 16 == 0 && i / 16 < kRounds && more) stage_round(A.in, nxt, gnxt, i / 16);
            if (PROJ && i >= NPJ && i % 16 == 8 && (i - NPJ) / 16 < kRounds && more) stage_round(A.in2, nxt, 2u * kPlaneB, (i - NPJ) / 16);
            if (PROJ && i == NPJ - 1) {
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();                    // every wave is done with the projection plane
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        TT(t1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                // the next position's planes have landed ...
        TT(t2);
        __builtin_amdgcn_s_barrier();                                   // ... for every wave, and all are done with this one
        TT(t3);
        // epilogue: bias, ELU, bf16, store.  The weight rows are packed so that a lane's 16 accumulator rows are the 16
        // consecutive couts 32*wave + 16*kg + r: two whole 8-channel units = two 16-byte stores per pixel tile, and the
        // 32 lanes of a k half cover 512 contiguous bytes.
        char* const o = A.out + (size_t)pos * kPlaneB;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                bf16x8 v;
#pragma unroll
                for (int e = 0; e < 8; e += 2) {
                    const f32x2 y = elu2(f32x2{acc[j][8 * hf + e], acc[j][8 * hf + e + 1]} + f32x2{bias_r[8 * hf + e], bias_r[8 * hf + e + 1]});
                    v[e] = (__bf16)y.x; v[e + 1] = (__bf16)y.y;
                }
                if (ok[j] && !(A.abl & 2)) *reinterpret_cast<bf16x8*>(o + (uint32_t)((4 * wv + 2 * kg + hf) * kPIX) * 16u + ob[j]) = v;
            }
#ifdef AF_TOWER_TIMING
        { TT(t4); tacc[0] += t1 - t0; tacc[1] += t2 - t1; tacc[2] += t3 - t2; tacc[3] += t4 - t3; tacc[4] += 1; }
#endif
    }
#ifdef AF_TOWER_TIMING
    if (lane == 0 && blockIdx.x < 256) for (int q = 0; q < 5; ++q) g_tower_cyc[PROJ ? 1 : 0][blockIdx.x][wv][q] = tacc[q];
#endif
}

template <bool PROJ, int DEPTH>
__global__ __launch_bounds__(256, 1) void af_tower_conv(TowerArgs A) {
    extern __shared__ __attribute__((aligned(16))) char smem[];         // [256 B][g0][g1]([h])
    tower_layer<PROJ, DEPTH>(A, smem);
}

// (r4 A/B record, removed from the source in r6: the whole tower as ONE launch — af_tower_persist, a workgroup keeping its 32 positions
//  from layer to layer with no grid barrier — was bit-identical and bought nothing: 4.217 vs 4.198 ms per 8192-position pass,
//  profiles/r4_36; 98 % of a position's MFMA loop already runs at the MFMA issue rate and the late workgroups are the same in every layer)

// ---------------------------------------------------------------------------------------------------------------
// af_tower_conv3<PROJ, DEPTH> (r4): af_tower_conv with its epilogue taken off the critical path.  In af_tower_conv a position is
// 288 [320] MFMAs over FOUR live accumulator sets (pixel tile = MFMA index mod 4) followed by ~300 VALU instructions of bias / ELU /
// bf16 / stores during which the matrix pipe idles (one wave per SIMD: nobody else can use it) — 0.65 of 4.2 ms per tower pass.
// Here the position runs as two PHASES of two pixel tiles each (tiles 0,1 then tiles 2,3: 144 [160] MFMAs per phase, still alternating
// accumulators), and the finished pair's epilogue is issued in 48 small pieces (a third of a two-element chunk: 3-5 VALU) inside the OTHER pair's MFMA
// stream — pair A's under phase B of the same position, pair B's under phase A of the next one — where a wave's VALU work costs a
// fraction of its serial price (tools/probes/mfma_valu_coissue.hip: 4 VALU per MFMA add ~7 cycles to its 32).  Live accumulators:
// the accumulating pair + the pair in its epilogue = the same 64 registers as before.
// PROJ: the 16 projection MFMAs of a phase sit at the END of phase A and at the START of phase B, so the single-buffered block-input
// plane is needed for one 32-MFMA window in the middle of a position and the next position's copy of it streams in behind that
// window (9 pieces, one every 16 MFMAs) with 144 MFMAs to land.  Same MFMAs on the same operands in the same per-accumulator order
// as af_tower_conv (k-steps ascending per tile; PROJ: projection steps last for tiles 0,1 and first for tiles 2,3 — af_tower_conv
// runs them first for all four), so tiles 2,3 of the PROJ layers and every tile of the others are bit-identical to af_tower_conv.
template <bool PROJ, int DEPTH>
__global__ __launch_bounds__(256, 1) void af_tower_conv3(TowerArgs A) {
    extern __shared__ __attribute__((aligned(16))) char smem[];         // [256 B][g0][g1]([h])
    constexpr int NS = PROJ ? 80 : 72;
    const int lane = threadIdx.x & 63, kg = lane >> 5, nn = lane & 31;
    const uint32_t wv = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t lds0 = (uint32_t)(uintptr_t)smem + kLds0;

    auto stage_round = [&](const char* src, int pos, uint32_t off, int r) {   // 4 KB piece r of a position's plane -> LDS
        glds16(src + (size_t)pos * kPlaneB + threadIdx.x * 16u + r * 4096,
               (uint32_t)__builtin_amdgcn_readfirstlane((int)(lds0 + off + wv * 1024u + r * 4096u)));
    };
    int pos = first_position();
    if (pos >= A.batch) return;
    constexpr uint32_t kZoff = kLds0 + (PROJ ? 3u : 2u) * kPlaneB;      // the zero region follows the planes
    for (uint32_t u = threadIdx.x; u < kZeroB / 16; u += 256) *reinterpret_cast<uint4*>(smem + kZoff + u * 16) = uint4{0, 0, 0, 0};
#pragma unroll
    for (int r = 0; r < kRounds; ++r) stage_round(A.in, pos, 0u, r);
    if (PROJ) {
#pragma unroll
        for (int r = 0; r < kRounds; ++r) stage_round(A.in2, pos, 2u * kPlaneB, r);
    }
    bf16x8 W[NS];
    {
        const uint4* wp = A.w + ((size_t)wv * NS * 64 + lane);
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const uint4 v = wp[s * 64];
            __builtin_memcpy(&W[s], &v, 16);
        }
    }
    float bias_r[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) bias_r[r] = A.bias[32 * wv + 16 * kg + r];
    uint32_t lb[4], ob[4], zb[4];
    bool ok[4], edgeL[4], edgeR[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int n = 32 * j + nn;
        ok[j] = n < kNPIX;
        const int nc = ok[j] ? n : 0;
        const int x = nc % kS;
        edgeL[j] = x == 0; edgeR[j] = x == kS - 1;
        lb[j] = kLds0 + (uint32_t)(kg * kPIX + nc + kS - (kS + 1)) * 16u;
        ob[j] = (uint32_t)(nc + kS) * 16u;
        zb[j] = kZoff + (lb[j] & 255u);
    }
    constexpr int kNVfit = (256 - 4 * DEPTH - AF_TOWER_VSLACK) / 4, kNVmin = NS - (256 - 64) / 4;
    constexpr int NV = kNVfit > kNVmin ? kNVfit : kNVmin;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        if (s < NV) asm volatile("" : "+v"(W[s]));
        else asm volatile("" : "+a"(W[s]));
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) asm volatile("" : "+v"(bias_r[r]));
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    // MFMA m of a position -> (pixel tile j, projection step?, k-step index): phase A = m < NH (tiles 0,1), phase B = tiles 2,3
    constexpr int NM = NS * 4, NH = NM / 2, NP = PROJ ? 16 : 0;          // per phase: NH MFMAs of which NP are projection steps
    struct MI { int j, proj, s; };
    auto mi = [](int m) constexpr -> MI {
        constexpr int NM_ = (PROJ ? 80 : 72) * 4, NH_ = NM_ / 2, NP_ = PROJ ? 16 : 0;
        const int ph = m >= NH_, q = m - ph * NH_, j = 2 * ph + (q & 1);
        // phase A: 144 3x3 steps then NP projection steps; phase B: NP projection steps then 144 3x3 steps
        const bool proj = ph ? q < NP_ : q >= NH_ - NP_;
        const int s = proj ? (ph ? q : q - (NH_ - NP_)) >> 1 : (ph ? q - NP_ : q) >> 1;
        return MI{j, proj ? 1 : 0, s};
    };
    f32x16 acc[4];
    const f32x16 kZero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    char* o_prev = nullptr;

    // epilogue of pixel-tile pair jp0 as 16 chunks (tile jp0 + (c >> 3), elements 2*(c & 7), +1) of three stages of 3-5 VALU
    // instructions each — bias + scale | exp2 + clamp - 1 | max, bf16 (+ a 16-byte store after every fourth chunk) — so that one stage
    // fits the shadow of one MFMA; stage k of chunk c is issued behind MFMA 2*(3c + k) + 3 of the phase, fenced so that hipcc keeps it
    // there (unfenced it gathers four chunks in front of their store and reads a whole accumulator set in one burst)
    bf16x8 vst;
    f32x2 ex_[16], et_[16];
#if AF_TOWER_EPI_SCALAR
    // r5: the same epilogue as 16 chunks x 8 stages of TWO single-issue VALU instructions, one stage behind every MFMA of the other
    // pair's phase (128 of its 144 [160]).  r4's stages used 2-vector arithmetic (v_pk_add_f32 / v_pk_mul_f32): a packed fp32
    // instruction beside an MFMA costs ~+22-26 cycles of a 32-cycle gap (MI355X_MICROARCH.md, "price of one filler beside MFMAs"),
    // i.e. 48 stages x ~20 cycles per phase gave back what hiding the epilogue had won (254.5 vs 258.3 us per layer).  Plain
    // v_add / v_mul / v_exp / v_max / v_cvt_pk_bf16 fillers are hidden up to ~5 per gap.  The opaque asm("" : "+v") after each
    // result keeps hipcc's SLP vectoriser from re-packing the two elements of a chunk.
    float e0_[16], e1_[16], t0_[16], t1_[16];
    auto epi_stage = [&](int jp0, int c, int k, char* o) {
        const int j = jp0 + (c >> 3), q = c & 7, hf = q >> 2, e = 2 * (q & 3);
#define AF_OPAQUE(x) asm volatile("" : "+v"(x))
        if (k == 0) {
            float a0;
            asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(a0) : "a"(acc[j][8 * hf + e]));
            e0_[c] = a0 + bias_r[8 * hf + e]; AF_OPAQUE(e0_[c]);
        } else if (k == 1) {
            float a1;
            asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(a1) : "a"(acc[j][8 * hf + e + 1]));
            e1_[c] = a1 + bias_r[8 * hf + e + 1]; AF_OPAQUE(e1_[c]);
        } else if (k == 2) {
            float t = e0_[c] * 1.44269504088896341f; AF_OPAQUE(t);
            t0_[c] = fminf(fmaxf(__builtin_amdgcn_exp2f(t), 0.0f), 1.0f); AF_OPAQUE(t0_[c]);
        } else if (k == 3) {
            float t = e1_[c] * 1.44269504088896341f; AF_OPAQUE(t);
            t1_[c] = fminf(fmaxf(__builtin_amdgcn_exp2f(t), 0.0f), 1.0f); AF_OPAQUE(t1_[c]);
        } else if (k == 4) {
            // (v_max_f32 by hand: fmaxf() comes with a canonicalising v_max x, x, x of each input in IEEE mode — three instructions)
            const float m = t0_[c] - 1.0f;
            asm volatile("v_max_f32 %0, %1, %2" : "=v"(e0_[c]) : "v"(e0_[c]), "v"(m));
        } else if (k == 5) {
            const float m = t1_[c] - 1.0f;
            asm volatile("v_max_f32 %0, %1, %2" : "=v"(e1_[c]) : "v"(e1_[c]), "v"(m));
        } else if (k == 6) {
            vst[e] = (__bf16)e0_[c]; vst[e + 1] = (__bf16)e1_[c];
        } else if ((q & 3) == 3) {
            char* dst = o + (uint32_t)((4 * wv + 2 * kg + hf) * kPIX) * 16u + ob[j];
            if (j == 3) dst = ok[3] ? dst : A.dump + threadIdx.x * 16u;            // (only tile 3 has lanes past the board)
            *reinterpret_cast<bf16x8*>(dst) = vst;
        }
#undef AF_OPAQUE
    };
    (void)ex_; (void)et_;
#else
    auto epi_stage = [&](int jp0, int c, int k, char* o) {
        const int j = jp0 + (c >> 3), q = c & 7, hf = q >> 2, e = 2 * (q & 3);
        if (k == 0) {
            // (read where they are used: left to itself hipcc copies the whole finished accumulator set out of the accumulator
            // registers in one burst of 32 v_accvgpr_read at the top of the phase)
            float a0, a1;
            asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(a0) : "a"(acc[j][8 * hf + e]));
            asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(a1) : "a"(acc[j][8 * hf + e + 1]));
            ex_[c] = f32x2{a0, a1} + f32x2{bias_r[8 * hf + e], bias_r[8 * hf + e + 1]};
            et_[c] = ex_[c] * 1.44269504088896341f;
        } else if (k == 1) {
            f32x2 ee;
            ee.x = fminf(fmaxf(__builtin_amdgcn_exp2f(et_[c].x), 0.0f), 1.0f);
            ee.y = fminf(fmaxf(__builtin_amdgcn_exp2f(et_[c].y), 0.0f), 1.0f);
            et_[c] = ee - 1.0f;
        } else {
            vst[e] = (__bf16)fmaxf(ex_[c].x, et_[c].x); vst[e + 1] = (__bf16)fmaxf(ex_[c].y, et_[c].y);
            if ((q & 3) == 3) {
                char* dst = o + (uint32_t)((4 * wv + 2 * kg + hf) * kPIX) * 16u + ob[j];
                if (j == 3) dst = ok[3] ? dst : A.dump + threadIdx.x * 16u;        // (only tile 3 has lanes past the board)
                *reinterpret_cast<bf16x8*>(dst) = vst;
            }
        }
    };
#endif
    constexpr int kEpiK = AF_TOWER_EPI_SCALAR ? 8 : 3;                   // stages per chunk
    constexpr int kEpiStride = AF_TOWER_EPI_SCALAR ? 1 : 2;              // one stage every kEpiStride MFMAs, from the 4th of a phase on

    auto position = [&](auto first_tag, int it, int pos_) {
        constexpr bool FIRST = decltype(first_tag)::value;
        const uint32_t gcur = (it & 1) ? kPlaneB : 0u, gnxt = kPlaneB - gcur;
        const int nxt = pos_ + (int)gridDim.x;
        const bool more = nxt < A.batch && !(A.abl & 1);
        char* const o = A.out + (size_t)pos_ * kPlaneB;
        uint32_t bC[4], bL[4], bR[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            bC[j] = (AF_TOWER_ZPAD && !ok[j]) ? zb[j] : gcur + lb[j];
            bL[j] = edgeL[j] ? zb[j] : bC[j];
            bR[j] = edgeR[j] ? zb[j] : bC[j];
        }
        auto rd = [&](int m) -> bf16x8 {
            const MI x = mi(m);
            if (x.proj) return *reinterpret_cast<const bf16x8*>(smem + 2u * kPlaneB + lb[x.j] + (uint32_t)(2 * x.s * kPIX + kS + 1) * 16u);
            const int t = x.s >> 3, cc = x.s & 7;
            const uint32_t base = t % 3 == 0 ? bL[x.j] : (t % 3 == 2 ? bR[x.j] : bC[x.j]);
            return *reinterpret_cast<const bf16x8*>(smem + base + (uint32_t)(2 * cc * kPIX + (t / 3) * kS + (t % 3)) * 16u);
        };
        bf16x8 ring[DEPTH];
#pragma unroll
        for (int i = 0; i < DEPTH; ++i) ring[i] = rd(i);
#pragma clang loop unroll(full)
        for (int m = 0; m < NM; ++m) {
            const MI x = mi(m);
            const int ws = x.proj ? 72 + x.s : x.s;
            // the first MFMA of a tile starts from zero: the accumulator is not cleared while the pair is still in its epilogue
            // (srcC = the constant 0: the accumulator registers are not touched before this MFMA writes them)
            const bool fresh = (m < NH ? m : m - NH) < 2;
            if (fresh) acc[x.j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W[ws], ring[m % DEPTH], kZero16, 0, 0, 0);
            else acc[x.j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W[ws], ring[m % DEPTH], acc[x.j], 0, 0, 0);
            if (m + DEPTH < NM && !AF_TOWER_ABL_NOLDS) ring[m % DEPTH] = rd(m + DEPTH);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // MFMA
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // DS read
            // the other pair's epilogue, one chunk every 8 MFMAs from the 5th on (16 chunks inside the phase's first 132 MFMAs)
            {
                const int q = m < NH ? m : m - NH;
                if (q >= 3 && (q - 3) % kEpiStride == 0 && (q - 3) / kEpiStride < 16 * kEpiK && (m >= NH || !FIRST)) {
                    const int st_ = (q - 3) / kEpiStride;
                    __builtin_amdgcn_sched_barrier(0);
                    if (m >= NH) epi_stage(0, st_ / kEpiK, st_ % kEpiK, o);        // pair A of this position, under phase B
                    else epi_stage(2, st_ / kEpiK, st_ % kEpiK, o_prev);           // pair B of the previous position, under phase A
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            // the next position's planes: one 4 KB piece every 16 MFMAs from the start of the position
            if (m % 16 == 0 && m / 16 < kRounds && more) stage_round(A.in, nxt, gnxt, m / 16);
            if (PROJ) {
                constexpr int kAfter = NH + NP;                  // first MFMA after the projection window
                if (m == kAfter - 1) {
                    __builtin_amdgcn_sched_barrier(0);
                    __builtin_amdgcn_s_barrier();                // every wave is done with the block-input plane
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (m >= kAfter && (m - kAfter) % 16 == 8 && (m - kAfter) / 16 < kRounds && more)
                    stage_round(A.in2, nxt, 2u * kPlaneB, (m - kAfter) / 16);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                // the next position's planes have landed ...
        __builtin_amdgcn_s_barrier();                                   // ... for every wave, and all are done with this one
        o_prev = o;
    };

    position(std::true_type{}, 0, pos);
    pos += gridDim.x;
    for (int it = 1; pos < A.batch; pos += gridDim.x, ++it) position(std::false_type{}, it, pos);
    // the last position's second pair
#pragma unroll
    for (int c = 0; c < 16; ++c)
#pragma unroll
        for (int k = 0; k < kEpiK; ++k) epi_stage(2, c, k, o_prev);
}

// (r2 A/B record, removed from the source in r6: af_tower_conv2, the same convolution on af_conv_f16s.hip's slab-ring structure —
//  5.37 ms per tower pass against 4.2: its 32-channel slabs give a wave only 36 MFMAs between barriers)

// ---------------------------------------------------------------------------------------------------------------
// af_tower_stem: the 5x5 stem (3 -> 128, SAME) + bias + ELU, planes fp32 [B][3][11][11] -> C8 bf16, on the same MFMA.
// K = 3 planes x 5 rows x (5 taps padded to 8) = 15 groups of 8 = 8 k-steps; a B fragment (pixel n, group (cin, ky))
// is the 8-wide input window starting at column x-2 of row y+ky-2, pre-expanded into LDS once per position as
// 3 x 15 x 11 entries of 8 bf16 ("im2row"), so it is one aligned ds_read_b128.  Weights (8 A fragments) stay in
// registers; workgroups are persistent over positions.
struct StemArgs {
    const float* planes;
    const uint4* w;      // [4 waves][8][64] A fragments
    const float* bias;   // [128]
    char* out;           // C8 bf16
    int batch;
};

__global__ __launch_bounds__(256) void af_tower_stem_kernel(StemArgs A) {
    __shared__ __attribute__((aligned(16))) uint4 ent[3 * 15 * kS + 1];      // entry (cin, yy, x) = window x-2..x+5 of row yy-2
    const int lane = threadIdx.x & 63, kg = lane >> 5, nn = lane & 31;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    bf16x8 W[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        const uint4 v = A.w[((size_t)wv * 8 + s) * 64 + lane];
        __builtin_memcpy(&W[s], &v, 16);
    }
    float bias_r[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) bias_r[r] = A.bias[32 * wv + 16 * kg + r];
    uint32_t lb[4], ob[4];
    bool ok[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int n = 32 * j + nn;
        ok[j] = n < kNPIX;
        const int nc = ok[j] ? n : 0;
        lb[j] = (uint32_t)nc;                                            // entry index of (cin 0, yy = y, x)
        ob[j] = (uint32_t)(nc + kS) * 16u;
    }
    // r3 (as in af_stem_mfma_f16s): the planes go through a zero-bordered bf16 image in LDS (rows -2..12, columns -2..13) filled from
    // registers loaded one position ahead; the im2row entries are built from it instead of five scattered global loads each
    __shared__ __bf16 img[3 * 15 * 16];
    for (int i = threadIdx.x; i < 3 * 15 * 16; i += 256) img[i] = (__bf16)0.0f;
    const int t0 = threadIdx.x, t1 = threadIdx.x + 256;
    const bool h1 = t1 < 3 * kNPIX;
    auto img_at = [](int i) -> int { const int c = i / kNPIX, p = i - c * kNPIX, y = p / kS, x = p - y * kS; return (c * 15 + y + 2) * 16 + x + 2; };
    const int a0 = img_at(t0), a1 = img_at(h1 ? t1 : 0);
    float nx0 = 0.0f, nx1 = 0.0f;
    if ((int)blockIdx.x < A.batch) {
        const float* pl = A.planes + (size_t)blockIdx.x * 3 * kNPIX;
        nx0 = pl[t0];
        if (h1) nx1 = pl[t1];
    }
    for (int pos = blockIdx.x; pos < A.batch; pos += gridDim.x) {
        __syncthreads();                                                 // the previous position's reads are done
        img[a0] = (__bf16)nx0;
        if (h1) img[a1] = (__bf16)nx1;
        if (pos + (int)gridDim.x < A.batch) {
            const float* pl = A.planes + (size_t)(pos + gridDim.x) * 3 * kNPIX;
            nx0 = pl[t0];
            if (h1) nx1 = pl[t1];
        }
        __syncthreads();
        for (int en = threadIdx.x; en < 3 * 15 * kS; en += 256) {
            const int cin = en / (15 * kS), rem = en - cin * 15 * kS, yy = rem / kS, x = rem - yy * kS;
            const __bf16* src = img + (cin * 15 + yy) * 16 + x;
            const bf16x8 v = {src[0], src[1], src[2], src[3], src[4], (__bf16)0.0f, (__bf16)0.0f, (__bf16)0.0f};
            __builtin_memcpy(&ent[en], &v, 16);
        }
        __syncthreads();
        f32x16 acc[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            // group g = 2s + kg -> (cin, ky); g = 15 is the zero pad of K (its weights are zero: any entry will do)
            const int ga = 2 * s, gb = 2 * s + 1 < 15 ? 2 * s + 1 : 0;
            const uint32_t offa = (uint32_t)((ga / 5) * 15 + ga % 5) * kS, offb = (uint32_t)((gb / 5) * 15 + gb % 5) * kS;
            const uint32_t off = kg ? offb : offa;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                bf16x8 b;
                __builtin_memcpy(&b, &ent[lb[j] + off], 16);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W[s], b, acc[j], 0, 0, 0);
            }
        }
        char* const o = A.out + (size_t)pos * kPlaneB;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                bf16x8 v;
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = (__bf16)elu1(acc[j][8 * hf + e] + bias_r[8 * hf + e]);
                if (ok[j]) *reinterpret_cast<bf16x8*>(o + (uint32_t)((4 * wv + 2 * kg + hf) * kPIX) * 16u + ob[j]) = v;
            }
    }
}

// af_tower_heads_kernel: the two heads' 1x1 convolutions (128 -> 4 value, 128 -> 16 policy) + bias + ELU straight from the
// C8 tower output into the flattened NCHW rows the dense layers take: vin [B][4*121], pin [B][16*121] (bf16).
// 0.3 MMAC per position against 36 KB of input: HBM-bound, plain VALU (thread = pixel x half of the 20 couts, the
// weights broadcast from LDS).
struct HeadsArgs {
    const char* x;       // C8 bf16
    const float* w;      // [20][128] fp32 (bf16-rounded values): rows 0-3 value conv, 4-19 policy conv
    const float* b;      // [20]
    __bf16* vin;         // [B][484]
    __bf16* pin;         // [B][1936]
    int batch;
};

__global__ __launch_bounds__(256) void af_tower_heads_kernel(HeadsArgs A) {
    __shared__ __attribute__((aligned(16))) float wl[20 * 128];
    __shared__ float bl[20];
    for (int i = threadIdx.x; i < 20 * 128; i += 256) wl[i] = A.w[i];
    if (threadIdx.x < 20) bl[threadIdx.x] = A.b[threadIdx.x];
    __syncthreads();
    const int t = threadIdx.x;
    if (t >= 2 * kNPIX) return;
    const int grp = t / kNPIX, p = t - grp * kNPIX;
    for (int pos = blockIdx.x; pos < A.batch; pos += gridDim.x) {
        const char* xp = A.x + (size_t)pos * kPlaneB + (uint32_t)(p + kS) * 16u;
        float acc[10];
#pragma unroll
        for (int c = 0; c < 10; ++c) acc[c] = bl[10 * grp + c];
#pragma unroll 4
        for (int cb = 0; cb < 16; ++cb) {
            bf16x8 v;
            const uint4 raw = *reinterpret_cast<const uint4*>(xp + (size_t)cb * kPIX * 16);
            __builtin_memcpy(&v, &raw, 16);
            float f[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] = (float)v[e];
#pragma unroll
            for (int c = 0; c < 10; ++c) {
                const float4 w0 = *reinterpret_cast<const float4*>(&wl[(10 * grp + c) * 128 + cb * 8]);
                const float4 w1 = *reinterpret_cast<const float4*>(&wl[(10 * grp + c) * 128 + cb * 8 + 4]);
                acc[c] += f[0] * w0.x + f[1] * w0.y + f[2] * w0.z + f[3] * w0.w + f[4] * w1.x + f[5] * w1.y + f[6] * w1.z + f[7] * w1.w;
            }
        }
#pragma unroll
        for (int c = 0; c < 10; ++c) {
            const int co = 10 * grp + c;
            const __bf16 y = (__bf16)elu1(acc[c]);
            if (co < 4) A.vin[(size_t)pos * (4 * kNPIX) + co * kNPIX + p] = y;
            else A.pin[(size_t)pos * (16 * kNPIX) + (co - 4) * kNPIX + p] = y;
        }
    }
}

// af_tower_heads_mfma_kernel (r4): the same 1x1 convolutions on the matrix cores.  The VALU kernel above re-reads its 20 x 128 weights
// from LDS for every pixel (320 16-byte LDS reads per thread) and takes 198 us per 8192 positions against a 75-us HBM bound.  Here a
// wave = one 32-pixel tile of a position: M = 32 output rows (4 value + 16 policy channels, the rest zero weights), K = 128 channels =
// 8 k-steps whose B fragment is exactly one 16-byte C8 unit per lane (8 channels of the lane's pixel: straight from global memory,
// 512 contiguous bytes per half wave), A fragments (8 x 16 bytes per lane) resident.  The weight rows are permuted at pack time like
// the tower's, so a lane holds outputs 16 kg + r: bias, ELU, bf16, 2-byte stores that are contiguous over the 32 pixels of a tile.
struct HeadsMfmaArgs {
    const char* x;       // C8 bf16
    const uint4* a;      // [8 k-steps][64 lanes] A fragments (8 bf16)
    const float* b;      // [32]: bias of output o (0 for o >= 20)
    __bf16* vin;         // [B][484]
    __bf16* pin;         // [B][1936]
    int batch;
};

__global__ __launch_bounds__(256) void af_tower_heads_mfma_kernel(HeadsMfmaArgs A) {
    const int lane = threadIdx.x & 63, kg = lane >> 5, nn = lane & 31;
    const int j = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));       // pixel tile
    const int p = 32 * j + nn, pc = p < kNPIX ? p : 0;
    bf16x8 W[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) { const uint4 v = A.a[s * 64 + lane]; __builtin_memcpy(&W[s], &v, 16); }
    float bias_r[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) bias_r[r] = A.b[16 * kg + r];
    const uint32_t off = (uint32_t)(kg * kPIX + pc + kS) * 16u;                   // the lane's unit inside channel-group plane kg
    uint4 cur[8], nxt[8];
    int pos = blockIdx.x;
    if (pos >= A.batch) return;
#pragma unroll
    for (int s = 0; s < 8; ++s) cur[s] = *reinterpret_cast<const uint4*>(A.x + (size_t)pos * kPlaneB + off + (uint32_t)(2 * s * kPIX) * 16u);
    for (; pos < A.batch; pos += gridDim.x) {
        const int np = pos + (int)gridDim.x < A.batch ? pos + (int)gridDim.x : pos;
#pragma unroll
        for (int s = 0; s < 8; ++s) nxt[s] = *reinterpret_cast<const uint4*>(A.x + (size_t)np * kPlaneB + off + (uint32_t)(2 * s * kPIX) * 16u);
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            bf16x8 y;
            __builtin_memcpy(&y, &cur[s], 16);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W[s], y, acc, 0, 0, 0);
        }
        if (p < kNPIX) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int o = 16 * kg + r;                                            // (kg = 1: only o = 16..19 exist)
                if (o < 20) {
                    const __bf16 yv = (__bf16)elu1(acc[r] + bias_r[r]);
                    if (o < 4) A.vin[(size_t)pos * (4 * kNPIX) + o * kNPIX + p] = yv;
                    else A.pin[(size_t)pos * (16 * kNPIX) + (o - 4) * kNPIX + p] = yv;
                }
            }
        }
#pragma unroll
        for (int s = 0; s < 8; ++s) cur[s] = nxt[s];
    }
}

// af_tower_dense_kernel: the three dense layers and the softmax behind the heads' 1x1 convolutions (network.py:73-76,85-88)
// for 32 positions per workgroup on the same MFMA: D[out][position] = sum_k W^T[out][k] * in[position][k].
//   policy: 1936 -> 121 (+ bias) -> softmax: wave w owns outputs 32w..32w+31 (121 padded to 128 with zero weights), 121 k-steps
//   value : 484 -> 64 (+ bias, ELU) on waves 0 and 1 (31 k-steps, K padded to 496), then 64 -> 1 (+ bias), tanh(x/2)
// A fragments (weights) are pre-packed [k-step][out tile][lane][8] and stream from L2; a B fragment is 16 bytes of one
// position's input row per lane, read straight from global memory (consecutive k-steps reuse the same lines).
struct DenseArgs {
    const __bf16* vin;    // [B][484]
    const __bf16* pin;    // [B][1936]
    const uint4* wp;      // [121][4][64] policy A fragments
    const uint4* wv;      // [31][2][64] value fc1 A fragments
    const float* pb;      // [128] policy bias (padded with -inf-like for the 7 dead outputs)
    const float* vb1;     // [64]
    const float* vw2;     // [64]
    float vb2;
    float* policy;        // fp32 [B][121]
    float* value;         // fp32 [B]
    int batch;
};

__global__ __launch_bounds__(256) void af_tower_dense_kernel(DenseArgs A) {
    __shared__ float s_red[4][2][32];         // [wave][lane half][position]: partial max / sum / dot
    constexpr int kStageBuf = 32 * 528;
    __shared__ __attribute__((aligned(16))) char s_stage[2 * kStageBuf];      // policy rows of the 32 positions, 16 k-steps at a time
    const int lane = threadIdx.x & 63, kg = lane >> 5, nn = lane & 31;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    typedef uint32_t u32x4d __attribute__((ext_vector_type(4)));
    for (int p0 = blockIdx.x * 32; p0 < A.batch; p0 += gridDim.x * 32) {
        const int pos = p0 + nn < A.batch ? p0 + nn : A.batch - 1;      // dead lanes of a ragged tail recompute the last position
        // ---------------- policy ----------------
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
        // B operands through LDS (r4): read straight from global memory a B fragment is one 16-byte piece of a DIFFERENT position's row
        // per lane — 32 rows = 64 cache lines per load instruction; staged, 32 consecutive threads copy 512 contiguous bytes of one row
        // (chunks of 16 k-steps, double buffered, rows 528 bytes apart in LDS: an odd number of 16-byte units, conflict-free
        // ds_read_b128) and the fragment is one LDS read.  A operands (weights, L2) stay on the kRing-deep register ring.  Same k order.
        constexpr int kRing = 8, kChunks = 8;                                   // 7 x 16 + 9 k-steps
        u32x4d stg[4];
        auto stage_load = [&](int c) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int u = 256 * i + (int)threadIdx.x, row = u >> 5, col = u & 31;
                const int prw = p0 + row < A.batch ? p0 + row : A.batch - 1;
                stg[i] = u32x4d{0u, 0u, 0u, 0u};
                if (32 * c + col < 242) stg[i] = *reinterpret_cast<const u32x4d*>(reinterpret_cast<const char*>(A.pin) + (size_t)prw * 3872 + (size_t)c * 512 + col * 16);
            }
        };
        auto stage_store = [&](int buf) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int u = 256 * i + (int)threadIdx.x, row = u >> 5, col = u & 31;
                *reinterpret_cast<u32x4d*>(s_stage + buf * kStageBuf + row * 528 + col * 16) = stg[i];
            }
        };
        u32x4d ra[kRing];
#pragma unroll
        for (int q = 0; q < kRing; ++q) ra[q] = *reinterpret_cast<const u32x4d*>(A.wp + ((size_t)q * 4 + wv) * 64 + lane);
        stage_load(0);
        stage_store(0);
        __syncthreads();
        for (int c = 0; c < kChunks; ++c) {
            if (c + 1 < kChunks) stage_load(c + 1);
            const char* bb = s_stage + (c & 1) * kStageBuf + nn * 528 + kg * 16;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int k = 16 * c + q;
                if (k < 121) {
                    bf16x8 x, y;
                    __builtin_memcpy(&x, &ra[q % kRing], 16);
                    const u32x4d yb = *reinterpret_cast<const u32x4d*>(bb + q * 32);
                    __builtin_memcpy(&y, &yb, 16);
                    const int kn = k + kRing < 121 ? k + kRing : 120;          // (past the end: the last step again, never used)
                    ra[q % kRing] = *reinterpret_cast<const u32x4d*>(A.wp + ((size_t)kn * 4 + wv) * 64 + lane);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, acc, 0, 0, 0);
                }
            }
            if (c + 1 < kChunks) stage_store((c + 1) & 1);
            __syncthreads();
        }
        // a lane holds the outputs 32*wv + 16*kg + r of position nn (rows permuted at pack time)
        float mx = -3.0e38f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[r] += A.pb[32 * wv + 16 * kg + r]; mx = acc[r] > mx ? acc[r] : mx; }
        s_red[wv][kg][nn] = mx;
        __syncthreads();
        float gmx = -3.0e38f;
#pragma unroll
        for (int w = 0; w < 4; ++w)
#pragma unroll
            for (int h = 0; h < 2; ++h) { const float v = s_red[w][h][nn]; gmx = v > gmx ? v : gmx; }
        __syncthreads();
        float sum = 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[r] = __expf(acc[r] - gmx); sum += acc[r]; }
        s_red[wv][kg][nn] = sum;
        __syncthreads();
        float gs = 0.0f;
#pragma unroll
        for (int w = 0; w < 4; ++w)
#pragma unroll
            for (int h = 0; h < 2; ++h) gs += s_red[w][h][nn];
        __syncthreads();
        const float inv = 1.0f / gs;
        if (p0 + nn < A.batch) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int o = 32 * wv + 16 * kg + r;
                if (o < 121) A.policy[(size_t)pos * 121 + o] = acc[r] * inv;
            }
        }
        // ---------------- value ----------------
        float part = 0.0f;
        if (wv < 2) {
            f32x16 av;
#pragma unroll
            for (int r = 0; r < 16; ++r) av[r] = 0.0f;
            const __bf16* vrow = A.vin + (size_t)pos * 484 + 8 * kg;
            auto load_b = [&](int k) -> u32x4d {
                u32x4d b = {0u, 0u, 0u, 0u};
                if (16 * k + 8 * kg + 8 <= 484) b = *reinterpret_cast<const u32x4d*>(vrow + 16 * k);
                else if (16 * k + 8 * kg < 484) {                          // the last, partial 8-group: 484 = 30*16 + 4
                    __bf16 t8[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) t8[e] = 16 * k + 8 * kg + e < 484 ? vrow[16 * k + e] : (__bf16)0.0f;
                    __builtin_memcpy(&b, t8, 16);
                }
                return b;
            };
            u32x4d va[kRing], vb[kRing];                                   // the same register ring as the policy loop
#pragma unroll
            for (int q = 0; q < kRing; ++q) {
                va[q] = *reinterpret_cast<const u32x4d*>(A.wv + ((size_t)q * 2 + wv) * 64 + lane);
                vb[q] = load_b(q);
            }
            for (int k0 = 0; k0 < 31; k0 += kRing) {
#pragma unroll
                for (int q = 0; q < kRing; ++q) {
                    const int k = k0 + q;
                    if (k < 31) {
                        bf16x8 x, y;
                        __builtin_memcpy(&x, &va[q], 16);
                        __builtin_memcpy(&y, &vb[q], 16);
                        const int kn = k + kRing < 31 ? k + kRing : 30;
                        va[q] = *reinterpret_cast<const u32x4d*>(A.wv + ((size_t)kn * 2 + wv) * 64 + lane);
                        vb[q] = load_b(kn);
                        av = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, av, 0, 0, 0);
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int o = 32 * wv + 16 * kg + r;
                part += elu1(av[r] + A.vb1[o]) * A.vw2[o];
            }
        }
        s_red[wv][kg][nn] = part;
        __syncthreads();
        if (wv == 0 && kg == 0 && p0 + nn < A.batch) {
            const float z = s_red[0][0][nn] + s_red[0][1][nn] + s_red[1][0][nn] + s_red[1][1][nn] + A.vb2;
            A.value[pos] = tanhf(0.5f * z);
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------ host ------------------------------------------------------------------
struct af_tower {
    int S = 0, width = 0, blocks = 0, device = 0;
    std::vector<uint4*> w1, w2;             // A fragments of a block's two convolutions (pack_tower)
    std::vector<float*> b1, b2;
    std::vector<char> set;
    uint4* stem_w = nullptr;
    char* dump = nullptr;
    float* stem_b = nullptr;
    float* heads_w = nullptr;
    float* heads_b = nullptr;
    uint4* heads_a = nullptr;     // af_tower_heads_mfma_kernel: A fragments
    float* heads_b32 = nullptr;   // ... and its 32 biases
    uint4 *dense_wp = nullptr, *dense_wv = nullptr;     // af_tower_dense_kernel: packed policy / value fc1 weights
    float *dense_pb = nullptr, *dense_vb1 = nullptr, *dense_vw2 = nullptr;
    float dense_vb2 = 0.0f;
};

static uint16_t bf16_rne(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);   // NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

static float bf16_round(float f) {
    const uint32_t u = (uint32_t)bf16_rne(f) << 16;
    float r;
    memcpy(&r, &u, 4);
    return r;
}

// A fragments: [wave][s][lane][e] = W[cout = 32*wave + perm(lane&31)][cin = 16*cc + 8*(lane>>5) + e][tap t], s = 8*t + cc;
// s = 72 + cc: the 1x1 projection.
static std::vector<uint16_t> pack_tower(const float* w3, const float* w1x1) {
    const int NS = w1x1 ? 80 : 72;
    std::vector<uint16_t> out((size_t)4 * NS * 64 * 8);
    for (int wv = 0; wv < 4; ++wv)
        for (int s = 0; s < NS; ++s)
            for (int lane = 0; lane < 64; ++lane)
                for (int e = 0; e < 8; ++e) {
                    const int m = lane & 31;                                    // MFMA row -> cout: lane half kg = (m>>2)&1 ends up holding
                    const int co = 32 * wv + 16 * ((m >> 2) & 1) + 8 * (m >> 4) + 4 * ((m >> 3) & 1) + (m & 3);   // 16 consecutive couts in its 16 C rows
                    const int cc = s < 72 ? s % 8 : s - 72;
                    const int ci = 16 * cc + 8 * (lane >> 5) + e;
                    const float v = s < 72 ? w3[((size_t)co * 128 + ci) * 9 + s / 8] : w1x1[(size_t)co * 128 + ci];
                    out[(((size_t)wv * NS + s) * 64 + lane) * 8 + e] = bf16_rne(v);
                }
    return out;
}

template <class T>
static int upload(T** dst, const void* src, size_t bytes) {
    if (*dst) (void)hipFree(*dst);
    *dst = nullptr;
    TW_HIP_OK(hipMalloc(reinterpret_cast<void**>(dst), bytes));
    TW_HIP_OK(hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice));
    return AF_TOWER_OK;
}

static int g_depth = 0;     // B-fragment ring depth (A/B knob; 0 = per-kernel default)
static int g_abl = 0;       // profiling ablations (results wrong by design)
static int g_grid = 0;      // persistent workgroups (0 = one per CU)
static int g_heads = 1;     // heads' 1x1 convolutions: 1 = af_tower_heads_mfma_kernel (r4), 0 = the VALU kernel (A/B)
static int g_engine = 3;    // 3 (default, r4): af_tower_conv3 for a block's first convolution + af_tower_conv for its second (bit-identical to 0,
                            //    4.143 vs 4.167 ms per 8192-position tower pass); 0: af_tower_conv for both; 2: af_tower_conv3 for both (its
                            //    PROJ form is slower: 4.224 ms)

extern "C" {

int af_tower_tune(int32_t key, int32_t value) {
    if (key == 0) { g_depth = value; return AF_TOWER_OK; }
    if (key == 1) { g_grid = value; return AF_TOWER_OK; }
    if (key == 2) { g_abl = value; return AF_TOWER_OK; }
    if (key == 3) { if (value != 0 && value != 2 && value != 3) return AF_TOWER_ERR_ARG; g_engine = value; return AF_TOWER_OK; }
    if (key == 4) { g_heads = value; return AF_TOWER_OK; }
    return AF_TOWER_ERR_ARG;
}

const char* af_tower_strerror(int code) {
    switch (code) {
        case AF_TOWER_OK: return "ok";
        case AF_TOWER_ERR_ARG: return "bad argument";
        case AF_TOWER_ERR_HIP: return "HIP runtime error";
        case AF_TOWER_ERR_STATE: return "not every block has weights";
        default: return "unknown error";
    }
}

int af_tower_create(int32_t S, int32_t width, int32_t blocks, int32_t device, af_tower** out) {
    if (!out || S != kS || width != 128 || blocks < 1) return AF_TOWER_ERR_ARG;
    TW_HIP_OK(hipSetDevice(device));
    af_tower* t = new af_tower();
    t->S = S; t->width = width; t->blocks = blocks; t->device = device;
    t->w1.assign(blocks, nullptr); t->w2.assign(blocks, nullptr);
    t->b1.assign(blocks, nullptr); t->b2.assign(blocks, nullptr);
    t->set.assign(blocks, 0);
#define TW_ATTR(P, D) TW_HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(af_tower_conv<P, D>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024))
    TW_ATTR(false, 8); TW_ATTR(true, 8); TW_ATTR(false, 12); TW_ATTR(true, 12); TW_ATTR(false, 16); TW_ATTR(true, 16);
#undef TW_ATTR
    TW_HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(af_tower_conv3<false, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    TW_HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(af_tower_conv3<true, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    { void* q = nullptr; TW_HIP_OK(hipMalloc(&q, 4096)); t->dump = static_cast<char*>(q); }
    *out = t;
    return AF_TOWER_OK;
}

void af_tower_destroy(af_tower* t) {
    if (!t) return;
    (void)hipSetDevice(t->device);
    for (auto p : t->w1) if (p) (void)hipFree(p);
    for (auto p : t->w2) if (p) (void)hipFree(p);
    for (auto p : t->b1) if (p) (void)hipFree(p);
    for (auto p : t->b2) if (p) (void)hipFree(p);
    if (t->stem_w) (void)hipFree(t->stem_w);
    if (t->dump) (void)hipFree(t->dump);
    if (t->stem_b) (void)hipFree(t->stem_b);
    if (t->heads_w) (void)hipFree(t->heads_w);
    if (t->heads_b) (void)hipFree(t->heads_b);
    if (t->heads_a) (void)hipFree(t->heads_a);
    if (t->heads_b32) (void)hipFree(t->heads_b32);
    if (t->dense_wp) (void)hipFree(t->dense_wp);
    if (t->dense_wv) (void)hipFree(t->dense_wv);
    if (t->dense_pb) (void)hipFree(t->dense_pb);
    if (t->dense_vb1) (void)hipFree(t->dense_vb1);
    if (t->dense_vw2) (void)hipFree(t->dense_vw2);
    delete t;
}

int af_tower_set_block(af_tower* t, int32_t b, const float* c1_w, const float* c1_b, const float* c2_w, const float* c2_b,
                       const float* res_w, const float* res_b) {
    if (!t || b < 0 || b >= t->blocks || !c1_w || !c1_b || !c2_w || !c2_b || !res_w || !res_b) return AF_TOWER_ERR_ARG;
    TW_HIP_OK(hipSetDevice(t->device));
    const std::vector<uint16_t> p1 = pack_tower(c1_w, nullptr), p2 = pack_tower(c2_w, res_w);
    std::vector<float> bb(128);
    int rc = upload(&t->w1[b], p1.data(), p1.size() * 2);
    if (!rc) rc = upload(&t->w2[b], p2.data(), p2.size() * 2);
    if (!rc) rc = upload(&t->b1[b], c1_b, 128 * 4);
    for (int i = 0; i < 128; ++i) bb[i] = c2_b[i] + res_b[i];
    if (!rc) rc = upload(&t->b2[b], bb.data(), 128 * 4);
    if (!rc) t->set[b] = 1;
    return rc;
}


int af_tower_set_stem(af_tower* t, const float* w, const float* b) {
    if (!t || !w || !b) return AF_TOWER_ERR_ARG;
    TW_HIP_OK(hipSetDevice(t->device));
    std::vector<uint16_t> pk((size_t)4 * 8 * 64 * 8, 0);                 // [wave][s][lane][e]
    for (int wv = 0; wv < 4; ++wv)
        for (int s = 0; s < 8; ++s)
            for (int lane = 0; lane < 64; ++lane) {
                const int m = lane & 31, g = 2 * s + (lane >> 5);
                const int co = 32 * wv + 16 * ((m >> 2) & 1) + 8 * (m >> 4) + 4 * ((m >> 3) & 1) + (m & 3);
                if (g >= 15) continue;
                for (int e = 0; e < 5; ++e)                               // w OIHW [128][3][5][5]
                    pk[(((size_t)wv * 8 + s) * 64 + lane) * 8 + e] = bf16_rne(w[(((size_t)co * 3 + g / 5) * 5 + g % 5) * 5 + e]);
            }
    int rc = upload(&t->stem_w, pk.data(), pk.size() * 2);
    if (!rc) rc = upload(&t->stem_b, b, 128 * 4);
    return rc;
}

int af_tower_set_heads(af_tower* t, const float* vconv_w, const float* vconv_b, const float* pconv_w, const float* pconv_b) {
    if (!t || !vconv_w || !vconv_b || !pconv_w || !pconv_b) return AF_TOWER_ERR_ARG;
    TW_HIP_OK(hipSetDevice(t->device));
    std::vector<float> w(20 * 128), b(20);
    for (int c = 0; c < 20; ++c) {
        b[c] = c < 4 ? vconv_b[c] : pconv_b[c - 4];
        for (int k = 0; k < 128; ++k) w[c * 128 + k] = bf16_round(c < 4 ? vconv_w[c * 128 + k] : pconv_w[(c - 4) * 128 + k]);
    }
    int rc = upload(&t->heads_w, w.data(), w.size() * 4);
    if (!rc) rc = upload(&t->heads_b, b.data(), b.size() * 4);
    // A fragments of the MFMA kernel: [k-step][lane][e] = W[o = perm(lane & 31)][ch = 16 step + 8 (lane >> 5) + e] (same row permutation
    // as pack_tower: a lane's accumulator rows are 16 consecutive outputs)
    auto perm = [](int m) { return 16 * ((m >> 2) & 1) + 8 * (m >> 4) + 4 * ((m >> 3) & 1) + (m & 3); };
    std::vector<uint16_t> a((size_t)8 * 64 * 8, 0);
    for (int step = 0; step < 8; ++step)
        for (int lane = 0; lane < 64; ++lane)
            for (int e = 0; e < 8; ++e) {
                const int o = perm(lane & 31), ch = 16 * step + 8 * (lane >> 5) + e;
                if (o < 20) a[((size_t)step * 64 + lane) * 8 + e] = bf16_rne(w[o * 128 + ch]);
            }
    std::vector<float> b32(32, 0.0f);
    for (int o = 0; o < 20; ++o) b32[o] = b[o];
    if (!rc) rc = upload(&t->heads_a, a.data(), a.size() * 2);
    if (!rc) rc = upload(&t->heads_b32, b32.data(), b32.size() * 4);
    return rc;
}

// dense layers (network.py:73-76,85): vfc1 [484][64] + [64], vfc2 [64] + [1], pfc [1936][121] + [121], all [in][out] fp32
int af_tower_set_dense(af_tower* t, const float* vfc1_w, const float* vfc1_b, const float* vfc2_w, const float* vfc2_b,
                       const float* pfc_w, const float* pfc_b) {
    if (!t || !vfc1_w || !vfc1_b || !vfc2_w || !vfc2_b || !pfc_w || !pfc_b) return AF_TOWER_ERR_ARG;
    TW_HIP_OK(hipSetDevice(t->device));
    auto perm = [](int m) { return 16 * ((m >> 2) & 1) + 8 * (m >> 4) + 4 * ((m >> 3) & 1) + (m & 3); };
    std::vector<uint16_t> wp((size_t)121 * 4 * 64 * 8, 0), wv((size_t)31 * 2 * 64 * 8, 0);
    for (int k = 0; k < 121; ++k)
        for (int tile = 0; tile < 4; ++tile)
            for (int lane = 0; lane < 64; ++lane)
                for (int e = 0; e < 8; ++e) {
                    const int o = 32 * tile + perm(lane & 31), ki = 16 * k + 8 * (lane >> 5) + e;
                    if (o < 121) wp[((((size_t)k * 4 + tile) * 64) + lane) * 8 + e] = bf16_rne(pfc_w[(size_t)ki * 121 + o]);
                }
    for (int k = 0; k < 31; ++k)
        for (int tile = 0; tile < 2; ++tile)
            for (int lane = 0; lane < 64; ++lane)
                for (int e = 0; e < 8; ++e) {
                    const int o = 32 * tile + perm(lane & 31), ki = 16 * k + 8 * (lane >> 5) + e;
                    if (ki < 484) wv[((((size_t)k * 2 + tile) * 64) + lane) * 8 + e] = bf16_rne(vfc1_w[(size_t)ki * 64 + o]);
                }
    std::vector<float> pb(128, -1.0e30f), vb1(vfc1_b, vfc1_b + 64), vw2(64);
    for (int o = 0; o < 121; ++o) pb[o] = bf16_round(pfc_b[o]);
    for (int o = 0; o < 64; ++o) { vb1[o] = bf16_round(vfc1_b[o]); vw2[o] = bf16_round(vfc2_w[o]); }
    t->dense_vb2 = bf16_round(vfc2_b[0]);
    int rc = upload(&t->dense_wp, wp.data(), wp.size() * 2);
    if (!rc) rc = upload(&t->dense_wv, wv.data(), wv.size() * 2);
    if (!rc) rc = upload(&t->dense_pb, pb.data(), pb.size() * 4);
    if (!rc) rc = upload(&t->dense_vb1, vb1.data(), vb1.size() * 4);
    if (!rc) rc = upload(&t->dense_vw2, vw2.data(), vw2.size() * 4);
    return rc;
}

int af_tower_dense(af_tower* t, void* stream, const void* vin_dev, const void* pin_dev, float* policy_dev, float* value_dev,
                   int32_t batch) {
    if (!t || !vin_dev || !pin_dev || !policy_dev || !value_dev || batch < 1) return AF_TOWER_ERR_ARG;
    if (!t->dense_wp) return AF_TOWER_ERR_STATE;
    DenseArgs a;
    a.vin = static_cast<const __bf16*>(vin_dev); a.pin = static_cast<const __bf16*>(pin_dev);
    a.wp = t->dense_wp; a.wv = t->dense_wv; a.pb = t->dense_pb; a.vb1 = t->dense_vb1; a.vw2 = t->dense_vw2; a.vb2 = t->dense_vb2;
    a.policy = policy_dev; a.value = value_dev; a.batch = batch;
    const int blocks = (batch + 31) / 32;
    hipLaunchKernelGGL(af_tower_dense_kernel, dim3(blocks < 1024 ? blocks : 1024), dim3(256), 0, static_cast<hipStream_t>(stream), a);
    TW_HIP_OK(hipGetLastError());
    return AF_TOWER_OK;
}

int af_tower_stem(af_tower* t, void* stream, const float* planes_dev, void* x_dev, int32_t batch) {
    if (!t || !planes_dev || !x_dev || batch < 1) return AF_TOWER_ERR_ARG;
    if (!t->stem_w) return AF_TOWER_ERR_STATE;
    StemArgs a;
    a.planes = planes_dev; a.w = t->stem_w; a.bias = t->stem_b; a.out = static_cast<char*>(x_dev); a.batch = batch;
    const int grid = batch < 1024 ? batch : 1024;
    hipLaunchKernelGGL(af_tower_stem_kernel, dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream), a);
    TW_HIP_OK(hipGetLastError());
    return AF_TOWER_OK;
}

int af_tower_heads(af_tower* t, void* stream, const void* x_dev, void* vin_dev, void* pin_dev, int32_t batch) {
    if (!t || !x_dev || !vin_dev || !pin_dev || batch < 1) return AF_TOWER_ERR_ARG;
    if (!t->heads_w) return AF_TOWER_ERR_STATE;
    HeadsArgs a;
    a.x = static_cast<const char*>(x_dev); a.w = t->heads_w; a.b = t->heads_b;
    a.vin = static_cast<__bf16*>(vin_dev); a.pin = static_cast<__bf16*>(pin_dev); a.batch = batch;
    const int grid = batch < 2048 ? batch : 2048;
    if (g_heads == 0) {
        hipLaunchKernelGGL(af_tower_heads_kernel, dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream), a);
    } else {
        HeadsMfmaArgs m;
        m.x = a.x; m.a = t->heads_a; m.b = t->heads_b32; m.vin = a.vin; m.pin = a.pin; m.batch = batch;
        hipLaunchKernelGGL(af_tower_heads_mfma_kernel, dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream), m);
    }
    TW_HIP_OK(hipGetLastError());
    return AF_TOWER_OK;
}

int32_t af_tower_pix(const af_tower*) { return kPIX; }
int64_t af_tower_plane_elems(const af_tower*) { return (int64_t)128 * kPIX; }

int af_tower_forward(af_tower* t, void* stream, void* x_dev, void* g_dev, int32_t batch) {
    if (!t || !x_dev || !g_dev || batch < 1) return AF_TOWER_ERR_ARG;
    for (char c : t->set) if (!c) return AF_TOWER_ERR_STATE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    int dev = 0, ncu = 256;
    TW_HIP_OK(hipGetDevice(&dev));
    TW_HIP_OK(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev));
    if (g_grid > 0) ncu = g_grid;
    const int grid = batch < ncu ? batch : ncu;
    for (int b = 0; b < t->blocks; ++b) {
        TowerArgs a;
        a.batch = batch; a.abl = g_abl; a.dump = t->dump;
        const char* xin = static_cast<const char*>(x_dev);
        a.in = xin; a.in2 = nullptr; a.w = t->w1[b]; a.bias = t->b1[b]; a.out = static_cast<char*>(g_dev);
        // ring depth per kernel: the deepest that hipcc allocates without scratch (a scratch reload's vmcnt(0) would
        // also wait for the LDS-DMA of the next position: measured 400 vs 230 us per launch)
        const int d1 = g_depth ? g_depth : 12, d2 = g_depth ? g_depth : 8;
        if (g_engine == 2 || g_engine == 3) {     // r4: epilogue overlapped with the other tile pair's MFMAs (3: first convolution only)
            hipLaunchKernelGGL((af_tower_conv3<false, 8>), dim3(grid), dim3(256), kLds0 + 2 * kPlaneB + kZeroB, st, a);
            a.in = static_cast<const char*>(g_dev); a.in2 = xin; a.w = t->w2[b]; a.bias = t->b2[b];
            a.out = static_cast<char*>(x_dev);
            if (g_engine == 2) hipLaunchKernelGGL((af_tower_conv3<true, 8>), dim3(grid), dim3(256), kLds0 + 3 * kPlaneB + kZeroB, st, a);
            else hipLaunchKernelGGL((af_tower_conv<true, 8>), dim3(grid), dim3(256), kLds0 + 3 * kPlaneB + kZeroB, st, a);
            continue;
        }
        if (d1 == 16) hipLaunchKernelGGL((af_tower_conv<false, 16>), dim3(grid), dim3(256), kLds0 + 2 * kPlaneB + kZeroB, st, a);
        else if (d1 == 12) hipLaunchKernelGGL((af_tower_conv<false, 12>), dim3(grid), dim3(256), kLds0 + 2 * kPlaneB + kZeroB, st, a);
        else hipLaunchKernelGGL((af_tower_conv<false, 8>), dim3(grid), dim3(256), kLds0 + 2 * kPlaneB + kZeroB, st, a);
        a.in = static_cast<const char*>(g_dev); a.in2 = xin; a.w = t->w2[b]; a.bias = t->b2[b];
        a.out = static_cast<char*>(x_dev);
        if (d2 == 16) hipLaunchKernelGGL((af_tower_conv<true, 16>), dim3(grid), dim3(256), kLds0 + 3 * kPlaneB + kZeroB, st, a);
        else if (d2 == 12) hipLaunchKernelGGL((af_tower_conv<true, 12>), dim3(grid), dim3(256), kLds0 + 3 * kPlaneB + kZeroB, st, a);
        else hipLaunchKernelGGL((af_tower_conv<true, 8>), dim3(grid), dim3(256), kLds0 + 3 * kPlaneB + kZeroB, st, a);
    }
    TW_HIP_OK(hipGetLastError());
    return AF_TOWER_OK;
}

#ifdef AF_TOWER_TIMING
int af_tower_debug_cycles(unsigned long long* host) {
    TW_HIP_OK(hipDeviceSynchronize());
    TW_HIP_OK(hipMemcpyFromSymbol(host, HIP_SYMBOL(g_tower_cyc), sizeof(unsigned long long) * 2 * 256 * 4 * 5));
    return AF_TOWER_OK;
}
#endif

int64_t af_tower_flops_per_position(const af_tower* t) {
    return (int64_t)2 * t->blocks * (128 * 128 * 9 * 2 + 128 * 128) * kNPIX;
}

}  // extern "C"
