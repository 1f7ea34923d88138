// af_conv_f16s.hip — the 3x3 convolutions of the reference network (genData/network.py:52-56: residual(f,u) =
// ELU(conv1x1(f) + conv3x3(ELU(conv3x3(f))))) as implicit GEMMs on the fp16 matrix cores with fp32-class accuracy.
//
// Arithmetic ("fp16 split operands"): every fp32 operand x is carried as two halves, x = hi + lo with
// hi = fp16(x), lo = fp16(x - hi) (22 mantissa bits; the MFMA keeps fp16 denormals, probed on gfx950), and a
// product is three v_mfma_f32_32x32x16_f16 with fp32 accumulation:  W_hi*X_hi + W_hi*X_lo + W_lo*X_hi  (the dropped
// W_lo*X_lo term is 2^-22 relative).  That is 16/3 of the fp32 MFMA rate.  Weights are scaled by a power of two per
// layer so that their halves sit in the fp16 normal range; the epilogue multiplies back (exact).  Measured against
// the fp64 restatement of the network: |dv| 1.1e-6, |dp| 1.0e-6 (fp32 arithmetic: 4e-7), bar 1e-5.
//
// Activations live in HBM already split, in "S32" layout: [position][C/32 slabs][hi|lo][4 unit rows][144 units][8 ch]
// fp16 — a unit is 8 channels of one pixel (16 bytes = one lane's MFMA B operand), pixel n = 11y + x sits at unit n + 11
// (rows back to back, 11 zero units above, 12 below; the zeros are never written).  A 32-channel slab is 18,432
// contiguous bytes.
//
// Kernel af_conv_f16s<NSM, NSP, CT, KS, PS, OUT32, XACC, PJ>: NSM 32-channel slabs of the 3x3 input, NSP slabs of the block input
// whose 1x1 projection is folded in as extra k-steps.  Weight-stationary and persistent: a workgroup = 4 waves (one per
// SIMD, 512 registers) = CT cout tiles of 32 x KS halves of every 32-channel slab (k-split: wave ks takes channels
// 16ks..16ks+15) x PS pixel groups (4/PS tiles of 32 pixels); each wave keeps ALL its weight fragments (hi and lo) in
// registers for the whole launch and the workgroup walks over positions.  Input slabs stream through a two-slot LDS
// ring by LDS-DMA (global_load_lds_dwordx4), slab t+1 landing while slab t multiplies; a B fragment is one
// ds_read_b128 at lane base + immediate (conflict-free: consecutive lanes, consecutive units), the left / right
// neighbours that fall off a board row are read from an all-zero LDS region through a second base register.
// k-split partial sums meet through LDS once per position; epilogue: scale, bias, ELU, split, 16-byte stores.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <map>
#include <string>
#include <vector>

#include "af_conv_f16s.h"

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define FS_HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "[af_conv_f16s] %s failed: %s\n", #x, hipGetErrorString(e_)); return -2; } } while (0)

namespace {

// Board geometry.  A kernel instance works on "pseudo-positions" of at most 128 pixels (4 MFMA pixel tiles): an 11x11 board is
// one of them; a 15x15 board is two HALVES (pixels 0..127 and 128..224), each with its own LDS window of the unit rows — that
// is what keeps a 32-channel slab of the ring at ~20 KB on the larger board (a whole 15x15 slab is 32 KB: ring + zero region +
// exchange would not fit 160 KB).  A workgroup only ever sees one half (the grid is even), so the geometry is static per lane.
//   HBM ("S32"): [position][C/32 slabs][hi|lo][4 unit rows][ROWU units][8 ch] fp16; pixel n sits at unit n + POFF, the other units
//   are zero and never written.  LDS slot: the same rows, units [WBASE(h), WBASE(h) + WINU) of each.
template <int S_> struct Geo;
template <> struct Geo<11> {
    static constexpr int S = 11, NPIX = 121, HALVES = 1, POFF = 11;
    static constexpr uint32_t ROWU = 144, WINU = 144;
    static constexpr int PR = 2, NPC = 4;                       // LDS-DMA pieces (64 units) per unit row; pieces per wave and slab
    static constexpr uint32_t PADL = 8, PADR = 8;               // window units LDS-DMA never writes (zeroed once)
    __host__ __device__ static constexpr uint32_t seg_unit(int seg) { return 8u + 64u * (uint32_t)seg; }    // units 8..71, 72..135
    __host__ __device__ static constexpr uint32_t wbase(int) { return 0u; }
};
template <> struct Geo<15> {
    static constexpr int S = 15, NPIX = 225, HALVES = 2, POFF = 16;
    static constexpr uint32_t ROWU = 256, WINU = 160;
    static constexpr int PR = 3, NPC = 6;                       // units 0..63, 64..127, 96..159 of the window (the third piece overlaps)
    static constexpr uint32_t PADL = 0, PADR = 0;
    __host__ __device__ static constexpr uint32_t seg_unit(int seg) { return seg == 0 ? 0u : (seg == 1 ? 64u : 96u); }
    __host__ __device__ static constexpr uint32_t wbase(int h) { return h ? 96u : 0u; }      // half 1 needs units 128..255
};
template <class G> struct Lay {
    static constexpr uint32_t kRowH = G::ROWU * 16u, kHalfH = 4u * kRowH, kSlabH = 2u * kHalfH;     // HBM: unit row / hi (lo) half / slab
    static constexpr uint32_t kRowL = G::WINU * 16u, kHalfL = 4u * kRowL, kSlotL = 2u * kHalfL;     // LDS slot
};
constexpr uint32_t kLds0 = 256u;                // slack so that unit -1 of a slot stays inside LDS
#ifndef AF_F16S_DIST
#define AF_F16S_DIST 3
#endif
constexpr int kDist = AF_F16S_DIST;             // prefetch distance in slabs: slab t multiplies while t+1 .. t+kDist land
template <class G, int DIST = kDist> struct Lds {                        // DIST + 2 LDS slots (see the fragment prefetch in the kernel)
    static constexpr uint32_t kZoff = kLds0 + (DIST + 2) * Lay<G>::kSlotL;    // all-zero region (edge lanes)
    static constexpr uint32_t kBiasOff = kZoff + Lay<G>::kSlotL;         // 128 floats
    static constexpr uint32_t kScrOff = kBiasOff + 512u;                 // k-split exchange
};
// fused head inputs (see "The dense layers of the two heads" below)
template <class G> struct Hx {
    static constexpr int HXP = 128 * G::HALVES;
    static constexpr uint32_t kValPos = HXP * 16u, kValLo = HXP * 8u;       // value: bytes per position / offset of the lo halves
    static constexpr uint32_t kPolPos = HXP * 64u, kPolLo = HXP * 32u;      // policy
    static constexpr int VSTEPS = (G::NPIX + 3) / 4;                        // value fc1 k-steps (4 pixels x 4 channels): 31 / 57
    static constexpr int NLT = (G::NPIX + 31) / 32;                         // policy logit tiles of 32: 4 / 8
};

// (11x11 only: the MFMA stem below)
constexpr int kS = 11, kPIX = 144, kNPIX = kS * kS;
constexpr uint32_t kRowB = kPIX * 16u, kHalfB = 4u * kRowB, kSlabB = 2u * kHalfB;

// Choices that were build-time A/B switches until r6 (tools/probe_f16s_ab.py on 4096 positions, r3_06 / r3_08), now fixed in the source:
//   the LDS-DMA slab loads carry the nt (streaming) hint: 1.385 -> 1.357 ms per forward (the other scope-bit codes measured no better);
//   a position's 3x3 slabs stream BEFORE the slabs of the folded 1x1 projection: 1.357 -> 1.352 ms, and |dv| 1.5e-6 -> 8.7e-7 (the
//   small projection terms are added last; the other order also moved |dp| from 6.0e-6 to 8.1e-6);
//   no nt hint on the activation stores: no gain (1.421 vs 1.421).
// Tried and removed (r3): the lo halves of the split by v_fma_mixlo/mixhi_f16 (one instruction per element instead of convert back,
// subtract, convert): same bits, 1.433 vs 1.436 ms (profiles/r3_15).  Leaving the epilogue's stores in flight across the next position's first LDS-DMA waits (vmcnt counts
// stores too and a wave's operations retire in issue order — tools/probes/vmcnt_order.hip — so the counts can be relaxed by the
// number of younger stores): correct, 1.421 vs 1.421 ms — the waits do not sit on store acknowledgements; the epilogue is
// VALU-bound (profiles/r3_07).  Stores issued by hand in an asm block: corrupt activations (a hazard the compiler cannot see).
// (A/B record r3_43, removed from the source in r6: m0 declared clobbered by the LDS-DMA asm instead of saved / restored around it:
//  -0.2 %, but hipcc warns that a clobber of the reserved m0 "may not be preserved")
// (r3_43, rejected: requesting past the end of the slab stream — the last position again — so that no branch surrounds the LDS-DMA
//  instructions: +1 %, the surplus loads cost more than the branches)
#ifndef AF_F16S_APIN
#define AF_F16S_APIN 1          // weights beyond AF_F16S_VW registers are pinned into accumulator registers (MFMA A operands)
#endif
#ifndef AF_F16S_VW
#define AF_F16S_VW 128          // architectural VGPRs given to weight fragments
#endif
//   AF_F16S_ZPAD       1: the lanes of a pixel tile that lie past the board (7 of 128 at 11x11, 31 of 128 in the second half of
//                         a 15x15 board) read their B fragments from the all-zero LDS region instead of a real pixel's: the
//                         results of those columns are discarded either way, but zero operands toggle fewer MFMA bits and the
//                         chip's clock is power-bound under this load (MI355X_MICROARCH.md, DVFS)
#ifndef AF_F16S_ZPAD
#define AF_F16S_ZPAD 1
#endif

__device__ __forceinline__ void st16(void* gdst, const h8& v) { *reinterpret_cast<h8*>(gdst) = v; }

// Write-through ("sc1": agent scope) stores for data that another workgroup of the SAME launch reads (the roles of af_small_forward_f16s):
// the line goes through this XCD's L2 to memory, so the consumer — on whatever XCD — needs no cache-wide release / acquire, only the
// producer's `s_waitcnt vmcnt(0)` before it raises its flag (MI355X_MICROARCH.md price table: "publish-large", "handoff-flag").  By hand:
// the compiler has no 16-byte scoped store; the s_nop covers the store-data hazard it would otherwise cover itself (the data
// registers of a > 64-bit store must not be overwritten in the next wait states).
template <class V16>
__device__ __forceinline__ void st16_sc1(void* gdst, const V16& v) {
    static_assert(sizeof(V16) == 16, "16-byte store");
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(gdst), "v"(v) : "memory");
}
template <class V8>
__device__ __forceinline__ void st8_sc1(void* gdst, const V8& v) {
    static_assert(sizeof(V8) == 8, "8-byte store");
    asm volatile("global_store_dwordx2 %0, %1, off sc1" : : "v"(gdst), "v"(v) : "memory");
}

// s_waitcnt vmcnt(n) for an n that is a constant once the surrounding loops are unrolled (the immediate is part of the instruction)
__device__ __forceinline__ void wait_vmcnt(int n) {
#define AF_VM(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
    switch (n) {
        AF_VM(0) AF_VM(1) AF_VM(2) AF_VM(3) AF_VM(4) AF_VM(5) AF_VM(6) AF_VM(7) AF_VM(8) AF_VM(9) AF_VM(10) AF_VM(11) AF_VM(12) AF_VM(13)
        AF_VM(14) AF_VM(15) AF_VM(16) AF_VM(17) AF_VM(18) AF_VM(19) AF_VM(20) AF_VM(21) AF_VM(22) AF_VM(23) AF_VM(24) AF_VM(25) AF_VM(26)
        AF_VM(27) AF_VM(28) AF_VM(29) AF_VM(30) AF_VM(31) AF_VM(32) AF_VM(33) AF_VM(34) AF_VM(35) AF_VM(36) AF_VM(37) AF_VM(38) AF_VM(39)
        AF_VM(40) AF_VM(41) AF_VM(42) AF_VM(43) AF_VM(44) AF_VM(45) AF_VM(46) AF_VM(47) AF_VM(48) AF_VM(49) AF_VM(50) AF_VM(51) AF_VM(52)
        AF_VM(53) AF_VM(54) AF_VM(55) AF_VM(56) AF_VM(57) AF_VM(58) AF_VM(59) AF_VM(60) AF_VM(61) AF_VM(62) AF_VM(63)
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
#undef AF_VM
}

// LDS-DMA: 16 bytes per lane from global memory into LDS at (wave-uniform lds_dst) + lane*16; counted on vmcnt.
__device__ __forceinline__ void glds16(const void* gsrc, uint32_t lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

// ... and the agent-scope form for slabs another workgroup of the same launch has just written (no nt on a hand-off: price table "nt-handoff")
__device__ __forceinline__ void glds16_sc1(const void* gsrc, uint32_t lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off sc1\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

// ELU as max(x, min(exp(x), 1) - 1): one v_max_f32 instead of compare + select, and the clamp rides on v_exp_f32 (r3_44).  Against
// x > 0 ? x : exp(x) - 1 the result differs only for -3e-4 < x < 0, where the rounding of v_exp_f32 can put exp(x) - 1 an ulp of 1.0
// (6e-8) below x and the max then returns x — itself within x^2/2 < 5e-8 of the true value: |dv| / |dp| against the fp64 restatement
// are unchanged to all printed digits (profiles/r3_44, r3_54; the compare / select form was removed from the source in r6).
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float elu1(float x) {
    const float e = __builtin_amdgcn_exp2f(x * 1.44269504088896341f);
    return fmaxf(x, fminf(fmaxf(e, 0.0f), 1.0f) - 1.0f);
}
// ... and two at a time on 2-vectors, which keeps the bias add in front and the "- 1" as v_pk_add_f32 (hipcc's SLP vectoriser pairs them
// in the compare / select form but not in the max form): 2.5 + v_exp_f32 instead of 4 + v_exp_f32 issue slots per element
__device__ __forceinline__ f32x2 elu2(f32x2 x) {
    const f32x2 t = x * 1.44269504088896341f;
    f32x2 e;
    e.x = fminf(fmaxf(__builtin_amdgcn_exp2f(t.x), 0.0f), 1.0f);
    e.y = fminf(fmaxf(__builtin_amdgcn_exp2f(t.y), 0.0f), 1.0f);
    const f32x2 em1 = e - 1.0f;
    return f32x2{fmaxf(x.x, em1.x), fmaxf(x.y, em1.y)};
}

#ifndef AF_F16S_XCD_SWIZZLE
#define AF_F16S_XCD_SWIZZLE 1
#endif
#ifdef AF_F16S_TIMING
// profiling build only (tools/probe_f16s_timing.py): cycles per phase, [layer][workgroup (x + 256 y)][wave][phase]
// phase 0 items (MFMA + reads + DMA issue), 1 vmcnt waits, 2 slab barriers, 3 k-split exchange (7: its barrier alone), 4 epilogue,
// 5 total, 6 positions; 8 start-up (kernel entry -> weights and first slabs in place)
__device__ unsigned long long g_f16s_cycles[10][512][4][9];
// [layer][workgroup][0 = entry, 1 = exit]: the device-wide 100 MHz clock (s_memrealtime), for the spread of workgroup finish times
__device__ unsigned long long g_f16s_wall[10][512][2];
#define AF_T(var) const unsigned long long var = __builtin_readcyclecounter()
#define AF_TACC(slot, a, b) tacc[slot] += (b) - (a)
#else
#define AF_T(var)
#define AF_TACC(slot, a, b)
#endif

// ---- roles of a single-launch forward (af_small_forward_f16s): counters in device memory, agent scope ----
// The producer's stores must be visible to a consumer workgroup on ANOTHER XCD (the L2s are per XCD).  Everything a role hands on is
// stored write-through (st16_sc1 / st8_sc1: agent scope per instruction); every wave drains its stores (vmcnt(0)), the workgroup
// meets at a barrier, one thread bumps the counter.  The consumer polls the counter with agent-scope loads and reads the slabs with
// agent-scope LDS-DMA (glds16_sc1) — df_wait_one — or, where it reads with ordinary loads (the dense layers), invalidates first
// (df_wait_inv).  No cache-wide release anywhere: measured 64.9 -> 57 us per batch-1 forward with one, see profiles/r6_09 / r6_10.
#ifdef AF_DF_TIMING
// profiling build only (tools/probe_small_forward_timeline.py): per role (workgroup), the device-wide 100 MHz clock at 0 entry | 1 wait
// satisfied | 2 body done (before the release) | 3 signalled | 4 exit
__device__ unsigned long long g_df_wall[256][5];
#define DF_T(i) if (threadIdx.x == 0 && blockIdx.x < 256) g_df_wall[blockIdx.x][i] = wall_clock64()
#else
#define DF_T(i)
#endif
__device__ __forceinline__ void df_signal(int* ctr) {
    DF_T(2);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // every wave: its write-through stores are acknowledged ...
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // ... then the counter
    DF_T(3);
}
__device__ __forceinline__ void df_poll(const int* ctr, int need, int* err) {      // one lane; bounded (~seconds)
    int it = 0;
    while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need) {
        __builtin_amdgcn_s_sleep(2);
        if (++it > (1 << 22)) { __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
    }
}
__device__ __forceinline__ void df_wait_one(const int* ctr, int need, int* err) {
    if (threadIdx.x == 0) df_poll(ctr, need, err);
    DF_T(1);
    __syncthreads();
}
__device__ __forceinline__ void df_wait_inv(const int* ctr, int need, int* err) {
    if (threadIdx.x == 0) {
        df_poll(ctr, need, err);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // one invalidate for the workgroup (the L1 is the CU's, the L2 the XCD's) ...
    }
    DF_T(1);
    __syncthreads();                                        // ... and nobody loads before it
}

struct F16sArgs {
    const char* in;       // S32, NSM slabs per position
    const char* in2;      // S32, NSP slabs per position (block input for the folded 1x1 projection)
    const uint4* w;       // [cout tile][ks][2 * items][64 lanes] A fragments (8 fp16), hi then lo per item
    const float* bias;    // [COUT] (conv2: conv2 bias + projection bias)
    char* out;            // S32, COUT/32 slabs per position
    float* out32;         // OUT32: fp32 [position][COUT][PP] padded planes (the head kernels' input)
    float inv_scale;      // 1 / weight scale
    // PJ = 1 (producer): this conv1 kernel also computes the block's 1x1 projection of ITS OWN input (the same slabs are in
    // LDS anyway) and writes it, fp32, in accumulator layout; PJ = 2 (consumer): the conv2 kernel adds it in its epilogue
    // instead of streaming the block input a second time.  pbuf: [position][cout tile][pixel tile][4][64 lanes] float4.
    const uint4* pw;      // PJ = 1: projection A fragments [cout tile][ks][2 * NSM * C16][64]
    float* pbuf;
    float inv_scale_p;
    // HD > 0 (fused head input): the last conv of a head branch also applies the head's 1x1 convolution (32 -> 4 value / 16 policy
    // channels) + ELU to the tile it has just finished — the activations a lane holds ARE two MFMA B fragments — and writes
    // the result split into halves in the layout the dense kernels read (see af_value_fc_f16s / af_policy_fc_f16s)
    const uint4* hw;      // [k-step 2][hi|lo][64 lanes] A fragments of the 1x1 convolution, MFMA row m -> cout 8*((m>>2)&1) + 4*(m>>3) + (m&3)
    const float* hbias;   // [16]
    char* hx;             // value: [position][hi|lo][128 pixels][4] fp16; policy: [position][hi|lo][128 pixels][16] fp16
    float inv_scale_h;
    int batch, WP, PP;
    const uint4* w2;      // fused block (FU): the second convolution's packed A fragments (36,864 bytes), its bias [32] and 1 / weight scale
    const float* bias2;
    float inv_scale2;
    int gx0;              // af_conv_f16s_h15: workgroups (per blockIdx.y) of the half-0 class
    char* stash;          // af_conv_f16s_h15 -> af_corner_f16s: [board][slab of the stream][hi|lo][4 unit rows][pixels 208, 209, 223, 224] units
    // DF = 1 (af_small_forward_f16s, r6): this body is one ROLE of a single-launch forward — it may start its slab stream only when
    // df_wait[position] has reached df_need (the roles that write its input have signalled) and signals df_done[position] when its
    // own stores are out; df_err is set if a wait gives up (a bounded spin: a lost role must not hang the device)
    const int* df_wait;
    int df_need;
    int* df_done;
    int* df_err;
    int abl;              // profiling: bit 0 no LDS-DMA after the first slabs, bit 1 no stores, bit 2 LDS-DMA from L2-hot addresses,
                          // bit 3 (r6) activation stores / bit 12 slab loads addressed modulo 128 positions (the upper bound of any
                          // "intermediate tensors never leave the L2s" design: same instruction streams, no HBM behind them)
};

// The kernel body.  NTW: pixel tiles per wave (4 / PS for a whole pseudo-position); tb: the first pixel tile of this workgroup
// (af_conv_f16s_sb, r5: at small batches a pseudo-position's tiles are split over several workgroups — a tile's accumulation does not
// depend on which workgroup runs it, so the results are the batched kernel's bit for bit).  HSEL (r5, 15x15 only): -1 = the workgroups of
// this launch take both halves of the boards alternately (bx even / odd, the r3 scheme); 0 / 1 = the `gxw` workgroups of this CLASS
// (index bx) take that half of EVERY board — af_conv_f16s_h15 runs a class of NTW = 4 workgroups on half 0 (pixels 0..127) next to
// a class of NTW = 3 workgroups on half 1 (pixels 128..223: three full tiles; pixel 224, the fourth tile's only pixel, is left to
// af_corner_f16s) in one launch.
// FU (r5): 1 / 2 = the block's SECOND convolution (32 -> 32, the value / policy branch's last one, with the fused head) runs in this kernel
// too, on the first one's output kept in LDS — see "fused block" below.
template <class G, int NSM, int NSP, int CT, int KS, int PS, bool OUT32, bool XACC, int PJ, int HD, int DIST, int WPE, int NTW, int HSEL, int FU = 0, int DF = 0>
__device__ __forceinline__ void f16s_body(const F16sArgs& A, char* smem, const int bx, const int gxw, const int tb = 0,
                                          const int by = (int)blockIdx.y, const int gy = (int)gridDim.y) {
    using L = Lay<G>;
    constexpr uint32_t kRowH = L::kRowH, kHalfH = L::kHalfH, kSlabH = L::kSlabH, kRowL = L::kRowL, kHalfL = L::kHalfL, kSlotL = L::kSlotL;
    constexpr uint32_t kZoff = Lds<G, DIST>::kZoff, kBiasOff = Lds<G, DIST>::kBiasOff, kScrOff = Lds<G, DIST>::kScrOff;
    constexpr int kDist = DIST, kRing = DIST + 2;                             // (shadow the file-level default)
    constexpr int NPC = G::NPC, HV = G::HALVES;
    // Fused block (FU > 0; blocks 3 and 5 of the 11x11 net: 128 -> 32 -> 32 and 64 -> 32 -> 32).  Unfused, the first convolution
    // writes its 32 channels (15 KB, split) and the block's projection (16 KB, fp32) to HBM and the second one reads both back: 64 KB
    // of a block's 103 / 129 KB per position, on layers that run at 4-5 TB/s (profiles/r5_01: HBM-bound).  Fused: every wave ends
    // the first convolution owning ONE pixel tile (PS = 4, or the k-split's finished tile; tile index = wave index in both
    // convolutions), writes its activated tile into an LDS slab in ring-slot layout, keeps the projection's accumulators in
    // registers, and after a barrier runs the second convolution on that slab: B fragments exactly like ring-slot reads, A fragments
    // from a copy of the second layer's 36 KB of packed weights in LDS (320 + 160 weight registers do not fit a wave) — the same
    // MFMAs on the same operands in the same order as af_conv_f16s<G,1,0,1,1,4,true,true,2,HD> runs them, then its epilogue and the
    // fused head: bit-identical outputs (the digest of 4096 positions equals the unfused build's).
    static_assert(FU == 0 || (G::HALVES == 1 && PJ == 1 && HD == 0 && !OUT32 && NSP == 0 && CT == 1 && HSEL < 0 &&
                              ((KS == 1 && NTW == 1) || (KS == 2 && NTW == 2))), "fused block: see above");
    constexpr uint32_t kScrBytes = KS == 2 ? (uint32_t)(CT * PS * NTW) * 4096u * (PJ == 1 ? 2u : 1u) : 0u;
    constexpr uint32_t kGOff = Lds<G, DIST>::kScrOff;                         // KS = 2: inside the exchange scratch (pads re-zeroed per position)
    constexpr uint32_t kW2Off = kGOff + (kScrBytes > Lay<G>::kSlotL ? kScrBytes : Lay<G>::kSlotL);
    constexpr int NT = NTW;                       // pixel tiles per wave
    static_assert(HSEL < 0 || (HV == 2 && PJ == 0 && HD == 0 && !OUT32), "half classes: plain 15x15 layers only");
    static_assert(NT % KS == 0 || PJ == 0, "an odd tile count under a k-split has no projection exchange");
    constexpr int C16 = 2 / KS;                   // 16-channel k-steps per slab and wave
    constexpr int ITP = C16, ITM = 9 * C16;       // items (k-step x tap) per projection / main slab
    constexpr int NIT = NSP * ITP + NSM * ITM;    // items per position and wave
    constexpr int SPP = NSP + NSM;                // slabs per position
    constexpr int NFIN = NT / KS;                 // tiles a wave finishes (epilogue) after the k-split exchange (odd NT: see OWN below)
    static_assert(CT * KS * PS == 4 && NT >= KS, "4 waves");
    static_assert(PJ == 0 || NSP == 0, "a producer / consumer of the separate projection has no projection slabs");
    constexpr int NPW = PJ == 1 ? NSM * C16 : 0;  // projection items (centre tap of every k-step)
    static_assert(NPC * (DIST > 1 ? DIST - 1 : 1) < 64, "vmcnt is a 6-bit field");
    static_assert(DIST >= 2 || (NSP == 0 && ITM - 1 >= NPC), "distance 1: a slab's pieces must all be issued before its predecessor's last item");
    const int lane = threadIdx.x & 63, kg = lane >> 5, nn = lane & 31;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int ct = wv % CT, ks = (wv / CT) % KS, ps = wv / (CT * KS);
    // k-split: register slot jj of a wave holds pixel tile (jj + rot) mod NT, so that the tiles a wave finishes are ALWAYS its slots
    // 0 .. NFIN-1 and the ones it hands to its partner its slots NFIN .. NT-1: exchange, combine and epilogue are unconditional
    // straight-line code (with "tile jj / NFIN == ks" they were branches on ks, and every accumulator made a round trip through
    // v_accvgpr_write / v_accvgpr_read at the merge points)
    // Odd NT (r5: the 3-tile half of a 15x15 board): wave ks = 0 finishes (NT + 1) / 2 tiles, wave ks = 1 the other NT / 2; the slot in
    // the middle is the only one whose role depends on the wave (a scalar branch in exchange / combine / epilogue).
    const int rot = KS == 2 ? ks * ((NT + 1) / 2) : 0;
#define AF_OWN(jj) (KS == 1 || (jj) < NT / 2 || ((jj) < (NT + 1) / 2 && ks == 0))
    const int ctg = by * CT + ct, nso = gy * CT;
    const uint32_t lds = (uint32_t)(uintptr_t)smem;

    // pseudo-position q = HALVES * position + half; this workgroup takes q0, q0 + gridDim.x, ... (gridDim.x is a multiple of
    // HALVES, so the half — and with it every lane's pixel, window and edge flags — is fixed for the whole launch)
    const int nq = HSEL >= 0 ? A.batch : A.batch * HV;
    // Which pseudo-positions a workgroup takes (r4): consecutive workgroup ids land on consecutive XCDs, so with q0 = blockIdx.x the 32
    // workgroups of XCD k would walk the positions = k (mod 8) only — one residue class of the activation addresses per XCD, all in
    // step — and the XCDs finish 3 % apart, launch after launch, even ones late (profiles/r4_03).  With the swizzle XCD k takes the
    // CONTIGUOUS positions 32k .. 32k+31 (+ gridDim.x per pass): every XCD touches every address residue.  A position's result does
    // not depend on which workgroup computes it (same bits); gridDim.x / 8 is a multiple of HALVES whenever the swizzle is used.
    int qpos = bx;
    if (AF_F16S_XCD_SWIZZLE && (gxw & (8 * (HSEL >= 0 ? 1 : HV) - 1)) == 0) qpos = (bx >> 3) + (gxw >> 3) * (bx & 7);
    if (qpos >= nq) return;
    const int hv = HSEL >= 0 ? HSEL : (HV == 1 ? 0 : qpos % HV);
    const uint32_t wsrc = G::wbase(hv) * 16u;          // byte offset of the half's window inside a unit row (HBM)
    AF_T(t_entry);
#ifdef AF_F16S_TIMING
    const unsigned long long wall_entry = wall_clock64();
#endif

    // slab j of a position: the NSM slabs of the 3x3 input first, then the NSP slabs of the projection input
    auto slab_src = [&](int qq, int j) -> const char* {             // (qq: pseudo-position)
        const int p_ = (HV == 1 || HSEL >= 0) ? qq : qq / HV;
        const int p = (A.abl & 0x1000) ? (p_ & 127) : p_;   // abl bit 12 (profiling): slab LOADS addressed modulo 128 positions (cache-resident ring); bit 3: the stores
        return (j < NSM ? A.in + ((size_t)p * NSM + j) * kSlabH : A.in2 + ((size_t)p * NSP + (j - NSM)) * kSlabH) + wsrc;
    };
    // piece wv + 4k of a slab, k = 0..NPC-1: every wave issues exactly NPC LDS-DMA instructions (1 KB each) per slab — one per
    // item, the rest behind the slab's last item — which is what lets the counted "s_waitcnt vmcnt" in front of each barrier
    // say "everything but what was requested after slab t+1 has landed" (see the last-item branch below).  A piece is 64
    // units of one unit row of the window (11x11: units 8..71 / 72..135, pixels sit at units 11..131 and the other units of a
    // slot are zeroed once and never written; 15x15: units 0..63 / 64..127 / 96..159 of the half's 160-unit window).
    auto dma_piece = [&](const char* src, uint32_t slot_off, int k) {
        const int piece = wv + 4 * k, row = piece / G::PR;                    // row: 0..7 = (hi|lo, unit row)
        const uint32_t su = G::seg_unit(piece % G::PR) * 16u;
        const uint32_t offh = (uint32_t)(row >> 2) * kHalfH + (uint32_t)(row & 3) * kRowH + su;
        const uint32_t offl = (uint32_t)(row >> 2) * kHalfL + (uint32_t)(row & 3) * kRowL + su;
        if (DF != 0) glds16_sc1(src + offh + lane * 16, (uint32_t)__builtin_amdgcn_readfirstlane((int)(lds + kLds0 + slot_off + offl)));
        else glds16(src + offh + lane * 16, (uint32_t)__builtin_amdgcn_readfirstlane((int)(lds + kLds0 + slot_off + offl)));
    };
    // slab number `idx` of the stream this workgroup consumes (pseudo-positions q0, q0 + gridDim.x, ...; SPP slabs each)
    const int q0 = qpos;
    auto stream_src = [&](uint32_t idx) -> const char* {
        const int p = q0 + (int)(idx / SPP) * gxw, j = (int)(idx % SPP);
        return slab_src(p, j);
    };
    const uint32_t nslabs = (uint32_t)((nq - q0 + gxw - 1) / gxw) * SPP;

    // Start-up (5-8 us of every launch, r2_39): the longest latencies first — the first kDist slabs (HBM -> LDS-DMA), then the
    // weights (L2) — and under them the LDS zeroing: only what LDS-DMA never writes (slack, the pad units 0..7 and 136..143 of
    // every unit row of every ring slot, the zero region), so it needs no ordering against the DMA
    // (a role of the single-launch forward asks for its slabs later: when its producers have signalled — the weights do not wait)
    if (DF == 0) {
#pragma unroll
        for (int d = 0; d < kDist; ++d) {
            if ((uint32_t)d < nslabs) {
#pragma unroll
                for (int k = 0; k < NPC; ++k) dma_piece(stream_src(d), (uint32_t)d * kSlotL, k);
            }
        }
    }

    // resident weights: 2 * NIT fragments
    h8 W[2 * NIT];
    {
        const uint4* wp = A.w + ((size_t)(ctg * KS + ks) * (2 * NIT)) * 64 + lane;
#pragma unroll
        for (int f = 0; f < 2 * NIT; ++f) {
            const uint4 v = wp[f * 64];
            __builtin_memcpy(&W[f], &v, 16);
        }
    }
    h8 PW[PJ == 1 ? 2 * NPW : 1];
    if (PJ == 1) {
        const uint4* wp = A.pw + ((size_t)(ctg * KS + ks) * (2 * NPW)) * 64 + lane;
#pragma unroll
        for (int f = 0; f < 2 * NPW; ++f) {
            const uint4 v = wp[f * 64];
            __builtin_memcpy(&PW[f], &v, 16);
        }
    }
    constexpr int HDS = HD > 0 ? HD : FU;         // which head's 1x1 convolution this kernel applies (0: none)
    h8 HA[HDS > 0 ? 4 : 1];
    float hb[HDS > 0 ? 8 : 1];
    if (FU > 0) {
        // the second convolution's A fragments -> LDS ([item][hi|lo][64 lanes] x 16 bytes, pack_layer's order), its bias -> registers
        const uint4* w2 = A.w2;
#pragma unroll
        for (int i = 0; i < 9; ++i) *reinterpret_cast<uint4*>(smem + kW2Off + (uint32_t)(i * 256 + (int)threadIdx.x) * 16u) = w2[i * 256 + threadIdx.x];
        if (threadIdx.x < 32) reinterpret_cast<float*>(smem + kBiasOff + 256u)[threadIdx.x] = A.bias2[threadIdx.x];      // (the first layer's 32 biases sit at + 0)
        if (KS == 1) {      // the activation slab's pad units: zeroed once (KS = 2: the slab lives in the exchange scratch, see the position loop)
            for (uint32_t u = threadIdx.x; u < kSlotL / 16; u += 256) *reinterpret_cast<uint4*>(smem + kGOff + u * 16) = uint4{0, 0, 0, 0};
        }
    }
    if (HDS > 0) {
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            const uint4 v = A.hw[f * 64 + lane];
            __builtin_memcpy(&HA[f], &v, 16);
        }
#pragma unroll
        for (int r = 0; r < 8; ++r) hb[r] = A.hbias[8 * kg + r];
    }
    if (threadIdx.x < kLds0 / 16) *reinterpret_cast<uint4*>(smem + threadIdx.x * 16) = uint4{0, 0, 0, 0};
    if (G::PADL + G::PADR > 0) {
        static_assert(G::PADL + G::PADR == 0 || (G::PADL == 8 && G::PADR == 8), "pad zeroing is written for 8 + 8 units");
        for (uint32_t u = threadIdx.x; u < (uint32_t)kRing * 128u; u += 256) {      // (slot, unit row 0..7, side, unit 0..7)
            const uint32_t off = kLds0 + (u >> 7) * kSlotL + ((u >> 4) & 7u) * kRowL + (((u >> 3) & 1u) ? (G::WINU - 8u) * 16u : 0u) + (u & 7u) * 16u;
            *reinterpret_cast<uint4*>(smem + off) = uint4{0, 0, 0, 0};
        }
    }
    for (uint32_t u = threadIdx.x; u < kSlotL / 16; u += 256) *reinterpret_cast<uint4*>(smem + kZoff + u * 16) = uint4{0, 0, 0, 0};
    if (threadIdx.x < CT * 32) reinterpret_cast<float*>(smem + kBiasOff)[threadIdx.x] = A.bias[by * CT * 32 + threadIdx.x];
    // lane geometry per pixel tile: LDS byte base of tap (0,0) in slot 0, zero-region twin, output unit
    uint32_t lb[NT], zb[NT];
    int pix[NT];
    bool ok[NT], edgeL[NT], edgeR[NT];
#pragma unroll
    for (int jj = 0; jj < NT; ++jj) {
        const int n = 128 * hv + 32 * (tb + ps * NT + ((jj + rot) % NT)) + nn;
        ok[jj] = n < G::NPIX;
        const int nc = ok[jj] ? n : 128 * hv;                                // (an invalid lane works on the half's first pixel)
        pix[jj] = nc;
        const int x = nc % G::S;
        edgeL[jj] = x == 0; edgeR[jj] = x == G::S - 1;
        // pixel nc sits at unit nc + POFF of a row; tap (ky,kx) reads unit nc + POFF + (ky-1)*S + (kx-1) = u0 + ky*S + kx with
        // u0 = nc + POFF - S - 1, i.e. window unit u0 - WBASE (11x11: nc - 1; 15x15: nc - 0 / nc - 96)
        lb[jj] = kLds0 + (uint32_t)((KS == 2 ? 2 * ks : 0) + kg) * kRowL + (uint32_t)((nc + G::POFF - G::S - 1 - (int)G::wbase(hv)) * 16);
        zb[jj] = kZoff + (lb[jj] & 255u);
    }
    // Pin the weight loads before the loop (see af_tower_bf16.hip).  Where the weights do not fit the 256 architectural VGPRs
    // next to the fragment buffers, the surplus is pinned into accumulator registers ("+a"), from which the MFMA takes its A operand
    // directly: with "+v" for all of them hipcc parked the surplus there anyway and copied every fragment back with four
    // v_accvgpr_read before use (0.7-2 VALU instructions per MFMA in the item loop of the wide layers; r3_38).
    constexpr int kAccRegs = 16 * NT * (1 + (XACC ? 1 : 0) + (PJ == 1 ? 1 : 0));
    constexpr int kWRegs = 4 * (2 * NIT + (PJ == 1 ? 2 * NPW : 0) + (HDS > 0 ? 4 : 0));
    constexpr int kNAmax = (256 - kAccRegs - 16) / 4, kNAwant = (kWRegs - AF_F16S_VW + 3) / 4;
    constexpr int NA = (AF_F16S_APIN && WPE == 1 && kNAwant > 0) ? (kNAwant < kNAmax ? kNAwant : kNAmax) : 0;
#pragma unroll
    for (int f = 0; f < 2 * NIT; ++f) {
        if (f < 2 * NIT - NA) asm volatile("" : "+v"(W[f]));
        else asm volatile("" : "+a"(W[f]));
    }
    if (PJ == 1) {
#pragma unroll
        for (int f = 0; f < 2 * NPW; ++f) asm volatile("" : "+v"(PW[f]));
    }
    if (HDS > 0) {
#pragma unroll
        for (int f = 0; f < 4; ++f) asm volatile("" : "+v"(HA[f]));
    }
    if (DF != 0) {
        df_wait_one(A.df_wait + qpos, A.df_need, A.df_err);
#pragma unroll
        for (int d = 0; d < kDist; ++d) {
            if ((uint32_t)d < nslabs) {
#pragma unroll
                for (int k = 0; k < NPC; ++k) dma_piece(stream_src(d), (uint32_t)d * kSlotL, k);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    AF_T(t_ready);

    // B fragment of an item (k-step c, tap) of a projection / main slab, half p, for the pixel tile whose centre / left /
    // right read bases are given
    auto rd = [](const char* sm, bool proj, int it, int p, uint32_t c_, uint32_t l_, uint32_t r_) -> h8 {
        const int c = proj ? it : it / 9, tap = proj ? 4 : it % 9, ky = tap / 3, kx = tap % 3;
        const uint32_t base = kx == 0 ? l_ : (kx == 2 ? r_ : c_);
        const uint32_t imm = (uint32_t)p * kHalfL + (KS == 1 ? 2u * c * kRowL : 0u) + (uint32_t)(ky * G::S + kx) * 16u;
        return *reinterpret_cast<const h8*>(sm + base + imm);
    };

    // the head's 1x1 convolution on a finished tile (the last convolution of a branch, or the fused block's second one)
    auto head_tile = [&](const float (&v)[16], const int jj, const int pos) {
      if constexpr (HDS > 0) {
                // a lane holds channels 16 kg + r of its pixel: r = 0..7 and r = 8..15 are the B fragments of two k-steps whose k
        // index 8 kgrp + e stands for channel 16 kgrp + 8 step + e (the A fragments are packed to match)
        h8 bh[2], bl[2];
#pragma unroll
        for (int st_ = 0; st_ < 2; ++st_)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float f = v[8 * st_ + e];
                const _Float16 h = (_Float16)f;
                bh[st_][e] = h;
                bl[st_][e] = (_Float16)(f - (float)h);
            }
        f32x16 a2;
#pragma unroll
        for (int r = 0; r < 16; ++r) a2[r] = 0.0f;
#pragma unroll
        for (int st_ = 0; st_ < 2; ++st_) {
            a2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(HA[2 * st_], bh[st_], a2, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(HA[2 * st_], bl[st_], a2, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(HA[2 * st_ + 1], bh[st_], a2, 0, 0, 0);
        }
        // accumulator rows 0..7 of a lane = head channels 8 kg + r
        constexpr int NR = HDS == 1 ? 4 : 8;
        _Float16 uh[8], ul[8];
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const float f = elu1(a2[r] * A.inv_scale_h + hb[r]);
            const _Float16 h = (_Float16)f;
            uh[r] = h;
            ul[r] = (_Float16)(f - (float)h);
        }
        if (ok[jj] && !(A.abl & 2)) {
            if (HDS == 1) {
                if (kg == 0) {
                    typedef _Float16 h4 __attribute__((ext_vector_type(4)));
                    char* o = A.hx + (size_t)pos * Hx<G>::kValPos + (uint32_t)pix[jj] * 8u;
                    if (DF != 0) {
                        st8_sc1(o, h4{uh[0], uh[1], uh[2], uh[3]});
                        st8_sc1(o + Hx<G>::kValLo, h4{ul[0], ul[1], ul[2], ul[3]});
                    } else {
                        *reinterpret_cast<h4*>(o) = h4{uh[0], uh[1], uh[2], uh[3]};
                        *reinterpret_cast<h4*>(o + Hx<G>::kValLo) = h4{ul[0], ul[1], ul[2], ul[3]};
                    }
                }
            } else {
                char* o = A.hx + (size_t)pos * Hx<G>::kPolPos + (uint32_t)pix[jj] * 32u + (uint32_t)kg * 16u;
                if (DF != 0) {
                    st16_sc1(o, h8{uh[0], uh[1], uh[2], uh[3], uh[4], uh[5], uh[6], uh[7]});
                    st16_sc1(o + Hx<G>::kPolLo, h8{ul[0], ul[1], ul[2], ul[3], ul[4], ul[5], ul[6], ul[7]});
                } else {
                    *reinterpret_cast<h8*>(o) = h8{uh[0], uh[1], uh[2], uh[3], uh[4], uh[5], uh[6], uh[7]};
                    *reinterpret_cast<h8*>(o + Hx<G>::kPolLo) = h8{ul[0], ul[1], ul[2], ul[3], ul[4], ul[5], ul[6], ul[7]};
                }
            }
        }
      }
    };
    uint32_t t = 0;
    uint32_t cur = 0u, nxd = (uint32_t)kDist * kSlotL;        // ring slots of slab t and of slab t + kDist
    // Fragments are double buffered per item and prefetched one item ahead ACROSS slab (and position) boundaries: the
    // wait + barrier that publishes slab t+1 sits in front of the LAST item of slab t, whose MFMAs then cover the LDS
    // latency of slab t+1's first fragments.  (Hence kRing = kDist + 2: the slot that LDS-DMA refills during slab t is
    // the one of slab t-2, which every wave left before the barrier inside slab t-1.)
    // XPOS: the prefetch also crosses the position boundary (the fragments stay live through the epilogue) — not where
    // weights + accumulators already fill the register file (4 pixel tiles and >= 288 weight registers)
    constexpr bool XPOS = !(NT >= 3 && NIT >= 36) && FU == 0;      // (fused block: the second convolution needs the registers)
    h8 fr[2][NT][2];
#define AF_FIRST_ITEM(slot)                                                                                      \
    _Pragma("unroll") for (int jj = 0; jj < NT; ++jj) {                                                          \
        const uint32_t c_ = (AF_F16S_ZPAD && !ok[jj]) ? zb[jj] : lb[jj] + (slot), l_ = edgeL[jj] ? zb[jj] : c_, r_ = edgeR[jj] ? zb[jj] : c_; \
        fr[0][jj][0] = rd(smem, false, 0, 0, c_, l_, r_);                                                        \
        fr[0][jj][1] = rd(smem, false, 0, 1, c_, l_, r_);                                                        \
    }
    if (XPOS) { AF_FIRST_ITEM(0u) }
#ifdef AF_F16S_TIMING
    unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    for (; qpos < nq; qpos += gxw) {
        const int pos = (HV == 1 || HSEL >= 0) ? qpos : qpos / HV;
        AF_T(tp0);
        if (!XPOS) { AF_FIRST_ITEM(cur) }
#undef AF_FIRST_ITEM
        // XACC: the cross terms (W_hi*X_lo, W_lo*X_hi; 2^-11 of the main term) get their own accumulator, so the main
        // accumulator is rounded once per item instead of three times (the MFMA's fp32 accumulation is where this path
        // loses accuracy); used where the register budget allows it
        f32x16 acc[NT], acx[XACC ? NT : 1], pac[PJ == 1 ? NT : 1];
#pragma unroll
        for (int jj = 0; jj < NT; ++jj)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[jj][r] = 0.0f; if (XACC) acx[jj][r] = 0.0f; if (PJ == 1) pac[jj][r] = 0.0f; }
        // consumer: this wave's share of the separately computed projection (issued now, used in the epilogue)
        f32x4 padd[PJ == 2 ? NFIN : 1][4];
        if (PJ == 2) {
#pragma unroll
            for (int jf = 0; jf < NFIN; ++jf) {
                const int tile = tb + ps * NT + (KS == 2 ? ks * NFIN : 0) + jf;
                const f32x4* src = reinterpret_cast<const f32x4*>(A.pbuf) + ((((size_t)qpos * nso + ctg) * 4 + tile) * 4) * 64 + lane;
#pragma unroll
                for (int q = 0; q < 4; ++q) padd[jf][q] = src[q * 64];
            }
        }

#pragma clang loop unroll(full)
        for (int j = 0; j < SPP; ++j) {
            // this slab / the next slab of the stream: a projection slab?  (ms = index among the position's 3x3 slabs)
            const bool proj = j >= NSM;
            const bool nproj = (j + 1) % SPP >= NSM;
            const int ms = j, pslab = j - NSM;
            const int NI = proj ? ITP : ITM;
            // ibase: the slab's first item in the weight array (packed projection items first); seq: its running number within the
            // position (fragment double-buffer parity)
            const int ibase = proj ? pslab * ITP : NSP * ITP + ms * ITM;
            const int seq = proj ? NSM * ITM + pslab * ITP : ms * ITM;
            // slab t + kDist of the stream: j is static (the loop is unrolled), so which position / slab that is costs no division
            const int npos = qpos + ((j + kDist) / SPP) * gxw;
            const bool more = npos < nq && !(A.abl & 1);
            const char* nsrc = slab_src((A.abl & 4) ? q0 : (more ? npos : qpos), (j + kDist) % SPP);   // abl bit 2: the first position's slabs again (L2-hot)
            const uint32_t nx1 = cur + kSlotL == kRing * kSlotL ? 0u : cur + kSlotL;     // slot of slab t+1
            uint32_t bC[NT], bL[NT], bR[NT];
#pragma unroll
            for (int jj = 0; jj < NT; ++jj) {
                bC[jj] = (AF_F16S_ZPAD && !ok[jj]) ? zb[jj] : lb[jj] + cur;
                bL[jj] = edgeL[jj] ? zb[jj] : bC[jj];
                bR[jj] = edgeR[jj] ? zb[jj] : bC[jj];
            }
            if (HSEL == 1 && wv == (j & 3) && by == 0 && lane < 32) {
                // The corner pixel's operands, while the slab is in LDS (r5): the four on-board taps of pixel 224 are pixels 208, 209,
                // 223, 224 = window units 128, 129, 143, 144 of each of the slab's 8 unit rows — 32 units of 16 bytes, copied to a compact
                // buffer that af_corner_f16s reads as contiguous 512-byte rows (read from the S32 tensor they are 3 cache lines of 128
                // bytes per row for 64 useful bytes, 50 MB per layer: 25-45 us per corner launch, profiles/r5_09).  The store is younger
                // than the LDS-DMA pieces the counted vmcnt waits are about: it can only make those waits more conservative.
                const uint32_t u = (uint32_t)(lane & 3), un = (u < 2 ? 128u + u : 141u + u);
                const uint4 q = *reinterpret_cast<const uint4*>(smem + kLds0 + cur + (uint32_t)(lane >> 4) * kHalfL + (uint32_t)((lane >> 2) & 3) * kRowL + un * 16u);
                *reinterpret_cast<uint4*>(A.stash + (((size_t)pos * SPP + j) * 32 + lane) * 16) = q;
            }
#pragma clang loop unroll(full)
            for (int it = 0; it < NI; ++it) {
                const int b = (seq + it) & 1;
                if (it + 1 < NI) {
#pragma unroll
                    for (int jj = 0; jj < NT; ++jj) {
                        fr[b ^ 1][jj][0] = rd(smem, proj, it + 1, 0, bC[jj], bL[jj], bR[jj]);
                        fr[b ^ 1][jj][1] = rd(smem, proj, it + 1, 1, bC[jj], bL[jj], bR[jj]);
                    }
                } else {
                    // last item of slab t: slab t+1 must have landed for every wave — only what was requested AFTER its pieces
                    // may still be in flight: kDist - 2 whole slabs and the pieces of slab t + kDist this slab has issued
                    // so far (one per earlier item, at most NPC) — then fetch slab t+1's first fragments
                    const int issued = NI - 1 < NPC ? NI - 1 : NPC;
                    AF_T(tw0);
                    if (more) {
                        wait_vmcnt(NPC * (kDist - 2) + issued);
                    } else {
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    }
                    AF_T(tw1);
                    __builtin_amdgcn_s_barrier();
                    AF_T(tw2);
                    AF_TACC(1, tw0, tw1);
                    AF_TACC(2, tw1, tw2);
                    if ((XPOS || j + 1 < SPP) && t + 1 < nslabs) {
#pragma unroll
                        for (int jj = 0; jj < NT; ++jj) {
                            const uint32_t c_ = (AF_F16S_ZPAD && !ok[jj]) ? zb[jj] : lb[jj] + nx1, l_ = edgeL[jj] ? zb[jj] : c_, r_ = edgeR[jj] ? zb[jj] : c_;
                            fr[b ^ 1][jj][0] = rd(smem, nproj, 0, 0, c_, l_, r_);
                            fr[b ^ 1][jj][1] = rd(smem, nproj, 0, 1, c_, l_, r_);
                        }
                    }
                }
                const h8 wh = W[2 * (ibase + it)], wl = W[2 * (ibase + it) + 1];
#pragma unroll
                for (int jj = 0; jj < NT; ++jj) acc[jj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, fr[b][jj][0], acc[jj], 0, 0, 0);
                if (XACC) {
#pragma unroll
                    for (int jj = 0; jj < NT; ++jj) acx[jj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, fr[b][jj][1], acx[jj], 0, 0, 0);
#pragma unroll
                    for (int jj = 0; jj < NT; ++jj) acx[jj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, fr[b][jj][0], acx[jj], 0, 0, 0);
                } else {
#pragma unroll
                    for (int jj = 0; jj < NT; ++jj) acc[jj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, fr[b][jj][1], acc[jj], 0, 0, 0);
#pragma unroll
                    for (int jj = 0; jj < NT; ++jj) acc[jj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, fr[b][jj][0], acc[jj], 0, 0, 0);
                }
                if (PJ == 1 && !proj && it % 9 == 4) {                     // centre tap: the 1x1 projection reads the very same fragments
                    const int pi = ms * C16 + it / 9;
                    const h8 ph = PW[2 * pi], pl = PW[2 * pi + 1];
#pragma unroll
                    for (int jj = 0; jj < NT; ++jj) pac[jj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ph, fr[b][jj][0], pac[jj], 0, 0, 0);
#pragma unroll
                    for (int jj = 0; jj < NT; ++jj) pac[jj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ph, fr[b][jj][1], pac[jj], 0, 0, 0);
#pragma unroll
                    for (int jj = 0; jj < NT; ++jj) pac[jj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(pl, fr[b][jj][0], pac[jj], 0, 0, 0);
                }
#pragma unroll
                for (int q = 0; q < 2 * NT; ++q) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // MFMA
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // DS read
                }
                __builtin_amdgcn_sched_group_barrier(0x008, NT, 0);
                // slab t + kDist (its slot was slab t-2's: free since the barrier inside slab t-1): this wave's NPC pieces of 1 KB,
                // one per item, the rest behind the last item
                if (more) {
                    if (it < NI - 1 && it < NPC) dma_piece(nsrc, nxd, it);
                    if (it == NI - 1) {
#pragma unroll
                        for (int k = (NI - 1 < NPC ? NI - 1 : NPC); k < NPC; ++k) dma_piece(nsrc, nxd, k);
                    }
                }
            }
            ++t;
            cur = nx1;
            nxd = nxd + kSlotL == kRing * kSlotL ? 0u : nxd + kSlotL;
        }
        if (XPOS && (NIT & 1)) {                                             // keep the fragment parity static per position
#pragma unroll
            for (int jj = 0; jj < NT; ++jj) { fr[0][jj][0] = fr[1][jj][0]; fr[0][jj][1] = fr[1][jj][1]; }
        }

        AF_T(tp1);
        if (XACC) {
#pragma unroll
            for (int jj = 0; jj < NT; ++jj)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[jj][r] += acx[jj][r];
        }
        // k-split: the two waves of a pair exchange the halves they do not finish (accumulators, and the projection's)
#define AF_EXCHANGE(ACC, SCR)                                                                                              \
    {                                                                                                                      \
        _Pragma("unroll") for (int jj = 0; jj < NT; ++jj) {                                                                \
            if (!AF_OWN(jj)) {                                                                                             \
                _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                            \
                    f32x4 v = {ACC[jj][4 * q], ACC[jj][4 * q + 1], ACC[jj][4 * q + 2], ACC[jj][4 * q + 3]};                \
                    *reinterpret_cast<f32x4*>((SCR) + (uint32_t)(((jj + rot) % NT) * 4 + q) * 1024u + lane * 16u) = v;     \
                }                                                                                                          \
            }                                                                                                              \
        }                                                                                                                  \
    }
#define AF_COMBINE(ACC, SCR)                                                                                               \
    {                                                                                                                      \
        _Pragma("unroll") for (int jj = 0; jj < NT; ++jj) {                                                                \
            if (AF_OWN(jj)) {                                                                                              \
                _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                            \
                    const f32x4 v = *reinterpret_cast<const f32x4*>((SCR) + (uint32_t)(((jj + rot) % NT) * 4 + q) * 1024u + lane * 16u); \
                    ACC[jj][4 * q] += v[0]; ACC[jj][4 * q + 1] += v[1]; ACC[jj][4 * q + 2] += v[2]; ACC[jj][4 * q + 3] += v[3]; \
                }                                                                                                          \
            }                                                                                                              \
        }                                                                                                                  \
    }
        if (KS == 2) {
            char* scr = smem + kScrOff + (uint32_t)(ct + CT * ps) * (NT * 4096u);
            char* scp = scr + (uint32_t)(CT * PS) * (NT * 4096u);
            AF_EXCHANGE(acc, scr)
            if (PJ == 1) AF_EXCHANGE(pac, scp)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            AF_T(tx0);
            __builtin_amdgcn_s_barrier();
            AF_T(tx1);
            AF_TACC(7, tx0, tx1);
            AF_COMBINE(acc, scr)
            if (PJ == 1) AF_COMBINE(pac, scp)
        }
#undef AF_EXCHANGE
#undef AF_COMBINE
        if (PJ == 1 && FU == 0) {                                            // the projection, fp32, in accumulator layout
#pragma unroll
            for (int jj = 0; jj < NT; ++jj) {
                if (!AF_OWN(jj)) continue;
                f32x4* dst = reinterpret_cast<f32x4*>(A.pbuf) + ((((size_t)qpos * nso + ctg) * 4 + tb + ps * NT + ((jj + rot) % NT)) * 4) * 64 + lane;
                if (!(A.abl & 2)) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        f32x4 v = {pac[jj][4 * q] * A.inv_scale_p, pac[jj][4 * q + 1] * A.inv_scale_p, pac[jj][4 * q + 2] * A.inv_scale_p, pac[jj][4 * q + 3] * A.inv_scale_p};
                        dst[q * 64] = v;
                    }
                }
            }
        }

        if (FU > 0 && KS == 2) {
            // the activation slab aliases the exchange scratch: every wave must have read its partner's partial sums before anyone
            // writes activations there, and the slab's pad units (a window's units left and right of the board) are zero again
            __builtin_amdgcn_s_barrier();
            for (uint32_t u = threadIdx.x; u < 8u * (G::WINU - (uint32_t)G::NPIX); u += 256) {
                const uint32_t row = u / (G::WINU - (uint32_t)G::NPIX), k = u % (G::WINU - (uint32_t)G::NPIX);
                const uint32_t unit = k < (uint32_t)G::POFF ? k : k + (uint32_t)G::NPIX;
                *reinterpret_cast<uint4*>(smem + kGOff + row * kRowL + unit * 16u) = uint4{0, 0, 0, 0};
            }
        }
        AF_T(tp2);
        // epilogue: scale back, bias, ELU; split into halves and store (or fp32 planes for the heads).  The weight
        // rows are packed so that a lane's 16 accumulator rows are the couts 32*ctg + 16*kg + r.
        float bs[16];
        {
            const f32x4* bp = reinterpret_cast<const f32x4*>(smem + kBiasOff + (uint32_t)(32 * ct + 16 * kg) * 4u);
#pragma unroll
            for (int q = 0; q < 4; ++q) { const f32x4 v = bp[q]; bs[4 * q] = v[0]; bs[4 * q + 1] = v[1]; bs[4 * q + 2] = v[2]; bs[4 * q + 3] = v[3]; }
        }
#pragma unroll
        for (int jj = 0; jj < NT; ++jj) {
            if (!AF_OWN(jj)) continue;
            float v[16];
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                f32x2 pre = f32x2{acc[jj][r], acc[jj][r + 1]} * A.inv_scale + f32x2{bs[r], bs[r + 1]};
                if (PJ == 2) pre += f32x2{padd[jj % NFIN][r / 4][r % 4], padd[jj % NFIN][r / 4][r % 4 + 1]};   // (a wave's own tiles are its slots 0 .. NFIN-1)
                const f32x2 y = elu2(pre);
                v[r] = y.x; v[r + 1] = y.y;
            }
            if (HD > 0) {
                head_tile(v, jj, pos);
            } else if (OUT32) {
                const int y = pix[jj] / G::S, x = pix[jj] - y * G::S;
                float* o = A.out32 + ((size_t)pos * (nso * 32) + 32 * ctg + 16 * kg) * A.PP + (y + 1) * A.WP + x + 1;
                if (ok[jj] && !(A.abl & 2)) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[(size_t)r * A.PP] = v[r];
                }
            } else {
                char* o = A.out + ((size_t)((A.abl & 8) ? (pos & 127) : pos) * nso + ctg) * kSlabH + (uint32_t)(2 * kg) * kRowH + (uint32_t)(pix[jj] + G::POFF) * 16u;
                // fused block: the tile goes into the LDS activation slab instead (ring-slot layout: [hi|lo][4 unit rows][WINU units])
                char* og = smem + kGOff + (uint32_t)(2 * kg) * kRowL + (uint32_t)(pix[jj] + G::POFF) * 16u;
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {
                    h8 hi, lo;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float f = v[8 * hf + e];
                        const _Float16 h = (_Float16)f;
                        hi[e] = h;
                        lo[e] = (_Float16)(f - (float)h);
                    }
                    if (FU > 0) {
                        if (ok[jj]) {
                            *reinterpret_cast<h8*>(og + hf * kRowL) = hi;
                            *reinterpret_cast<h8*>(og + hf * kRowL + kHalfL) = lo;
                        }
                    } else if (ok[jj] && !(A.abl & 2)) {      // (tile 3 always has valid lanes: the two stores are always issued)
                        if (DF != 0) { st16_sc1(o + hf * kRowH, hi); st16_sc1(o + hf * kRowH + kHalfH, lo); }
                        else { st16(o + hf * kRowH, hi); st16(o + hf * kRowH + kHalfH, lo); }
                    }
                }
            }
        }
        if constexpr (FU > 0) {
            // ---- the block's second convolution (32 -> 32, one slab, K = 2 k-steps x 9 taps) on the slab just written ----
            // this wave's tile: its slot 0 (tile index = wave index, as in af_conv_f16s<G,1,0,1,1,4,...> where ps = wave)
            float pj[16];                                     // the block's projection, scaled as the pbuf hand-off scales it
#pragma unroll
            for (int r = 0; r < 16; ++r) { pj[r] = pac[0][r] * A.inv_scale_p; asm volatile("" : "+v"(pj[r])); }   // (a product, then a plain add: no fma)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                     // every wave's tile is in the slab
            const uint32_t gb = kGOff + (uint32_t)kg * kRowL + (uint32_t)((pix[0] + G::POFF - G::S - 1) * 16);
            const uint32_t gz = kZoff + (gb & 255u);
            const uint32_t c2 = (AF_F16S_ZPAD && !ok[0]) ? gz : gb, l2 = edgeL[0] ? gz : c2, r2 = edgeR[0] ? gz : c2;
            auto rd2 = [&](int it, int p) -> h8 {
                const int c = it / 9, tap = it % 9, ky = tap / 3, kx = tap % 3;
                const uint32_t base = kx == 0 ? l2 : (kx == 2 ? r2 : c2);
                return *reinterpret_cast<const h8*>(smem + base + (uint32_t)p * kHalfL + 2u * c * kRowL + (uint32_t)(ky * G::S + kx) * 16u);
            };
            auto wa2 = [&](int it, int hl) -> h8 { return *reinterpret_cast<const h8*>(smem + kW2Off + (uint32_t)((2 * it + hl) * 64 + lane) * 16u); };
            f32x16 a2, x2;
#pragma unroll
            for (int r = 0; r < 16; ++r) { a2[r] = 0.0f; x2[r] = 0.0f; }
            h8 f2[2][2], w2f[2][2];
            f2[0][0] = rd2(0, 0); f2[0][1] = rd2(0, 1); w2f[0][0] = wa2(0, 0); w2f[0][1] = wa2(0, 1);
#pragma clang loop unroll(full)
            for (int it = 0; it < 18; ++it) {
                const int b = it & 1;
                if (it + 1 < 18) {
                    f2[b ^ 1][0] = rd2(it + 1, 0); f2[b ^ 1][1] = rd2(it + 1, 1);
                    w2f[b ^ 1][0] = wa2(it + 1, 0); w2f[b ^ 1][1] = wa2(it + 1, 1);
                }
                a2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(w2f[b][0], f2[b][0], a2, 0, 0, 0);
                x2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(w2f[b][0], f2[b][1], x2, 0, 0, 0);
                x2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(w2f[b][1], f2[b][0], x2, 0, 0, 0);
            }
            __builtin_amdgcn_s_barrier();                     // every wave is done reading the slab (the next position's epilogue / exchange writes it)
#pragma unroll
            for (int r = 0; r < 16; ++r) a2[r] += x2[r];
            float v2[16], bs2[16];
            {
                const f32x4* bp = reinterpret_cast<const f32x4*>(smem + kBiasOff + 256u + (uint32_t)(16 * kg) * 4u);
#pragma unroll
                for (int q = 0; q < 4; ++q) { const f32x4 v = bp[q]; bs2[4 * q] = v[0]; bs2[4 * q + 1] = v[1]; bs2[4 * q + 2] = v[2]; bs2[4 * q + 3] = v[3]; }
            }
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                f32x2 pre = f32x2{a2[r], a2[r + 1]} * A.inv_scale2 + f32x2{bs2[r], bs2[r + 1]};
                pre += f32x2{pj[r], pj[r + 1]};
                const f32x2 y = elu2(pre);
                v2[r] = y.x; v2[r + 1] = y.y;
            }
            head_tile(v2, 0, pos);
        }
#ifdef AF_F16S_TIMING
        {
            AF_T(tp3);
            tacc[3] += tp2 - tp1; tacc[4] += tp3 - tp2; tacc[5] += tp3 - tp0; tacc[6] += 1;
        }
#endif
    }
    if (DF != 0) df_signal(A.df_done + q0);
#ifdef AF_F16S_TIMING
    if (lane == 0) {
        tacc[0] = tacc[5] - tacc[1] - tacc[2] - tacc[3] - tacc[4];
        const int layer = (A.abl >> 8) & 15;
        for (int q = 0; q < 8; ++q) g_f16s_cycles[layer][blockIdx.x + 256 * blockIdx.y][wv][q] = tacc[q];
        g_f16s_cycles[layer][blockIdx.x + 256 * blockIdx.y][wv][8] = t_ready - t_entry;
        if (wv == 0) { g_f16s_wall[layer][blockIdx.x + 256 * blockIdx.y][0] = wall_entry; g_f16s_wall[layer][blockIdx.x + 256 * blockIdx.y][1] = wall_clock64(); }
    }
#endif
#undef AF_OWN
}

// DIST: prefetch distance of this instantiation (default kDist); WPE: waves per SIMD = workgroups per CU.  WPE = 2 (r3) is for the
// narrow layers whose whole register need fits 256: with DIST = 1 the ring is 3 slots (74 KB per workgroup), two workgroups share
// a CU and one's epilogue / barriers / waits run under the other's MFMAs.
template <class G, int NSM, int NSP, int CT, int KS, int PS, bool OUT32, bool XACC, int PJ, int HD, int DIST = kDist, int WPE = 1>
__global__ __launch_bounds__(256, WPE) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void af_conv_f16s(F16sArgs A) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    f16s_body<G, NSM, NSP, CT, KS, PS, OUT32, XACC, PJ, HD, DIST, WPE, 4 / PS, -1>(A, smem, (int)blockIdx.x, (int)gridDim.x);
}

// A whole residual block whose second convolution is 32 -> 32 (blocks 3 and 5) in one kernel: see "fused block" in f16s_body.
template <class G, int NSM, int KS, int PS, bool XACC, int FU, int DIST>
__global__ __launch_bounds__(256, 1) __attribute__((amdgpu_waves_per_eu(1, 1))) void af_block_f16s(F16sArgs A) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    f16s_body<G, NSM, 0, 1, KS, PS, false, XACC, 1, 0, DIST, 1, 4 / PS, -1, FU>(A, smem, (int)blockIdx.x, (int)gridDim.x);
}

// Small batches, 11x11: the first convolution of the policy branch (128 -> 64, pixel tiles split over two workgroups per position) and the
// whole value block (af_block_f16s's body) depend on the trunk only — one launch runs both as two workgroup classes, so that the value
// branch needs no second stream: at batch 1 the fork and the join of the side stream were 12 + 10 us of a 124-us tick
// (profiles/r5_22_selfplay_timeline.txt), more than the value branch's own kernels take.
template <class G>
__global__ __launch_bounds__(256, 1) __attribute__((amdgpu_waves_per_eu(1, 1))) void af_small_pair_f16s(F16sArgs P, F16sArgs V) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int nb = P.batch, id = (int)blockIdx.x;
    if (id < 2 * nb) f16s_body<G, 4, 0, 2, 2, 1, false, false, 0, 0, kDist, 1, 2, -1>(P, smem, id % nb, nb, (id / nb) * 2);
    else f16s_body<G, 4, 0, 1, 2, 2, false, false, 1, 0, 2, 1, 2, -1, 1>(V, smem, id - 2 * nb, nb);
}

// ... and the same pairing at full batch (r5): the first P.gx0 workgroups are the policy branch's first convolution (one per CU, walking the
// positions), the rest the value block; the hardware places workgroups in index order, so the value block's workgroups start on CUs as
// the policy layer's finish — the tail filling the side stream was kept for, without its fork and join.
template <class G>
__global__ __launch_bounds__(256, 1) __attribute__((amdgpu_waves_per_eu(1, 1))) void af_big_pair_f16s(F16sArgs P, F16sArgs V) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int gp = P.gx0, id = (int)blockIdx.x;
    if (id < gp) f16s_body<G, 4, 0, 2, 2, 1, false, false, 0, 0, kDist, 1, 4, -1>(P, smem, id, gp);
    else f16s_body<G, 4, 0, 1, 2, 2, false, false, 1, 0, 2, 1, 2, -1, 1>(V, smem, id - gp, (int)gridDim.x - gp);
}

// Small batches (r5; the drop-in Player evaluates ONE leaf per simulation): with <= 8 positions a launch is one workgroup per position
// that works through all four pixel tiles — 14-16 k cycles of MFMAs in the wide layers next to an 8 k-cycle start-up
// (profiles/r5_14_f16s_phases_batch1.txt).  Here gridDim.z workgroups share a pseudo-position, NTW tiles per wave each (1 where every
// wave holds its whole K, 2 under a k-split: each wave of a pair finishes one).  Per tile the same MFMAs in the same order: bit-identical.
template <class G, int NSM, int NSP, int CT, int KS, int PS, bool OUT32, bool XACC, int PJ, int HD, int DIST, int WPE, int NTW>
__global__ __launch_bounds__(256, WPE) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void af_conv_f16s_sb(F16sArgs A) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    f16s_body<G, NSM, NSP, CT, KS, PS, OUT32, XACC, PJ, HD, DIST, WPE, NTW, -1>(A, smem, (int)blockIdx.x, (int)gridDim.x, (int)blockIdx.z * PS * NTW);
}

// 15x15, the layers whose waves hold all four pixel tiles of a pseudo-position (PS = 1: 71 % of that forward): 225 pixels are seven
// tiles and ONE pixel, and with two 4-tile halves the second half's fourth tile multiplies 31 zero columns out of 32 (88 % tile
// efficiency).  Here the first A.gx0 workgroups of the launch take half 0 of every board with 4 tiles and the rest
// take half 1 with 3 (gx0 : gx1 = 9 : 7 ~ 4 : 3, both multiples of 8 so that every XCD gets its share of each class and the
// XCD-contiguous position map holds inside a class); the corner pixel is computed by af_corner_f16s, batched over 32 boards.
template <class G, int NSM, int NSP, int CT, int KS, bool XACC>
__global__ __launch_bounds__(256, 1) __attribute__((amdgpu_waves_per_eu(1, 1))) void af_conv_f16s_h15(F16sArgs A) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int gx0 = A.gx0;
    if ((int)blockIdx.x < gx0) f16s_body<G, NSM, NSP, CT, KS, 1, false, XACC, 0, 0, kDist, 1, 4, 0>(A, smem, (int)blockIdx.x, gx0);
    else f16s_body<G, NSM, NSP, CT, KS, 1, false, XACC, 0, 0, kDist, 1, 3, 1>(A, smem, (int)blockIdx.x - gx0, (int)gridDim.x - gx0);
}

// Pixel 224 (the corner (14, 14)) of a 15x15 board for the layers that run af_conv_f16s_h15: one MFMA pixel tile = the corner pixels
// of 32 BOARDS.  Of the 3x3 taps only (0,0), (0,1), (1,0), (1,1) lie on the board (pixels 208, 209, 223, 224); a lane's B fragment is
// one 16-byte unit of ITS board, read from the compact copy the half-1 workgroups of af_conv_f16s_h15 made while the slab was in LDS
// (512 contiguous bytes per slab and board: 9 MB per layer and 4096 boards, L2 / Infinity-Cache resident; straight from the S32 tensor
// the same units cost three 128-byte lines per 64 useful bytes = 50 MB and 25-45 us per launch: profiles/r5_09); the A fragments are
// the layer's own packed weights (pack_layer: [cout tile][ks][item][hi|lo]).
// A workgroup = 32 boards x up to 4 cout tiles (one per wave).
// BIT-IDENTICAL to what the two-halves launch (af_conv_f16s on half 1, tile 3, lane 0) computes for this pixel — small batches still
// take that launch, and a board's result must not depend on the batch it is evaluated in: the same MFMAs on the same operands in the
// same order per accumulator (slabs in stream order, k-step, tap; hi x hi, hi x lo, lo x hi; XACC: cross terms in their own accumulator,
// added at the end; KS = 2: one accumulator per k-half, summed like the exchange does), minus the MFMAs whose B column is all zero
// for this pixel (taps off the board: they add +0), and the same epilogue expression.
struct CornerArgs {
    const char* stash;    // the corner's operands: [board][slab][hi|lo][4 unit rows][4 pixels] units (written by af_conv_f16s_h15)
    const uint4* w;       // the layer's packed A fragments
    const float* bias;
    char* out;            // S32, NCT slabs per board
    float inv_scale;
    int batch, NCT;
};
template <class G, int KS, bool XACC, int NSM, int NSP>
__global__ __launch_bounds__(256) void af_corner_f16s(CornerArgs A) {
    using L = Lay<G>;
    constexpr int CPIX = G::NPIX - 1;                                          // 224
    constexpr int C16 = 2 / KS, NIT = NSP * C16 + NSM * 9 * C16;
    constexpr int NST = NSM * 8 + NSP * 2;                                     // steps (k-step x on-board tap): three MFMAs each
    constexpr int D = NST < 12 ? NST : 12;                                     // operand ring: the loads of step t + D are issued behind step t
    const int lane = threadIdx.x & 63, kg = lane >> 5, nn = lane & 31;
    const int tile = (int)blockIdx.y * 4 + (int)(threadIdx.x >> 6);
    if (tile >= A.NCT) return;
    const int p = 32 * (int)blockIdx.x + nn;
    const bool valid = p < A.batch;
    const size_t pc = (size_t)(valid ? p : A.batch - 1);
    f32x16 acc[KS], acx[XACC ? KS : 1];
#pragma unroll
    for (int h = 0; h < KS; ++h)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[h][r] = 0.0f; if (XACC) acx[h][r] = 0.0f; }
    // (the corner kernel restates the main kernel's slab order: 3x3 slabs, then the projection's)
    // step t -> (k-half, A fragment, B unit): the main kernel's order (slab, k-step, tap; the projection's slabs last)
    const uint4* const wbase = A.w + (size_t)tile * KS * NIT * 2 * 64 + lane;
    // B operands: the compact copy the half-1 workgroups of af_conv_f16s_h15 made while the slabs were in LDS:
    // [board][slab of the stream][hi|lo][unit row][pixel 208, 209, 223, 224] x 16 bytes
    constexpr int SPP = NSM + NSP;
    const char* const st0 = A.stash + (pc * SPP * 32 + (uint32_t)kg * 4u) * 16;
    struct Ops { uint4 a0, a1, b0, b1; };
    auto half_of = [](int t) constexpr -> int {
        const int k16 = t < NSM * 8 ? (t >> 2) & 1 : (t - NSM * 8) & 1;
        return KS == 2 ? k16 : 0;
    };
    auto fetch = [&](int t) -> Ops {
        int k16, item;
        const char* bsrc;
        if (t < NSM * 8) {
            const int s_ = t >> 3, tq = t & 3, ky = tq >> 1, kx = tq & 1;
            k16 = (t >> 2) & 1;
            item = NSP * C16 + s_ * 9 * C16 + (KS == 2 ? 0 : k16) * 9 + 3 * ky + kx;
            bsrc = st0 + (uint32_t)((s_ * 32 + 2 * k16 * 4 + tq) * 16);
        } else {
            const int u = t - NSM * 8, s_ = u >> 1;
            k16 = u & 1;
            item = s_ * C16 + (KS == 2 ? 0 : k16);
            bsrc = st0 + (uint32_t)(((NSM + s_) * 32 + 2 * k16 * 4 + 3) * 16);
        }
        const uint4* wp = wbase + ((size_t)((KS == 2 ? k16 : 0) * NIT + item) * 2) * 64;
        Ops o;
        o.a0 = wp[0]; o.a1 = wp[64];
        o.b0 = *reinterpret_cast<const uint4*>(bsrc); o.b1 = *reinterpret_cast<const uint4*>(bsrc + 256);
        return o;
    };
    Ops ring[D];
#pragma unroll
    for (int t = 0; t < D; ++t) ring[t] = fetch(t);
    __builtin_amdgcn_sched_barrier(0);                                         // all D steps' loads in flight before the first MFMA
#pragma unroll
    for (int t = 0; t < NST; ++t) {
        const Ops o = ring[t % D];
        if (t + D < NST) ring[t % D] = fetch(t + D);
        const int h = half_of(t);
        h8 wh, wl, bh, bl;
        __builtin_memcpy(&wh, &o.a0, 16); __builtin_memcpy(&wl, &o.a1, 16); __builtin_memcpy(&bh, &o.b0, 16); __builtin_memcpy(&bl, &o.b1, 16);
        acc[h] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, bh, acc[h], 0, 0, 0);
        if (XACC) {
            acx[h] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, bl, acx[h], 0, 0, 0);
            acx[h] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, bh, acx[h], 0, 0, 0);
        } else {
            acc[h] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, bl, acc[h], 0, 0, 0);
            acc[h] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, bh, acc[h], 0, 0, 0);
        }
        // keep the ring a ring: hipcc otherwise sinks every load to its use (54 registers, one memory round trip per step)
        __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);                     // 3 MFMA
        __builtin_amdgcn_sched_group_barrier(0x020, 4, 0);                     // 4 VMEM reads (step t + D)
    }
    if (XACC) {
#pragma unroll
        for (int h = 0; h < KS; ++h)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[h][r] += acx[h][r];
    }
    if (KS == 2) {                                                             // the exchange: the finishing wave adds its partner's partial sum
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][r] = acc[KS - 1][r] + acc[0][r];
    }
    if (!valid) return;
    // epilogue as in the main kernel: a lane's 16 accumulator rows are the couts 32 tile + 16 kg + r
    float v[16];
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
        const f32x2 pre = f32x2{acc[0][r], acc[0][r + 1]} * A.inv_scale + f32x2{A.bias[32 * tile + 16 * kg + r], A.bias[32 * tile + 16 * kg + r + 1]};
        const f32x2 y = elu2(pre);
        v[r] = y.x; v[r + 1] = y.y;
    }
    char* o = A.out + ((size_t)p * A.NCT + tile) * L::kSlabH + (uint32_t)(2 * kg) * L::kRowH + (uint32_t)(CPIX + G::POFF) * 16u;
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
        h8 hi, lo;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float f = v[8 * hf + e];
            const _Float16 h = (_Float16)f;
            hi[e] = h;
            lo[e] = (_Float16)(f - (float)h);
        }
        st16(o + hf * L::kRowH, hi);
        st16(o + hf * L::kRowH + L::kHalfH, lo);
    }
}

// 5x5 stem (3 -> 32, SAME) + bias + ELU (network.py:63) on the VALU (2,400 MAC per pixel), output split into S32 — any board
// geometry (the path of the 15x15 boards; 11x11 runs af_stem_mfma_f16s below).  planes fp32 [B][3][S][S] (utils.py:256
// board_to_inputs); w [75][32] HWIO.
template <class G>
__global__ __launch_bounds__(256) void af_stem_f16s(const float* __restrict__ planes, const float* __restrict__ w,
                                                    const float* __restrict__ bias, char* __restrict__ out, int batch) {
    constexpr int S = G::S, NPIX = G::NPIX, SP = S + 4;
    constexpr uint32_t kRowH = Lay<G>::kRowH, kHalfH = Lay<G>::kHalfH, kSlabH = Lay<G>::kSlabH;
    __shared__ float sx[3 * SP * SP];
    __shared__ float sw[75 * 32];
    __shared__ float sb[32];
    const int t = threadIdx.x;
    for (int i = t; i < 75 * 32; i += 256) sw[i] = w[i];
    if (t < 32) sb[t] = bias[t];
    for (int i = t; i < 3 * SP * SP; i += 256) sx[i] = 0.0f;
    for (int b = blockIdx.x; b < batch; b += gridDim.x) {
        __syncthreads();
        for (int i = t; i < 3 * NPIX; i += 256) {
            const int c = i / NPIX, p = i - c * NPIX, y = p / S, x = p - y * S;
            sx[c * SP * SP + (y + 2) * SP + x + 2] = planes[(size_t)b * 3 * NPIX + i];
        }
        __syncthreads();
        if (t < NPIX) {
            const int y = t / S, x = t - y * S;
            float acc[32];
#pragma unroll
            for (int co = 0; co < 32; ++co) acc[co] = sb[co];
            for (int ky = 0; ky < 5; ++ky)
                for (int kx = 0; kx < 5; ++kx)
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        const float xv = sx[c * SP * SP + (y + ky) * SP + x + kx];
                        const float* wr = sw + ((ky * 5 + kx) * 3 + c) * 32;
#pragma unroll
                        for (int co = 0; co < 32; ++co) acc[co] = fmaf(xv, wr[co], acc[co]);
                    }
            char* o = out + (size_t)b * kSlabH + (uint32_t)(t + G::POFF) * 16u;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                h8 hi, lo;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float f = elu1(acc[8 * u + e]);
                    const _Float16 h = (_Float16)f;
                    hi[e] = h;
                    lo[e] = (_Float16)(f - (float)h);
                }
                *reinterpret_cast<h8*>(o + u * kRowH) = hi;
                *reinterpret_cast<h8*>(o + u * kRowH + kHalfH) = lo;
            }
        }
    }
}

// The same stem on the matrix cores (default): K = 3 planes x 5 rows x (5 taps padded to 8) = 15 groups of 8 = 8 k-steps; a B
// fragment (pixel n, group (cin, ky)) is the 8-wide input window starting at column x-2 of row y+ky-2, pre-expanded into
// LDS once per position as 3 x 15 x 11 entries of 8 fp16 ("im2row"): one aligned ds_read_b128.  The planes are 0/1 —
// exact in fp16 — so only the weights are split: two MFMAs per k-step.  One wave = one pixel tile x the 32 couts.
template <class G> struct StemLds {       // LDS of the stem body: the im2row entries and the zero-bordered fp16 image
    static constexpr int SR = G::S + 4, IW = G::S + 5;
    static constexpr int kEnt = 3 * SR * G::S + 1, kImg = 3 * SR * IW;
    static constexpr uint32_t kImgOff = (uint32_t)((kEnt * 16 + 255) / 256 * 256), kBytes = kImgOff + (uint32_t)kImg * 2u;
};
// (the body: workgroup bx of gx walks the positions bx, bx + gx, ...; ent / img: StemLds<G>::kEnt uint4 and kImg fp16 of LDS)
template <class G, bool SC1 = false>
__device__ __forceinline__ void f16s_stem_body(const float* __restrict__ planes, const uint4* __restrict__ w /*[8][hi|lo][64]*/,
                                               const float* __restrict__ bias, float inv_scale, char* __restrict__ out, int batch,
                                               const int bx, const int gx, uint4* ent, _Float16* img) {
    constexpr int S = G::S, NPIX = G::NPIX, SR = S + 4, IW = S + 5;          // image: rows -2..S+1, IW columns (-2 .. S+2)
    constexpr int NPL = 3 * NPIX, NLD = (NPL + 255) / 256;                    // plane elements per position / per thread
    constexpr int NTL = (NPIX + 31) / 32, NRND = (NTL + 3) / 4;               // pixel tiles per position / tile rounds per wave
    constexpr uint32_t kRowH = Lay<G>::kRowH, kHalfH = Lay<G>::kHalfH, kSlabH = Lay<G>::kSlabH;
    // entry (cin, yy, x) of ent = window x-2..x+5 of row yy-2
    const int lane = threadIdx.x & 63, kg = lane >> 5, nn = lane & 31;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    h8 W[16];
#pragma unroll
    for (int f = 0; f < 16; ++f) {
        const uint4 v = w[f * 64 + lane];
        __builtin_memcpy(&W[f], &v, 16);
    }
    float bs[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) bs[r] = bias[16 * kg + r];
    // r3: the planes of a position go through a zero-bordered fp16 image in LDS (rows -2..S+1, columns -2..S+2), filled from
    // registers that were loaded one position ahead, and the im2row entries are built from that image — the build used to read
    // the planes from global memory five scattered words per entry inside the loop (11x11: 44 -> 22 us per 4096 positions; same
    // values).  15x15 (r3): the same kernel, two rounds of four pixel tiles per wave (the VALU stem took 125 us).
    for (int i = threadIdx.x; i < 3 * SR * IW; i += 256) img[i] = (_Float16)0.0f;
    auto img_at = [](int i) -> int { const int c = i / NPIX, p = i - c * NPIX, y = p / S, x = p - y * S; return (c * SR + y + 2) * IW + x + 2; };
    int at[NLD];
    float nx[NLD];
#pragma unroll
    for (int q = 0; q < NLD; ++q) {
        const int i = (int)threadIdx.x + 256 * q;
        at[q] = img_at(i < NPL ? i : 0);
        nx[q] = 0.0f;
        if (i < NPL && bx < batch) nx[q] = planes[(size_t)bx * NPL + i];
    }
    for (int pos = bx; pos < batch; pos += gx) {
        __syncthreads();                                                     // the previous position's reads are done
#pragma unroll
        for (int q = 0; q < NLD; ++q)
            if ((int)threadIdx.x + 256 * q < NPL) img[at[q]] = (_Float16)nx[q];
        if (pos + gx < batch) {                                              // next position's planes: in flight under this one
            const float* pl = planes + (size_t)(pos + gx) * NPL;
#pragma unroll
            for (int q = 0; q < NLD; ++q)
                if ((int)threadIdx.x + 256 * q < NPL) nx[q] = pl[threadIdx.x + 256 * q];
        }
        __syncthreads();
        for (int en = threadIdx.x; en < 3 * SR * S; en += 256) {
            const int cin = en / (SR * S), rem = en - cin * SR * S, yy = rem / S, x = rem - yy * S;
            const _Float16* src = img + (cin * SR + yy) * IW + x;            // padded row yy = board row yy - 2, columns x-2 .. x+2
            const h8 v = {src[0], src[1], src[2], src[3], src[4], 0, 0, 0};
            __builtin_memcpy(&ent[en], &v, 16);
        }
        __syncthreads();
#pragma unroll
        for (int rd_ = 0; rd_ < NRND; ++rd_) {
            const int n = 32 * (wv + 4 * rd_) + nn;
            const bool ok = n < NPIX;
            const int nc = ok ? n : 0;
            if (32 * (wv + 4 * rd_) >= NPIX) continue;                       // (wave-uniform: this round has no tile for this wave)
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
            for (int s_ = 0; s_ < 8; ++s_) {
                // group g = 2 s_ + kg -> (cin, ky); g = 15 is the zero pad of K (its weights are zero: any entry will do)
                const int ga = 2 * s_, gb = 2 * s_ + 1 < 15 ? 2 * s_ + 1 : 0;
                const uint32_t offa = (uint32_t)((ga / 5) * SR + ga % 5) * S, offb = (uint32_t)((gb / 5) * SR + gb % 5) * S;
                h8 b;
                __builtin_memcpy(&b, &ent[(uint32_t)nc + (kg ? offb : offa)], 16);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(W[2 * s_], b, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(W[2 * s_ + 1], b, acc, 0, 0, 0);
            }
            char* o = out + (size_t)pos * kSlabH + (uint32_t)(2 * kg) * kRowH + (uint32_t)(nc + G::POFF) * 16u;
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                h8 hi, lo;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float f = elu1(acc[8 * hf + e] * inv_scale + bs[8 * hf + e]);
                    const _Float16 h = (_Float16)f;
                    hi[e] = h;
                    lo[e] = (_Float16)(f - (float)h);
                }
                if (ok) {
                    if (SC1) { st16_sc1(o + hf * kRowH, hi); st16_sc1(o + hf * kRowH + kHalfH, lo); }
                    else {
                        *reinterpret_cast<h8*>(o + hf * kRowH) = hi;
                        *reinterpret_cast<h8*>(o + hf * kRowH + kHalfH) = lo;
                    }
                }
            }
        }
    }
}

template <class G>
__global__ __launch_bounds__(256) void af_stem_mfma_f16s(const float* __restrict__ planes, const uint4* __restrict__ w, const float* __restrict__ bias,
                                                         float inv_scale, char* __restrict__ out, int batch) {
    __shared__ __attribute__((aligned(16))) uint4 ent[StemLds<G>::kEnt];
    __shared__ _Float16 img[StemLds<G>::kImg];
    f16s_stem_body<G>(planes, w, bias, inv_scale, out, batch, (int)blockIdx.x, (int)gridDim.x, ent, img);
}

// ----------------------------------------------------------------------------------------------------------------------------
// The dense layers of the two heads on the same split-operand MFMA (network.py:70-76 value, :82-88 policy).  Their inputs
// come from the fused 1x1 convolution of the branch's last conv kernel (HD above), already split into halves
// (HXP = 128 x HALVES pixel slots per position; the slots past S*S stay zero):
//   value   xv [position][hi|lo][HXP pixels][4 channels]   fp16 (2 KB / position at 11x11, 4 KB at 15x15)
//   policy  xp [position][hi|lo][HXP pixels][16 channels]  fp16 (8 KB / 16 KB)
// and a k-step of 16 is chosen so that 16 contiguous bytes of a position ARE a lane's B operand: value k = 4 pixels x 4
// channels, policy k = the 16 channels of one pixel.  The dense weights are streamed from L2 as pre-packed A fragments.
// ----------------------------------------------------------------------------------------------------------------------------
// value: fc1 4*S*S -> 64 + ELU, fc2 64 -> 1, tanh(x/2).  Workgroup = 32 positions x 2 waves (cout tile mt of 32).  The whole job
// is 93 (171 at 15x15) MFMAs per wave: what matters is that a wave's operand loads are all in flight at once (batches of 16 k-steps).
// (the body: a workgroup of 128 threads — or of 256 whose upper half repeats the lower half's arithmetic and stores nothing, so that
// the small-batch launch can run it as one workgroup class of a 256-thread kernel: af_small_l7v_f16s)
template <class G>
__device__ __forceinline__ void f16s_vfc_body(const char* __restrict__ xv, const uint4* __restrict__ a, const float* __restrict__ b1,
                                              const float* __restrict__ w2, const float* __restrict__ b2, float inv_scale,
                                              float* __restrict__ value, int batch, const int bx) {
    __shared__ float red[2][32];
    constexpr int NS = Hx<G>::VSTEPS;
    const bool act = threadIdx.x < 128;
    const int lane = threadIdx.x & 63, mt = (threadIdx.x >> 6) & 1, n = lane & 31, kg = lane >> 5;
    const int pos = bx * 32 + n, posc = pos < batch ? pos : batch - 1;
    const char* xb = xv + (size_t)posc * Hx<G>::kValPos + (uint32_t)kg * 16u;
    const uint4* ap = a + (size_t)mt * 128 + lane;                          // + step * 256 + half * 64
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
    for (int half = 0; half < (NS + 15) / 16; ++half) {
        uint4 bh[16], bl[16], ah[16], al[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int s_ = 16 * half + q;
            if (s_ < NS) {
                bh[q] = *reinterpret_cast<const uint4*>(xb + s_ * 32);
                bl[q] = *reinterpret_cast<const uint4*>(xb + Hx<G>::kValLo + s_ * 32);
                ah[q] = ap[(size_t)s_ * 256];
                al[q] = ap[(size_t)s_ * 256 + 64];
            }
        }
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
        for (int q = 0; q < 16; ++q) {                          // every load above is issued before the first MFMA waits
            if (16 * half + q < NS) {
                u32x4 t0, t1, t2, t3;
                __builtin_memcpy(&t0, &bh[q], 16); __builtin_memcpy(&t1, &bl[q], 16); __builtin_memcpy(&t2, &ah[q], 16); __builtin_memcpy(&t3, &al[q], 16);
                asm volatile("" : "+v"(t0), "+v"(t1), "+v"(t2), "+v"(t3));
                __builtin_memcpy(&bh[q], &t0, 16); __builtin_memcpy(&bl[q], &t1, 16); __builtin_memcpy(&ah[q], &t2, 16); __builtin_memcpy(&al[q], &t3, 16);
            }
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            if (16 * half + q < NS) {
                h8 xh, xl, wh, wl;
                __builtin_memcpy(&xh, &bh[q], 16); __builtin_memcpy(&xl, &bl[q], 16);
                __builtin_memcpy(&wh, &ah[q], 16); __builtin_memcpy(&wl, &al[q], 16);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xh, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xl, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, xh, acc, 0, 0, 0);
            }
        }
    }
    float part = 0.0f;                                   // a lane's 16 accumulator rows = outputs 32 mt + 16 kg + r
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int j = 32 * mt + 16 * kg + r;
        part += elu1(acc[r] * inv_scale + b1[j]) * w2[j];
    }
    part += __shfl_xor(part, 32);
    if (act && kg == 0) red[mt][n] = part;
    __syncthreads();
    if (act && mt == 0 && kg == 0 && pos < batch) value[pos] = tanhf((red[0][n] + red[1][n] + b2[0]) * 0.5f);
}
template <class G>
__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(1, 2))) void af_value_fc_f16s(
    const char* __restrict__ xv, const uint4* __restrict__ a /*[VSTEPS][2][hi|lo][64]*/, const float* __restrict__ b1, const float* __restrict__ w2,
    const float* __restrict__ b2, float inv_scale, float* __restrict__ value, int batch) {
    f16s_vfc_body<G>(xv, a, b1, w2, b2, inv_scale, value, batch, (int)blockIdx.x);
}
// Small batches, 11x11: the policy branch's second convolution (64 -> 64 + projection, tiles split over two workgroups per position)
// next to the value head's dense layers (one more workgroup; both depend on af_small_pair_f16s only) in one launch
struct VfcArgs { const char* xv; const uint4* a; const float *b1, *w2, *b2; float inv_scale; float* value; };
template <class G>
__global__ __launch_bounds__(256, 1) __attribute__((amdgpu_waves_per_eu(1, 1))) void af_small_l7v_f16s(F16sArgs P, VfcArgs V) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int nb = P.batch, id = (int)blockIdx.x;
    if (id < 2 * nb) f16s_body<G, 2, 4, 2, 2, 1, false, true, 0, 0, kDist, 1, 2, -1>(P, smem, id % nb, nb, (id / nb) * 2);
    else f16s_vfc_body<G>(V.xv, V.a, V.b1, V.w2, V.b2, V.inv_scale, V.value, nb, id - 2 * nb);
}
// ... and at full batch: P.gx0 workgroups of the policy branch's second convolution, then one workgroup per 32 positions of the value head
template <class G>
__global__ __launch_bounds__(256, 1) __attribute__((amdgpu_waves_per_eu(1, 1))) void af_big_l7v_f16s(F16sArgs P, VfcArgs V) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int gp = P.gx0, id = (int)blockIdx.x;
    if (id < gp) f16s_body<G, 2, 4, 2, 2, 1, false, true, 0, 0, kDist, 1, 4, -1>(P, smem, id, gp);
    else f16s_vfc_body<G>(V.xv, V.a, V.b1, V.w2, V.b2, V.inv_scale, V.value, P.batch, id - gp);
}

// ----------------------------------------------------------------------------------------------------------------------------
// The whole small-batch forward (<= 8 positions, 11x11) up to the policy head's input as ONE launch (r6; VERDICT r5 item 6: the drop-in
// Player evaluates one leaf per simulation, and its tick was 9 dependent launches whose per-launch start-up — weights into 512 registers
// per wave out of L2, LDS zeroing, ramp — cost more than their MFMAs: 93 us per tick, profiles/r5_20).  Every workgroup of the nine
// launches becomes a ROLE of this one, indexed in dependency order (stem, block 1, block 2, {policy conv1 | value block}, {policy conv2 |
// value dense}, block 5): all roles are dispatched at once, load their weights at once, and a role starts its slab stream when the
// roles that write its input have signalled (F16sArgs::df_*; per-position counters in device memory, agent-scope release / acquire
// because producer and consumer may sit on different XCDs).  A role only ever waits for roles of LOWER workgroup index, and
// workgroups are dispatched in index order, so the launch cannot deadlock whatever else occupies the device; the waits are bounded
// all the same (df_err).  The role bodies are the multi-launch path's bodies with the same template arguments, tile bases and
// operands: the same MFMAs in the same order — bit-identical outputs (tests/test_gpu_net.py, af_net_tune(7, 2048) = the launches).
// The last workgroup to finish zeroes the counters for the next launch.
// ----------------------------------------------------------------------------------------------------------------------------
constexpr int kDfStem = 0, kDfL0 = 8, kDfL1 = 16, kDfL2 = 24, kDfL3 = 32, kDfP = 40, kDfV = 48, kDfQ = 56, kDfB5 = 64, kDfPF = 72, kDfFin = 73, kDfErr = 74, kDfWords = 75;
struct SmallFwdArgs {
    const float* planes; const uint4* stem_w; const float* stem_b; float stem_inv; char* f0;
    F16sArgs L0, L1, L2, L3, P, V, Q, B5;
    VfcArgs F;
    // the policy head's dense layer as two roles (K halves): input, packed A fragments, bias, 1 / weight scale, output, and the
    // fp32 scratch through which K half 1 hands its partial sums to K half 0 ([tile 4][float4 4][lane 64])
    const char* xp; const uint4* pfa; const float* pfb; float pf_inv; float* policy; float* pfx;
    int* df;
    int batch;
};

// The policy head's dense layer + softmax (af_policy_fc_f16s) for <= 8 positions as two ROLES of the single-launch forward: role kh =
// K half kh (pixels 64 kh .. 64 kh + 63), wave mt = logit tile mt — the batched kernel's eight waves, four per workgroup.  What a role
// does BEFORE its input exists is the point: a wave's first 32 k-steps of dense weights (256 registers) are loaded while the trunk is
// still running, the other 32 stream in behind the MFMA chain.  Per (kh, mt) the same MFMAs on the same operands in the same order
// (step ascending; A_hi B_hi, A_hi B_lo, A_lo B_hi into one accumulator; steps past the board add exact zeros), K half 0 + K half 1,
// the same scale / bias expression and the same 16-threads-per-position softmax tree: bit-identical to af_policy_fc_f16s.
template <class G>
__device__ __forceinline__ void f16s_pfc_role(const SmallFwdArgs& S, const int kh, char* smem) {
    constexpr int NPIX = G::NPIX, HXP = Hx<G>::HXP, NLT = Hx<G>::NLT, LW = 32 * NLT, LWP = LW + 1, NS = HXP / 2, RD = 32;
    static_assert(NLT == 4 && NS == 64, "the role form of the policy head is written for 11x11 (one logit tile per wave)");
    const int nb = S.batch, lane = threadIdx.x & 63, n = lane & 31, kg = lane >> 5;
    const int mt = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    int* const df = S.df;
    const int step0 = NS * kh, nstep = NPIX - step0 < NS ? NPIX - step0 : NS;
    const uint4* ap = S.pfa + ((size_t)step0 * NLT + mt) * 128 + lane;          // + step * NLT * 128 (+ 64: the lo halves)
    uint4 ring[RD][2];
#pragma unroll
    for (int q = 0; q < RD; ++q) {
        const int st_ = q < nstep ? q : nstep - 1;
        ring[q][0] = ap[(size_t)st_ * NLT * 128];
        ring[q][1] = ap[(size_t)st_ * NLT * 128 + 64];
    }
    float bj[16];                       // this lane's 16 logit biases: loaded before the wait (the acquire below empties the caches)
#pragma unroll
    for (int r = 0; r < 16; ++r) { const int j = 32 * mt + 16 * kg + r; bj[r] = S.pfb[j < NPIX ? j : 0]; }
    // the weights of steps 32..63 cannot be held yet (the ring is full): touch them, so that the refills under the MFMA chain are L2
    // hits on this XCD instead of 256 KB per workgroup out of memory (one dword per lane and fragment: every line of the rows)
    {
        unsigned sink = 0;
#pragma unroll
        for (int q = RD; q < NS; ++q) {
            const int st_ = q < nstep ? q : nstep - 1;
            sink ^= reinterpret_cast<const unsigned*>(ap + (size_t)st_ * NLT * 128)[0] ^ reinterpret_cast<const unsigned*>(ap + (size_t)st_ * NLT * 128 + 64)[0];
        }
        asm volatile("" : : "v"(sink));
    }
    if (threadIdx.x == 0) {
        for (int p = 0; p < nb; ++p) df_poll(df + kDfB5 + p, 1, df + kDfErr);
    }
    DF_T(1);
    __syncthreads();
    // this K half of every position's input -> LDS ([position][hi|lo][64 pixels][16 channels] fp16, positions kXS bytes apart so that
    // the lanes of different positions fall into different banks) by agent-scope LDS-DMA (the producer stored write-through: no
    // invalidate, the weights stay where they are): 1 KB per instruction, 4 per position; then the B fragments are ds_read_b128 and the
    // only vector-memory traffic under the MFMA chain is the weight ring 32 steps ahead
    constexpr uint32_t kXS = 2u * NS * 32u + 16u;
    for (int c = mt; c < nb * 4; c += 4) {                                     // chunk c = (position, half, 1-KB part)
        const int pp = c >> 2, hl = (c >> 1) & 1, part = c & 1;
        glds16_sc1(S.xp + (size_t)pp * Hx<G>::kPolPos + (uint32_t)hl * Hx<G>::kPolLo + (uint32_t)step0 * 32u + (uint32_t)part * 1024u + (uint32_t)lane * 16u,
                   (uint32_t)__builtin_amdgcn_readfirstlane((int)((uint32_t)(uintptr_t)smem + (uint32_t)pp * kXS + (uint32_t)hl * (NS * 32u) + (uint32_t)part * 1024u)));
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const int pq = n < nb ? n : nb - 1;                                        // (column n of the MFMA = position n; the columns past the batch repeat the last one)
    const char* xb = smem + (uint32_t)pq * kXS + (uint32_t)kg * 16u;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    h8 bh = *reinterpret_cast<const h8*>(xb), bl = *reinterpret_cast<const h8*>(xb + NS * 32);
#pragma unroll
    for (int st_ = 0; st_ < NS; ++st_) {
        h8 ah, al, nbh = bh, nbl = bl;
        __builtin_memcpy(&ah, &ring[st_ % RD][0], 16); __builtin_memcpy(&al, &ring[st_ % RD][1], 16);
        if (st_ + 1 < NS) {                                                    // the next step's B fragments: their LDS latency sits under this step's MFMAs
            nbh = *reinterpret_cast<const h8*>(xb + (st_ + 1) * 32);
            nbl = *reinterpret_cast<const h8*>(xb + NS * 32 + (st_ + 1) * 32);
        }
        if (st_ + RD < NS) {
            const int pf = st_ + RD < nstep ? st_ + RD : nstep - 1;
            ring[st_ % RD][0] = ap[(size_t)pf * NLT * 128];
            ring[st_ % RD][1] = ap[(size_t)pf * NLT * 128 + 64];
        }
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc, 0, 0, 0);
        // keep the ring a ring: hipcc otherwise sinks every refill to its use 32 steps later (one memory round trip per step: the role
        // took 8.8 us instead of 4, profiles/r6_11)
        if (st_ + 1 < NS) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);  // 2 DS reads (the next step's B fragments)
        if (st_ + RD < NS) __builtin_amdgcn_sched_group_barrier(0x020, 2, 0); // 2 VMEM reads (step + RD's A fragments)
        __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);                     // 3 MFMA
        bh = nbh; bl = nbl;
    }
    __syncthreads();                                                           // (the staging area becomes the logit array below)
    f32x4* const px = reinterpret_cast<f32x4*>(S.pfx) + (size_t)mt * 4 * 64 + lane;
    if (kh == 1) {
#pragma unroll
        for (int q = 0; q < 4; ++q) st16_sc1(px + q * 64, f32x4{acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]});
        df_signal(df + kDfPF);
        return;
    }
    df_wait_inv(df + kDfPF, 1, df + kDfErr);
    float* Lg = reinterpret_cast<float*>(smem);                                // [32 positions][LWP]
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4 o = px[q * 64];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int r = 4 * q + e, j = 32 * mt + 16 * kg + r;
            Lg[n * LWP + j] = j < NPIX ? (acc[r] + o[e]) * S.pf_inv + bj[r] : -3.0e38f;
        }
    }
    __syncthreads();
    const int p = (int)threadIdx.x >> 4, sub = (int)threadIdx.x & 15;         // 16 threads per position (256 threads: positions 0..15 >= the batch)
    constexpr int NV = LW / 16;
    float v[NV], mx = -3.0e38f;
#pragma unroll
    for (int q = 0; q < NV; ++q) { v[q] = Lg[p * LWP + sub + 16 * q]; mx = fmaxf(mx, v[q]); }
#pragma unroll
    for (int off = 8; off >= 1; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
    float sum = 0.0f;
#pragma unroll
    for (int q = 0; q < NV; ++q) { v[q] = (sub + 16 * q) < NPIX ? expf(v[q] - mx) : 0.0f; sum += v[q]; }
#pragma unroll
    for (int off = 8; off >= 1; off >>= 1) sum += __shfl_xor(sum, off);
    if (p < nb) {
#pragma unroll
        for (int q = 0; q < NV; ++q) {
            const int j = sub + 16 * q;
            if (j < NPIX) S.policy[(size_t)p * NPIX + j] = v[q] / sum;
        }
    }
}

template <class G>
__global__ __launch_bounds__(256, 1) __attribute__((amdgpu_waves_per_eu(1, 1))) void af_small_forward_f16s(SmallFwdArgs S) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int nb = S.batch;
    int id = (int)blockIdx.x;
    int* const df = S.df;
    DF_T(0);
    if (id < nb) {
        DF_T(1);
        f16s_stem_body<G, true>(S.planes, S.stem_w, S.stem_b, S.stem_inv, S.f0, nb, id, nb, reinterpret_cast<uint4*>(smem),
                          reinterpret_cast<_Float16*>(smem + StemLds<G>::kImgOff));
        df_signal(df + kDfStem + id);
    } else if ((id -= nb) < 2 * nb) {                 // bone/block1 conv1: a position's two pixel-tile pairs on two workgroups
        f16s_body<G, 1, 0, 2, 1, 2, false, false, 0, 0, 1, 1, 1, -1, 0, 1>(S.L0, smem, id % nb, nb, (id / nb) * 2, 0, 1);
    } else if ((id -= 2 * nb) < 2 * nb) {             // bone/block1 conv2 + projection
        f16s_body<G, 2, 1, 2, 1, 2, false, true, 0, 0, kDist, 1, 1, -1, 0, 1>(S.L1, smem, id % nb, nb, (id / nb) * 2, 0, 1);
    } else if ((id -= 2 * nb) < 4 * nb) {             // bone/block2 conv1: one pixel tile per workgroup
        f16s_body<G, 2, 0, 4, 1, 1, false, false, 0, 0, kDist, 1, 1, -1, 0, 1>(S.L2, smem, id % nb, nb, id / nb, 0, 1);
    } else if ((id -= 4 * nb) < 4 * nb) {             // bone/block2 conv2 + projection: 2 cout-tile pairs x 2 pixel-tile pairs
        const int r = id / nb;
        f16s_body<G, 4, 2, 2, 2, 1, false, false, 0, 0, kDist, 1, 2, -1, 0, 1>(S.L3, smem, id % nb, nb, (r >> 1) * 2, r & 1, 2);
    } else if ((id -= 4 * nb) < 3 * nb) {             // policy/block4 conv1 (two workgroups) | the whole value block (one)
        if (id < 2 * nb) f16s_body<G, 4, 0, 2, 2, 1, false, false, 0, 0, kDist, 1, 2, -1, 0, 1>(S.P, smem, id % nb, nb, (id / nb) * 2, 0, 1);
        else f16s_body<G, 4, 0, 1, 2, 2, false, false, 1, 0, 2, 1, 2, -1, 1, 1>(S.V, smem, id - 2 * nb, nb, 0, 0, 1);
    } else if ((id -= 3 * nb) < 2 * nb + 1) {         // policy/block4 conv2 + projection (two workgroups) | the value head's dense layers (one for all)
        if (id < 2 * nb) {
            f16s_body<G, 2, 4, 2, 2, 1, false, true, 0, 0, kDist, 1, 2, -1, 0, 1>(S.Q, smem, id % nb, nb, (id / nb) * 2, 0, 1);
        } else {
            if (threadIdx.x == 0) {
                for (int p = 0; p < nb; ++p) df_poll(df + kDfV + p, 1, df + kDfErr);
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            }
            DF_T(1);
            __syncthreads();
            f16s_vfc_body<G>(S.F.xv, S.F.a, S.F.b1, S.F.w2, S.F.b2, S.F.inv_scale, S.F.value, nb, 0);
        }
    } else if ((id -= 2 * nb + 1) < nb) {             // policy/block5 (fused block + the policy head's 1x1 convolution)
        f16s_body<G, 2, 0, 1, 1, 4, false, true, 1, 0, 2, 1, 1, -1, 2, 1>(S.B5, smem, id, nb, 0, 0, 1);
    } else {                                          // the policy head's dense layer: K half 1, then K half 0 (which waits for it)
        f16s_pfc_role<G>(S, id - nb == 0 ? 1 : 0, smem);
    }
    __syncthreads();
    DF_T(4);
    if (threadIdx.x == 0) {
        if (__hip_atomic_fetch_add(df + kDfFin, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (int)gridDim.x - 1) {
            for (int i = 0; i <= kDfFin; ++i) __hip_atomic_store(df + i, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // (kDfErr stays: the host reads it)
        }
    }
}

// policy: fc 16*S*S -> S*S, softmax.  Workgroup = 32 positions x 8 waves = (MT tiles of 32 logits: mt, mt + 4, ...) x (half of K):
// two waves per SIMD, so that one wave's LDS / L2 latency sits under the other's MFMAs (the compiler would not keep the operand
// fetch of a single wave ahead of its MFMAs: 30 us; a fully unrolled, hand-prefetched single-wave variant was slower still).  The
// positions' inputs are staged through LDS 16 pixels at a time per K half (transposed to [half][pixel][channel octet][position]
// rows of 528 bytes: conflict-free ds_read_b128 B fragments), the A fragments ride a register ring of 16 / MT steps.
constexpr uint32_t kPfRow = 528u, kPfBuf = 64u * kPfRow;
#ifndef AF_PFC_POS
#define AF_PFC_POS 32
#endif
// positions per workgroup of af_policy_fc_f16s (A/B knob, r4_26): 32 = 128 workgroups 31.3 us, 16 = 256 workgroups 31.6 us, 8 = 512
// workgroups 54 us per 4096 positions — twice the workgroups stream twice the dense weights out of L2 in the same time: the kernel
// is bound by that stream, not by the number of CUs it runs on
constexpr int kPfPos = AF_PFC_POS;
constexpr uint32_t kPfLds = 4u * kPfBuf;                                    // [K half][double buffer]; later the logits [32][32 NLT] floats
#ifdef AF_PFC_TIMING
// profiling build only (tools/probe_pfc_timing.py): per wave, device-wide 100 MHz clock at entry | operands of the first chunk in place |
// K loop done | K halves combined | softmax + stores done
__device__ unsigned long long g_pfc_wall[256][8][5];
#define PF_T(i) if ((threadIdx.x & 63) == 0 && blockIdx.x < 256) g_pfc_wall[blockIdx.x][threadIdx.x >> 6][i] = wall_clock64()
#else
#define PF_T(i)
#endif
template <class G>
__global__ __launch_bounds__(512, 1) void af_policy_fc_f16s(const char* __restrict__ xp, const uint4* __restrict__ a /*[S*S][NLT][hi|lo][64]*/,
                                                            const float* __restrict__ bf, float inv_scale, float* __restrict__ policy, int batch) {
    extern __shared__ __attribute__((aligned(16))) char psm[];
    constexpr int NPIX = G::NPIX, HXP = Hx<G>::HXP, NLT = Hx<G>::NLT, MT = NLT / 4, RD = 16 / MT, NCH = HXP / 32, LW = 32 * NLT;
    // a position's logits in LDS are LWP = LW + 1 floats apart: with LW (a multiple of 32 banks) the 32 position lanes of a store
    // hit ONE bank — 32-way conflicts through the whole combine phase (5.3 of the kernel's 32 us, profiles/r4_27)
    constexpr int LWP = LW + 1;
    static_assert(NLT % 4 == 0 && LWP * 32 * 4 <= (int)kPfLds, "logit tiles");
    // kh and mt are wave-uniform: as scalars (readfirstlane) the step bounds below are scalar branches; as functions of threadIdx they
    // were exec-masked regions whose ring refills hipcc waited for on the spot (vmcnt(0) after every refill: r3_47)
    const int t = threadIdx.x & 255, lane = t & 63, n = lane & 31, kg = lane >> 5;
    const int kh = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 8)), mt = __builtin_amdgcn_readfirstlane(t >> 6);
    // (a workgroup's MFMA columns are 32 position slots; with kPfPos < 32 slot n holds position n mod kPfPos: see kPfPos)
    PF_T(0);
    const int b0 = (int)blockIdx.x * kPfPos;
    char* const sm = psm + (uint32_t)kh * 2u * kPfBuf;
    // K half kh = chunks NCH kh .. NCH kh + NCH - 1 of 16 pixels (the last ones are partly / wholly past S*S).  Staging of a
    // chunk: unit u = 256 i + t, i = 0..7: segment u >> 5 = (position, half), 16-byte unit u & 31 = (pixel, octet)
    const char* src[8];
    uint32_t dst[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int u = 256 * i + t, seg = u >> 5, pl = seg >> 1, hl = seg & 1, wi = u & 31, px = wi >> 1, oc = wi & 1;
        const int pq = b0 + (pl & (kPfPos - 1));
        const int p = pq < batch ? pq : batch - 1;
        src[i] = xp + (size_t)p * Hx<G>::kPolPos + (uint32_t)hl * Hx<G>::kPolLo + (uint32_t)((HXP / 2) * kh + px) * 32u + (uint32_t)oc * 16u;
        dst[i] = (uint32_t)((hl * 16 + px) * 2 + oc) * kPfRow + (uint32_t)pl * 16u;
    }
    uint4 stg[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) stg[i] = *reinterpret_cast<const uint4*>(src[i]);
    uint4 ring[RD][MT][2];
    const int step0 = (HXP / 2) * kh, nstep = NPIX - step0 < HXP / 2 ? NPIX - step0 : HXP / 2;
    const uint4* ap = a + ((size_t)step0 * NLT + mt) * 128 + lane;           // + step * NLT * 128 + (4 m) * 128 + half * 64
#pragma unroll
    for (int q = 0; q < RD; ++q)
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            ring[q][m][0] = ap[((size_t)q * NLT + 4 * m) * 128];
            ring[q][m][1] = ap[((size_t)q * NLT + 4 * m) * 128 + 64];
        }
#pragma unroll
    for (int i = 0; i < 8; ++i) *reinterpret_cast<uint4*>(sm + dst[i]) = stg[i];
    f32x16 acc[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][r] = 0.0f;
    __syncthreads();
    PF_T(1);
    for (int c = 0; c < NCH; ++c) {
        const uint32_t cur = (uint32_t)(c & 1) * kPfBuf, nxt = kPfBuf - cur;
        if (c < NCH - 1) {
#pragma unroll
            for (int i = 0; i < 8; ++i) stg[i] = *reinterpret_cast<const uint4*>(src[i] + (size_t)(c + 1) * 512u);
        }
        const char* bb = sm + cur + (uint32_t)kg * kPfRow + (uint32_t)n * 16u;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int step = 16 * c + q;
            {   // steps past the board (K half 1: pixels S*S .. HXP-1) run too: their B fragments are the zero pixel slots of the head
                // buffer and the A fragment is the last valid step's again, so they add exact zeros — and the body has no condition
                const h8 bh = *reinterpret_cast<const h8*>(bb + (uint32_t)(2 * q) * kPfRow);
                const h8 bl = *reinterpret_cast<const h8*>(bb + (uint32_t)(2 * (16 + q)) * kPfRow);
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    h8 ah, al;
                    __builtin_memcpy(&ah, &ring[q % RD][m][0], 16);
                    __builtin_memcpy(&al, &ring[q % RD][m][1], 16);
                    {   // refill unconditionally (past the end: the last step again, never used): no branch, no phi around the loads
                        const int pf = step + RD < nstep ? step + RD : nstep - 1;
                        ring[q % RD][m][0] = ap[((size_t)pf * NLT + 4 * m) * 128];
                        ring[q % RD][m][1] = ap[((size_t)pf * NLT + 4 * m) * 128 + 64];
                    }
                    acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[m], 0, 0, 0);
                    acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc[m], 0, 0, 0);
                    acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc[m], 0, 0, 0);
                }
            }
        }
        if (c < NCH - 1) {
#pragma unroll
            for (int i = 0; i < 8; ++i) *reinterpret_cast<uint4*>(sm + nxt + dst[i]) = stg[i];
        }
        __syncthreads();
    }
    PF_T(2);
    // the two K halves meet in LDS (the staging buffers are dead): rows of a lane = logits 32 (mt + 4 m) + 16 kg + r
    float* Lg = reinterpret_cast<float*>(psm);                               // [32 positions][LWP]
    if (kh == 1) {
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) Lg[n * LWP + 32 * (mt + 4 * m) + 16 * kg + r] = acc[m][r];
    }
    __syncthreads();
    if (kh == 0) {
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int j = 32 * (mt + 4 * m) + 16 * kg + r;
                Lg[n * LWP + j] = j < NPIX ? (acc[m][r] + Lg[n * LWP + j]) * inv_scale + bf[j] : -3.0e38f;
            }
    }
    __syncthreads();
    PF_T(3);
    const int p = (int)threadIdx.x >> 4, sub = (int)threadIdx.x & 15;        // 16 threads per position, LW / 16 logits each
    constexpr int NV = LW / 16;
    float v[NV], mx = -3.0e38f;
#pragma unroll
    for (int q = 0; q < NV; ++q) { v[q] = Lg[p * LWP + sub + 16 * q]; mx = fmaxf(mx, v[q]); }
#pragma unroll
    for (int off = 8; off >= 1; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
    float sum = 0.0f;
#pragma unroll
    for (int q = 0; q < NV; ++q) { v[q] = (sub + 16 * q) < NPIX ? expf(v[q] - mx) : 0.0f; sum += v[q]; }
#pragma unroll
    for (int off = 8; off >= 1; off >>= 1) sum += __shfl_xor(sum, off);
    if (p < kPfPos && b0 + p < batch) {
#pragma unroll
        for (int q = 0; q < NV; ++q) {
            const int j = sub + 16 * q;
            if (j < NPIX) policy[(size_t)(b0 + p) * NPIX + j] = v[q] / sum;
        }
    }
    PF_T(4);
}
#ifdef AF_PFC_TIMING
extern "C" int af_pfc_debug_wall(unsigned long long* host) {
    FS_HIP_OK(hipDeviceSynchronize());
    FS_HIP_OK(hipMemcpyFromSymbol(host, HIP_SYMBOL(g_pfc_wall), sizeof(unsigned long long) * 256 * 8 * 5));
    return 0;
}
#endif

// ------------------------------------------------------------------ host ------------------------------------------------------------------
struct LayerCfg { int cin, cout, pcin, CT, KS, PS, gy, pj; };
// block b: conv1 = layer 2b, conv2 (+ projection) = layer 2b+1.  pcin: channels of the projection input folded into this
// layer as extra k-steps; pj: 1 = this conv1 also produces the block's projection (from its own input), 2 = this conv2
// adds the produced projection (blocks 3 and 5: their block input is 4x / 2x wider than the block, so streaming it
// again for a 1x1 convolution costs more HBM time than the layer's own work)
const LayerCfg kLayers[10] = {
    {32, 64, 0, 2, 1, 2, 1, 0},   {64, 64, 32, 2, 1, 2, 1, 0},      // bone/block1
    {64, 128, 0, 4, 1, 1, 1, 0},  {128, 128, 64, 2, 2, 1, 2, 0},    // bone/block2
    {128, 32, 0, 1, 2, 2, 1, 1},  {32, 32, 0, 1, 1, 4, 1, 2},       // value/block3
    {128, 64, 0, 2, 2, 1, 1, 0},  {64, 64, 128, 2, 2, 1, 1, 0},     // policy/block4
    {64, 32, 0, 1, 1, 4, 1, 1},   {32, 32, 0, 1, 1, 4, 1, 2},       // policy/block5
};
const char* const kBlockNames[5] = {"bone/block1", "bone/block2", "value/block3", "policy/block4", "policy/block5"};

template <class T>
int dev_upload(std::vector<void*>& allocs, T** dst, const void* src, size_t bytes) {
    void* q = nullptr;
    FS_HIP_OK(hipMalloc(&q, bytes));
    FS_HIP_OK(hipMemcpy(q, src, bytes, hipMemcpyHostToDevice));
    allocs.push_back(q);
    *dst = (T*)q;
    return 0;
}

// A fragments of one layer: [cout tile][ks][item][hi|lo][lane][8]; item order = the kernel's consumption order
// (projection slabs first; per slab: 16-channel k-step c, then the 9 taps); MFMA row m of a tile holds cout
// 32*tile + 16*((m>>2)&1) + 8*(m>>4) + 4*((m>>3)&1) + (m&3) so that a lane's 16 accumulator rows are consecutive couts.
// w3: HWIO [9][cin][cout]; w1: [pcin][cout] or null.
std::vector<_Float16> pack_layer(const LayerCfg& L, const float* w3, const float* w1, float scale) {
    const int NSM = L.cin / 32, NSP = L.pcin / 32, C16 = 2 / L.KS;
    const int NIT = NSP * C16 + NSM * 9 * C16, tiles = L.cout / 32;
    std::vector<_Float16> out((size_t)tiles * L.KS * 2 * NIT * 64 * 8);
    for (int tile = 0; tile < tiles; ++tile)
        for (int ks = 0; ks < L.KS; ++ks) {
            int item = 0;
            for (int j = 0; j < NSP + NSM; ++j) {
                const bool proj = j < NSP;
                const int s = proj ? j : j - NSP;
                for (int c = 0; c < C16; ++c)
                    for (int tap = 0; tap < (proj ? 1 : 9); ++tap, ++item) {
                        const int c16 = L.KS == 2 ? ks : c;
                        for (int lane = 0; lane < 64; ++lane)
                            for (int e = 0; e < 8; ++e) {
                                const int m = lane & 31;
                                const int co = 32 * tile + 16 * ((m >> 2) & 1) + 8 * (m >> 4) + 4 * ((m >> 3) & 1) + (m & 3);
                                const int ci = 32 * s + 16 * c16 + 8 * (lane >> 5) + e;
                                const float v = (proj ? w1[(size_t)ci * L.cout + co] : w3[((size_t)tap * L.cin + ci) * L.cout + co]) * scale;
                                const _Float16 h = (_Float16)v;
                                const size_t base = ((((size_t)(tile * L.KS + ks) * NIT + item) * 2) * 64 + lane) * 8 + e;
                                out[base] = h;
                                out[base + 64 * 8] = (_Float16)(v - (float)h);
                            }
                    }
            }
        }
    return out;
}

// A fragments of a separately produced 1x1 projection (producer layer L: its own cin -> `cout` couts): [cout tile][ks][item]
// [hi|lo][lane][8], one item per (slab, 16-channel k-step), in the producer's order.  w1: [cin][cout].
std::vector<_Float16> pack_proj(const LayerCfg& L, int cout, const float* w1, float scale) {
    const int NSM = L.cin / 32, C16 = 2 / L.KS, NPW = NSM * C16, tiles = cout / 32;
    std::vector<_Float16> out((size_t)tiles * L.KS * 2 * NPW * 64 * 8);
    for (int tile = 0; tile < tiles; ++tile)
        for (int ks = 0; ks < L.KS; ++ks)
            for (int s = 0; s < NSM; ++s)
                for (int c = 0; c < C16; ++c) {
                    const int item = s * C16 + c, c16 = L.KS == 2 ? ks : c;
                    for (int lane = 0; lane < 64; ++lane)
                        for (int e = 0; e < 8; ++e) {
                            const int m = lane & 31;
                            const int co = 32 * tile + 16 * ((m >> 2) & 1) + 8 * (m >> 4) + 4 * ((m >> 3) & 1) + (m & 3);
                            const int ci = 32 * s + 16 * c16 + 8 * (lane >> 5) + e;
                            const float v = w1[(size_t)ci * cout + co] * scale;
                            const _Float16 h = (_Float16)v;
                            const size_t base = ((((size_t)(tile * L.KS + ks) * NPW + item) * 2) * 64 + lane) * 8 + e;
                            out[base] = h;
                            out[base + 64 * 8] = (_Float16)(v - (float)h);
                        }
                }
    return out;
}

float pick_scale(const std::vector<float>& a, const std::vector<float>* b) {
    float mx = 0.0f;
    for (float v : a) mx = std::max(mx, std::fabs(v));
    if (b) for (float v : *b) mx = std::max(mx, std::fabs(v));
    if (!(mx > 0.0f) || !std::isfinite(mx)) return 1.0f;
    int e;
    std::frexp(mx, &e);                     // mx = f * 2^e, f in [0.5, 1)
    return std::ldexp(1.0f, 13 - e);        // mx * scale in [4096, 8192): 8x below the fp16 maximum
}


// MFMA row m of a 32-row tile -> the output it carries, such that a lane's 16 accumulator rows are 16 consecutive outputs
inline int row_perm(int m) { return 16 * ((m >> 2) & 1) + 8 * (m >> 4) + 4 * ((m >> 3) & 1) + (m & 3); }

template <class F>
std::vector<_Float16> pack_frags(int nfrag, F&& val) {          // [fragment][hi|lo][lane][8], val(fragment, lane, e) already scaled
    std::vector<_Float16> out((size_t)nfrag * 2 * 64 * 8);
    for (int f = 0; f < nfrag; ++f)
        for (int lane = 0; lane < 64; ++lane)
            for (int e = 0; e < 8; ++e) {
                const float v = val(f, lane, e);
                const _Float16 h = (_Float16)v;
                out[(((size_t)f * 2 + 0) * 64 + lane) * 8 + e] = h;
                out[(((size_t)f * 2 + 1) * 64 + lane) * 8 + e] = (_Float16)(v - (float)h);
            }
    return out;
}

constexpr int kSmallBatch = 8;             // <= this many positions: a position's pixel tiles are split over workgroups (af_conv_f16s_sb)
template <class G, int NSM, int NSP, int CT, int KS, int PS, bool OUT32, bool XACC, int PJ = 0, int HD = 0, int DIST = kDist, int WPE = 1>
int launch_cfg(hipStream_t st, const F16sArgs& a, int gy, int ncu) {
    constexpr int NT = 4 / PS;
    constexpr size_t lds = Lds<G, DIST>::kScrOff + (KS == 2 ? (size_t)CT * PS * NT * 4096 * (PJ == 1 ? 2 : 1) : 0);
    static_assert(WPE * lds <= 160 * 1024, "LDS budget of the workgroups sharing a CU");
#ifndef AF_F16S_NO_LDS_ASSERT                    // (A/B builds with a deeper ring only fit the 11x11 geometry)
    static_assert(lds <= 160 * 1024, "LDS budget");
#endif
    // the attribute belongs to the (function, device) pair: one bit per device for this instantiation (a process may drive
    // several GPUs, one handle each; handles are created on their own device and launched with it current)
    static std::atomic<uint64_t> attr_devs{0};
    int dev = 0;
    FS_HIP_OK(hipGetDevice(&dev));
    const uint64_t bit = 1ull << (dev & 63);
    if (!(attr_devs.load(std::memory_order_relaxed) & bit)) {
        FS_HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(af_conv_f16s<G, NSM, NSP, CT, KS, PS, OUT32, XACC, PJ, HD, DIST, WPE>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_devs.fetch_or(bit, std::memory_order_relaxed);
    }
    constexpr int NTS = KS == 2 ? 2 : 1;                                  // tiles per wave of the small-batch split
    if constexpr (NTS < NT && HD == 0) {
        if (a.batch <= kSmallBatch && !(a.abl & 128)) {                    // (abl bit 7: the one-workgroup-per-position launch, for A/B)
            static std::atomic<uint64_t> attr_sb{0};
            if (!(attr_sb.load(std::memory_order_relaxed) & bit)) {
                FS_HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(af_conv_f16s_sb<G, NSM, NSP, CT, KS, PS, OUT32, XACC, PJ, HD, DIST, WPE, NTS>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                attr_sb.fetch_or(bit, std::memory_order_relaxed);
            }
            hipLaunchKernelGGL((af_conv_f16s_sb<G, NSM, NSP, CT, KS, PS, OUT32, XACC, PJ, HD, DIST, WPE, NTS>),
                               dim3(a.batch * G::HALVES, gy, NT / NTS), dim3(256), lds, st, a);
            FS_HIP_OK(hipGetLastError());
            return 0;
        }
    }
    // one workgroup per CU, a multiple of HALVES of them (a workgroup keeps one half of the board for the whole launch)
    int gx = std::max(1, std::min(a.batch * G::HALVES, WPE * ncu / gy));
    gx = std::max(G::HALVES, gx / G::HALVES * G::HALVES);
    hipLaunchKernelGGL((af_conv_f16s<G, NSM, NSP, CT, KS, PS, OUT32, XACC, PJ, HD, DIST, WPE>), dim3(gx, gy), dim3(256), lds, st, a);
    FS_HIP_OK(hipGetLastError());                   // a rejected launch (LDS attribute, grid) must not pass silently
    return 0;
}

// blocks 3 and 5 (11x11): both convolutions + the fused head in one launch (af_block_f16s)
template <class G, int NSM, int KS, int PS, bool XACC, int FU>
int launch_block(hipStream_t st, const F16sArgs& a, int ncu) {
    constexpr int DIST = 2;                       // (a 5-slot ring + the activation slab + 36 KB of second-layer weights exceed 160 KB)
    constexpr size_t scr = KS == 2 ? (size_t)PS * (4 / PS) * 4096 * 2 : 0;
    constexpr size_t lds = Lds<G, DIST>::kScrOff + (scr > Lay<G>::kSlotL ? scr : Lay<G>::kSlotL) + 36864;
    static_assert(lds <= 160 * 1024, "LDS budget of the fused block");
    static std::atomic<uint64_t> attr_devs{0};
    int dev = 0;
    FS_HIP_OK(hipGetDevice(&dev));
    const uint64_t bit = 1ull << (dev & 63);
    if (!(attr_devs.load(std::memory_order_relaxed) & bit)) {
        FS_HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(af_block_f16s<G, NSM, KS, PS, XACC, FU, DIST>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_devs.fetch_or(bit, std::memory_order_relaxed);
    }
    const int gx = std::max(1, std::min(a.batch, ncu));
    hipLaunchKernelGGL((af_block_f16s<G, NSM, KS, PS, XACC, FU, DIST>), dim3(gx), dim3(256), lds, st, a);
    FS_HIP_OK(hipGetLastError());
    return 0;
}

// 15x15, PS = 1 layers: the two half classes in one launch (af_conv_f16s_h15) + the corner pixel (af_corner_f16s) behind it
template <class G, int NSM, int NSP, int CT, int KS, bool XACC>
int launch_h15(hipStream_t st, F16sArgs a, int gy, int ncu) {
    constexpr size_t lds = Lds<G, kDist>::kScrOff + (KS == 2 ? (size_t)CT * 4 * 4096 : 0);
    static_assert(lds <= 160 * 1024, "LDS budget");
    static std::atomic<uint64_t> attr_devs{0};
    int dev = 0;
    FS_HIP_OK(hipGetDevice(&dev));
    const uint64_t bit = 1ull << (dev & 63);
    if (!(attr_devs.load(std::memory_order_relaxed) & bit)) {
        FS_HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(af_conv_f16s_h15<G, NSM, NSP, CT, KS, XACC>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_devs.fetch_or(bit, std::memory_order_relaxed);
    }
    // gx workgroups per blockIdx.y, split 9 : 7 between the 4-tile class (half 0) and the 3-tile class (half 1), each a multiple of 8
    // (4 x 28.4 = 114 against 3 x 36.6 = 110 tile passes per workgroup at 4096 boards on 144 + 112 workgroups)
    const int gx = std::max(16, ncu / gy / 16 * 16);
    // class sizes: KS = 1 (every wave finishes all its tiles: a workgroup's time scales with its tile count) 9 : 7; KS = 2 (the k-split
    // exchange, the finishing wave's two epilogues and the slab stream do not shrink with the third tile: a half-1 workgroup's board costs
    // ~0.86 of a half-0 one, profiles/r5_10) 17 : 15.  A/B knobs: AF_F16S_H15_NUM1 / _NUM2 = gx0 * 32 / gx, AF_F16S_H15_ROUND = 1 rounds to 8.
    static const int num1 = [] { const char* e = getenv("AF_F16S_H15_NUM1"); const int v = e ? atoi(e) : 0; return v > 0 && v < 32 ? v : 18; }();
    static const int num2 = [] { const char* e = getenv("AF_F16S_H15_NUM2"); const int v = e ? atoi(e) : 0; return v > 0 && v < 32 ? v : 17; }();
    static const int rnd = [] { const char* e = getenv("AF_F16S_H15_ROUND"); return e ? atoi(e) : 1; }();
    const int num = KS == 1 ? num1 : num2;
    a.gx0 = rnd ? std::max(8, (gx * num / 32 + 4) / 8 * 8) : std::max(1, gx * num / 32);
    if (a.gx0 >= gx) a.gx0 = gx - 8;
    hipLaunchKernelGGL((af_conv_f16s_h15<G, NSM, NSP, CT, KS, XACC>), dim3(gx, gy), dim3(256), lds, st, a);
    FS_HIP_OK(hipGetLastError());
    CornerArgs c;
    c.stash = a.stash; c.w = a.w; c.bias = a.bias; c.out = a.out; c.inv_scale = a.inv_scale;
    c.batch = a.batch; c.NCT = gy * CT;
    hipLaunchKernelGGL((af_corner_f16s<G, KS, XACC, NSM, NSP>), dim3((a.batch + 31) / 32, (c.NCT + 3) / 4), dim3(256), 0, st, c);
    FS_HIP_OK(hipGetLastError());
    return 0;
}

}  // namespace

struct f16s_net {
    int S = 11;                   // board size: 11 (one pseudo-position per board) or 15 (two halves)
    int max_batch = 0, device = 0, ncu = 256;
    std::vector<void*> allocs;
    float *stem_w = nullptr, *stem_b = nullptr;
    uint4* stem_wm = nullptr;     // af_stem_mfma_f16s A fragments
    float stem_inv_scale = 1.0f;
    uint4* w[10] = {};
    uint4* pw[10] = {};           // producer layers: the block's projection weights
    float* pbuf[5] = {};          // blocks with a separately produced projection: fp32, accumulator layout
    float* bias[10] = {};
    float inv_scale[10] = {}, inv_scale_p[10] = {};
    // S32 activations: f0, then per block g (conv1 output) and o (block output; blocks 2 and 4 end in fp32 planes)
    char *f0 = nullptr, *g[5] = {}, *o[5] = {};
    // fused head inputs + dense layers (value = 0, policy = 1)
    uint4 *hcw[2] = {}, *hfw[2] = {};     // 1x1 conv A fragments; dense A fragments
    float *hcb[2] = {}, *hfb[2] = {}, *v2w = nullptr, *v2b = nullptr;
    float hc_inv[2] = {}, hf_inv[2] = {};
    char* hx[2] = {};
    int* df = nullptr;            // af_small_forward_f16s: role counters (kDfWords ints, zero between launches)
    float* pfx = nullptr;         // ... and the policy head's K-half hand-over (16 KB)
    char* stash = nullptr;        // 15x15: corner operands of the layer in flight (af_conv_f16s_h15 -> af_corner_f16s), 6 slabs x 512 B per board
    int abl = 0;
};

int f16s_supported(int board_size) { return board_size == 11 || board_size == 15; }

int f16s_create(f16s_net** out, int board_size, int max_batch, int device, const std::map<std::string, std::vector<float>>& V) {
    if (!f16s_supported(board_size)) return -1;
    FS_HIP_OK(hipSetDevice(device));
    f16s_net* n = new f16s_net();
    n->S = board_size; n->max_batch = max_batch; n->device = device;
    const bool s11 = board_size == 11;
    const int halves = s11 ? 1 : 2;
    const size_t slab_bytes = s11 ? Lay<Geo<11>>::kSlabH : Lay<Geo<15>>::kSlabH;
    FS_HIP_OK(hipDeviceGetAttribute(&n->ncu, hipDeviceAttributeMultiprocessorCount, device));
    if (const char* e = getenv("AF_F16S_NCU")) {      // profiling only (tools/probe_power_bound.py): persistent workgroups of a launch = CUs it may occupy
        const int v = atoi(e);
        if (v >= 8 && v <= n->ncu) n->ncu = v / 8 * 8;
    }
    int rc = 0;
    auto get = [&](const std::string& k) -> const std::vector<float>& { return V.at(k); };
    rc = dev_upload(n->allocs, &n->stem_w, get("bone/conv1/kernel").data(), 75 * 32 * 4);
    if (!rc) rc = dev_upload(n->allocs, &n->stem_b, get("bone/conv1/bias").data(), 32 * 4);
    if (!rc) {              // [k-step s][hi|lo][lane][8]: MFMA row m -> cout perm(m); k = 8*(lane>>5) + e of group g = 2s + (lane>>5) = cin*5 + ky, tap kx = e
        const std::vector<float>& ks = get("bone/conv1/kernel");         // HWIO [5][5][3][32]
        const float sc = pick_scale(ks, nullptr);
        std::vector<_Float16> pk((size_t)8 * 2 * 64 * 8, (_Float16)0.0f);
        for (int s_ = 0; s_ < 8; ++s_)
            for (int lane = 0; lane < 64; ++lane)
                for (int e = 0; e < 5; ++e) {
                    const int m = lane & 31, g = 2 * s_ + (lane >> 5);
                    if (g >= 15) continue;
                    const int co = 16 * ((m >> 2) & 1) + 8 * (m >> 4) + 4 * ((m >> 3) & 1) + (m & 3), cin = g / 5, ky = g % 5;
                    const float v = ks[(size_t)((ky * 5 + e) * 3 + cin) * 32 + co] * sc;
                    const _Float16 h = (_Float16)v;
                    pk[(((size_t)s_ * 2 + 0) * 64 + lane) * 8 + e] = h;
                    pk[(((size_t)s_ * 2 + 1) * 64 + lane) * 8 + e] = (_Float16)(v - (float)h);
                }
        rc = dev_upload(n->allocs, &n->stem_wm, pk.data(), pk.size() * 2);
        n->stem_inv_scale = 1.0f / sc;
    }
    for (int b = 0; b < 5 && !rc; ++b) {
        const std::string s = kBlockNames[b];
        const std::vector<float>&k1 = get(s + "_conv1/kernel"), &k2 = get(s + "_conv2/kernel"), &kr = get(s + "_res/kernel");
        const LayerCfg &L1 = kLayers[2 * b], &L2 = kLayers[2 * b + 1];
        const bool split_proj = L1.pj == 1;
        const float s1 = pick_scale(k1, nullptr), s2 = pick_scale(k2, split_proj ? nullptr : &kr);
        const std::vector<_Float16> p1 = pack_layer(L1, k1.data(), nullptr, s1), p2 = pack_layer(L2, k2.data(), kr.data(), s2);
        rc = dev_upload(n->allocs, &n->w[2 * b], p1.data(), p1.size() * 2);
        if (!rc && split_proj) {
            const float sp = pick_scale(kr, nullptr);
            const std::vector<_Float16> pp = pack_proj(L1, L2.cout, kr.data(), sp);
            rc = dev_upload(n->allocs, &n->pw[2 * b], pp.data(), pp.size() * 2);
            n->inv_scale_p[2 * b] = 1.0f / sp;
            void* q = nullptr;
            const size_t bytes = (size_t)max_batch * halves * (L2.cout / 32) * 4 * 4 * 64 * 16;      // [pseudo-position][cout tile][4 pixel tiles][4][64] float4
            if (!rc) { FS_HIP_OK(hipMalloc(&q, bytes)); n->allocs.push_back(q); n->pbuf[b] = (float*)q; }
        }

        if (!rc) rc = dev_upload(n->allocs, &n->w[2 * b + 1], p2.data(), p2.size() * 2);
        std::vector<float> bsum(get(s + "_conv2/bias"));
        for (size_t i = 0; i < bsum.size(); ++i) bsum[i] += get(s + "_res/bias")[i];
        if (!rc) rc = dev_upload(n->allocs, &n->bias[2 * b], get(s + "_conv1/bias").data(), L1.cout * 4);
        if (!rc) rc = dev_upload(n->allocs, &n->bias[2 * b + 1], bsum.data(), L2.cout * 4);
        n->inv_scale[2 * b] = 1.0f / s1; n->inv_scale[2 * b + 1] = 1.0f / s2;
    }
    auto act = [&](char** p, int ch) -> int {
        void* q = nullptr;
        const size_t bytes = (size_t)max_batch * (ch / 32) * slab_bytes;
        FS_HIP_OK(hipMalloc(&q, bytes));
        FS_HIP_OK(hipMemset(q, 0, bytes));           // the zero units of S32 are never written again
        n->allocs.push_back(q);
        *p = (char*)q;
        return 0;
    };
    if (!rc && !s11) {
        void* q = nullptr;
        FS_HIP_OK(hipMalloc(&q, (size_t)max_batch * 6 * 512));
        n->allocs.push_back(q);
        n->stash = (char*)q;
    }
    if (!rc) {
        void* q = nullptr;
        FS_HIP_OK(hipMalloc(&q, kDfWords * sizeof(int)));
        FS_HIP_OK(hipMemset(q, 0, kDfWords * sizeof(int)));
        n->allocs.push_back(q);
        n->df = (int*)q;
        FS_HIP_OK(hipMalloc(&q, 4 * 4 * 64 * 16));
        n->allocs.push_back(q);
        n->pfx = (float*)q;
    }
    if (!rc) rc = act(&n->f0, 32);
    for (int b = 0; b < 5 && !rc; ++b) {
        rc = act(&n->g[b], kLayers[2 * b].cout);
        if (!rc && b != 2 && b != 4) rc = act(&n->o[b], kLayers[2 * b + 1].cout);
    }
    if (!rc) {          // heads: value/conv [32][4], value/fc1 [4*S*S][64], value/fc2 [64][1]; policy/conv [32][16], policy/fc [16*S*S][S*S]
        const int npix = board_size * board_size, hxp = 128 * halves, vsteps = (npix + 3) / 4, nlt = (npix + 31) / 32;
        const char* cname[2] = {"value/conv", "policy/conv"};
        const int nco[2] = {4, 16};
        for (int h = 0; h < 2 && !rc; ++h) {
            const std::vector<float>& wc = get(std::string(cname[h]) + "/kernel");
            const float sc = pick_scale(wc, nullptr);
            const int NC = nco[h];
            const std::vector<_Float16> pk = pack_frags(2, [&](int st_, int lane, int e) -> float {
                const int m = lane & 31, co = m < 16 ? 8 * ((m >> 2) & 1) + 4 * (m >> 3) + (m & 3) : 99, ci = 16 * (lane >> 5) + 8 * st_ + e;
                return co < NC ? wc[(size_t)ci * NC + co] * sc : 0.0f;
            });
            rc = dev_upload(n->allocs, &n->hcw[h], pk.data(), pk.size() * 2);
            std::vector<float> bpad(16, 0.0f);
            const std::vector<float>& bc = get(std::string(cname[h]) + "/bias");
            for (int i = 0; i < NC; ++i) bpad[i] = bc[i];
            if (!rc) rc = dev_upload(n->allocs, &n->hcb[h], bpad.data(), 64);
            n->hc_inv[h] = 1.0f / sc;
            void* q = nullptr;
            const size_t bytes = (size_t)max_batch * hxp * (h == 0 ? 16 : 64);
            if (!rc) { FS_HIP_OK(hipMalloc(&q, bytes)); FS_HIP_OK(hipMemset(q, 0, bytes)); n->allocs.push_back(q); n->hx[h] = (char*)q; }   // the pixel slots past S*S stay zero
        }
        if (!rc) {
            const std::vector<float>& w1 = get("value/fc1/kernel");
            const float sc = pick_scale(w1, nullptr);
            const std::vector<_Float16> pk = pack_frags(vsteps * 2, [&](int f, int lane, int e) -> float {
                const int s_ = f >> 1, mt = f & 1, px = 4 * s_ + 2 * (lane >> 5) + (e >> 2), ch = e & 3, j = 32 * mt + row_perm(lane & 31);
                return px < npix ? w1[(size_t)(ch * npix + px) * 64 + j] * sc : 0.0f;
            });
            rc = dev_upload(n->allocs, &n->hfw[0], pk.data(), pk.size() * 2);
            n->hf_inv[0] = 1.0f / sc;
            if (!rc) rc = dev_upload(n->allocs, &n->hfb[0], get("value/fc1/bias").data(), 64 * 4);
            if (!rc) rc = dev_upload(n->allocs, &n->v2w, get("value/fc2/kernel").data(), 64 * 4);
            if (!rc) rc = dev_upload(n->allocs, &n->v2b, get("value/fc2/bias").data(), 4);
        }
        if (!rc) {
            const std::vector<float>& wf = get("policy/fc/kernel");
            const float sc = pick_scale(wf, nullptr);
            const std::vector<_Float16> pk = pack_frags(npix * nlt, [&](int f, int lane, int e) -> float {
                const int px = f / nlt, mt = f % nlt, c16 = 8 * (lane >> 5) + e, j = 32 * mt + row_perm(lane & 31);
                return j < npix ? wf[(size_t)(c16 * npix + px) * npix + j] * sc : 0.0f;
            });
            rc = dev_upload(n->allocs, &n->hfw[1], pk.data(), pk.size() * 2);
            n->hf_inv[1] = 1.0f / sc;
            if (!rc) rc = dev_upload(n->allocs, &n->hfb[1], get("policy/fc/bias").data(), npix * 4);
        }
        if (!rc) FS_HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(af_policy_fc_f16s<Geo<11>>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kPfLds));
        if (!rc) FS_HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(af_policy_fc_f16s<Geo<15>>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kPfLds));
    }
    if (rc) { f16s_destroy(n); return rc; }
    *out = n;
    return 0;
}

void f16s_destroy(f16s_net* n) {
    if (!n) return;
    (void)hipSetDevice(n->device);
    for (void* p : n->allocs) (void)hipFree(p);
    delete n;
}

void f16s_set_ablation(f16s_net* n, int bits) { if (n) n->abl = bits; }

// the ten layers on one board geometry.  <NSM, NSP, CT, KS, PS, OUT32, XACC[, PJ, HD]>: XACC wherever weights + 2 x accumulators +
// fragments fit 512 registers; the fused head variants (HD = 1 value, 2 policy) are launched for both geometries
template <class G>
static int launch_layer_g(f16s_net* n, hipStream_t st, int li, const F16sArgs& a, int head) {
    switch (li) {
        // the three 32-channel-input layers (one slab per position): two workgroups per CU, prefetch distance 1 (abl bit 5 = the
        // one-workgroup instantiation, for A/B)
        // (11x11 only: two 3-slot rings of 20-KB slots + zero regions exceed 160 KB on 15x15)
        case 0:
            if constexpr (G::S == 11) { if (!(n->abl & 32)) return launch_cfg<G, 1, 0, 2, 1, 2, false, false, 0, 0, 1, 2>(st, a, 1, n->ncu); }   // (no XACC: 256 registers)
            return launch_cfg<G, 1, 0, 2, 1, 2, false, true>(st, a, 1, n->ncu);
        case 1: return launch_cfg<G, 2, 1, 2, 1, 2, false, true>(st, a, 1, n->ncu);
        // 15x15 (r5): the four PS = 1 layers run the 4-tile / 3-tile half classes + the corner kernel (abl bit 6 = the r3 two-halves launch,
        // for A/B; small batches keep it too: the class split assumes every workgroup has boards of its half)
        case 2:
            if constexpr (G::S == 15) { if (!(n->abl & 64) && a.batch >= 128) return launch_h15<G, 2, 0, 4, 1, false>(st, a, 1, n->ncu); }
            return launch_cfg<G, 2, 0, 4, 1, 1, false, false>(st, a, 1, n->ncu);   // (as 2 workgroup kinds x (2 tiles, k-split, XACC): 1.605 vs 1.580 ms per forward, |dp| 1.23e-5 vs 1.28e-5)
        case 3:
            if constexpr (G::S == 15) { if (!(n->abl & 64) && a.batch >= 128) return launch_h15<G, 4, 2, 2, 2, false>(st, a, 2, n->ncu); }
            return launch_cfg<G, 4, 2, 2, 2, 1, false, false>(st, a, 2, n->ncu);
        case 4: return launch_cfg<G, 4, 0, 1, 2, 2, false, false, 1>(st, a, 1, n->ncu);   // (no XACC: weights + projection weights + 3 accumulator sets would spill)
        case 5:                                                                           // (one pixel tile per wave, whole K: no k-split exchange)
            if constexpr (G::S == 11) { if (head == 0 && !(n->abl & 32)) return launch_cfg<G, 1, 0, 1, 1, 4, true, true, 2, 1, 1, 2>(st, a, 1, n->ncu); }
            if (head == 0) return launch_cfg<G, 1, 0, 1, 1, 4, true, true, 2, 1>(st, a, 1, n->ncu);
            return launch_cfg<G, 1, 0, 1, 1, 4, true, true, 2>(st, a, 1, n->ncu);
        case 6:
            if constexpr (G::S == 15) { if (!(n->abl & 64) && a.batch >= 128) return launch_h15<G, 4, 0, 2, 2, false>(st, a, 1, n->ncu); }
            return launch_cfg<G, 4, 0, 2, 2, 1, false, false>(st, a, 1, n->ncu);
        case 7:
            if constexpr (G::S == 15) { if (!(n->abl & 64) && a.batch >= 128) return launch_h15<G, 2, 4, 2, 2, true>(st, a, 1, n->ncu); }
            return launch_cfg<G, 2, 4, 2, 2, 1, false, true>(st, a, 1, n->ncu);
        case 8: return launch_cfg<G, 2, 0, 1, 1, 4, false, true, 1>(st, a, 1, n->ncu);
        default:
            if constexpr (G::S == 11) { if (head == 1 && !(n->abl & 32)) return launch_cfg<G, 1, 0, 1, 1, 4, true, true, 2, 2, 1, 2>(st, a, 1, n->ncu); }
            if (head == 1) return launch_cfg<G, 1, 0, 1, 1, 4, true, true, 2, 2>(st, a, 1, n->ncu);
            return launch_cfg<G, 1, 0, 1, 1, 4, true, true, 2>(st, a, 1, n->ncu);
    }
}

static int launch_layer(f16s_net* n, hipStream_t st, int li, const char* in, const char* in2, char* out, float* out32, int batch,
                        int WP, int PP, int head = -1) {
    F16sArgs a;
    a.hw = nullptr; a.hbias = nullptr; a.hx = nullptr; a.inv_scale_h = 1.0f;
    if (head >= 0) { a.hw = n->hcw[head]; a.hbias = n->hcb[head]; a.hx = n->hx[head]; a.inv_scale_h = n->hc_inv[head]; }
    a.in = in; a.in2 = in2; a.w = n->w[li]; a.bias = n->bias[li]; a.out = out; a.out32 = out32;
    a.inv_scale = n->inv_scale[li]; a.batch = batch; a.WP = WP; a.PP = PP; a.abl = (n->abl & 0xf0ff) | (li << 8); a.gx0 = 0; a.stash = n->stash;
    {   // profiling only: AF_F16S_ABL_LAYERS = bit mask of the layers the traffic-ablation bits (1, 2, 4, 8, 4096) apply to (default: all)
        static const int mask = [] { const char* e = getenv("AF_F16S_ABL_LAYERS"); return e ? (int)strtol(e, nullptr, 0) : 0x3ff; }();
        if (!((mask >> li) & 1)) a.abl &= ~0x100f;
    }
    a.w2 = nullptr; a.bias2 = nullptr; a.inv_scale2 = 1.0f;
    a.pw = n->pw[li]; a.pbuf = n->pbuf[li / 2]; a.inv_scale_p = n->inv_scale_p[li];
    return n->S == 11 ? launch_layer_g<Geo<11>>(n, st, li, a, head) : launch_layer_g<Geo<15>>(n, st, li, a, head);
}

int f16s_trunk(f16s_net* n, hipStream_t st, const float* planes, int batch) {
    if (!n || batch < 1 || batch > n->max_batch) return -1;
    if (n->abl & 16) {       // A/B: the VALU stem
        if (n->S == 15) hipLaunchKernelGGL(af_stem_f16s<Geo<15>>, dim3(std::min(batch, 2048)), dim3(256), 0, st, planes, n->stem_w, n->stem_b, n->f0, batch);
        else hipLaunchKernelGGL(af_stem_f16s<Geo<11>>, dim3(std::min(batch, 2048)), dim3(256), 0, st, planes, n->stem_w, n->stem_b, n->f0, batch);
    } else if (n->S == 15) {
        hipLaunchKernelGGL(af_stem_mfma_f16s<Geo<15>>, dim3(std::min(batch, 1024)), dim3(256), 0, st, planes, n->stem_wm, n->stem_b, n->stem_inv_scale,
                           n->f0, batch);
    } else {
        hipLaunchKernelGGL(af_stem_mfma_f16s<Geo<11>>, dim3(std::min(batch, 1024)), dim3(256), 0, st, planes, n->stem_wm, n->stem_b, n->stem_inv_scale,
                           n->f0, batch);
    }
    int rc = launch_layer(n, st, 0, n->f0, nullptr, n->g[0], nullptr, batch, 0, 0);
    if (!rc) rc = launch_layer(n, st, 1, n->g[0], n->f0, n->o[0], nullptr, batch, 0, 0);
    if (!rc) rc = launch_layer(n, st, 2, n->o[0], nullptr, n->g[1], nullptr, batch, 0, 0);
    if (!rc) rc = launch_layer(n, st, 3, n->g[1], n->o[0], n->o[1], nullptr, batch, 0, 0);
    return rc;
}

// F16sArgs of a fused block: the first convolution's (layer li) + the second one's weights / bias / scale + the head's
static F16sArgs block_args(f16s_net* n, int li, const char* in, int batch, int head) {
    F16sArgs a;
    memset(&a, 0, sizeof(a));
    a.in = in; a.w = n->w[li]; a.bias = n->bias[li]; a.inv_scale = n->inv_scale[li];
    a.pw = n->pw[li]; a.inv_scale_p = n->inv_scale_p[li];
    a.w2 = n->w[li + 1]; a.bias2 = n->bias[li + 1]; a.inv_scale2 = n->inv_scale[li + 1];
    a.hw = n->hcw[head]; a.hbias = n->hcb[head]; a.hx = n->hx[head]; a.inv_scale_h = n->hc_inv[head];
    a.batch = batch; a.abl = (n->abl & 0xf0ff) | (li << 8); a.stash = n->stash;
    return a;
}

int f16s_value_branch(f16s_net* n, hipStream_t st, int batch, float* o3, int WP, int PP, float* value) {
    int rc;
    if (n->S == 11 && value && !(n->abl & 256)) {          // (abl bit 8: the two-launch block, for A/B)
        rc = launch_block<Geo<11>, 4, 2, 2, false, 1>(st, block_args(n, 4, n->o[1], batch, 0), n->ncu);
    } else {
        rc = launch_layer(n, st, 4, n->o[1], nullptr, n->g[2], nullptr, batch, 0, 0);
        if (!rc) rc = launch_layer(n, st, 5, n->g[2], n->o[1], nullptr, o3, batch, WP, PP, value ? 0 : -1);
    }
    if (!rc && value) {
        if (n->S == 11)
            hipLaunchKernelGGL(af_value_fc_f16s<Geo<11>>, dim3((batch + 31) / 32), dim3(128), 0, st, n->hx[0], n->hfw[0], n->hfb[0], n->v2w, n->v2b,
                               n->hf_inv[0], value, batch);
        else
            hipLaunchKernelGGL(af_value_fc_f16s<Geo<15>>, dim3((batch + 31) / 32), dim3(128), 0, st, n->hx[0], n->hfw[0], n->hfb[0], n->v2w, n->v2b,
                               n->hf_inv[0], value, batch);
    }
    return rc;
}

int f16s_policy_branch(f16s_net* n, hipStream_t st, int batch, float* o5, int WP, int PP, float* policy) {
    int rc = launch_layer(n, st, 6, n->o[1], nullptr, n->g[3], nullptr, batch, 0, 0);
    if (!rc) rc = launch_layer(n, st, 7, n->g[3], n->o[1], n->o[3], nullptr, batch, 0, 0);
    if (!rc && n->S == 11 && policy && !(n->abl & 256)) {
        rc = launch_block<Geo<11>, 2, 1, 4, true, 2>(st, block_args(n, 8, n->o[3], batch, 1), n->ncu);
    } else {
        if (!rc) rc = launch_layer(n, st, 8, n->o[3], nullptr, n->g[4], nullptr, batch, 0, 0);
        if (!rc) rc = launch_layer(n, st, 9, n->g[4], n->o[3], nullptr, o5, batch, WP, PP, policy ? 1 : -1);
    }
    if (!rc && policy) {
        if (n->S == 11)
            hipLaunchKernelGGL(af_policy_fc_f16s<Geo<11>>, dim3((batch + kPfPos - 1) / kPfPos), dim3(512), kPfLds, st, n->hx[1], n->hfw[1], n->hfb[1],
                               n->hf_inv[1], policy, batch);
        else
            hipLaunchKernelGGL(af_policy_fc_f16s<Geo<15>>, dim3((batch + kPfPos - 1) / kPfPos), dim3(512), kPfLds, st, n->hx[1], n->hfw[1], n->hfb[1],
                               n->hf_inv[1], policy, batch);
    }
    return rc;
}

// Small batches on 11x11 (heads fused): both branches on ONE stream — {policy conv1 || value block} as one launch, the value head's
// dense layers, then the rest of the policy branch.  The same kernels' bodies with the same arguments: the same bits.
int f16s_small_branches(f16s_net* n, hipStream_t st, int batch, float* value, float* policy) {
    if (!n || n->S != 11 || batch < 1 || !value || !policy) return -1;
    using G = Geo<11>;
    const bool small = batch <= kSmallBatch;
    F16sArgs P;
    memset(&P, 0, sizeof(P));
    P.in = n->o[1]; P.w = n->w[6]; P.bias = n->bias[6]; P.out = n->g[3]; P.inv_scale = n->inv_scale[6];
    P.batch = batch; P.abl = (n->abl & 0xf0ff) | (6 << 8); P.stash = n->stash; P.inv_scale2 = 1.0f; P.inv_scale_h = 1.0f;
    const F16sArgs V = block_args(n, 4, n->o[1], batch, 0);
    constexpr size_t ldsP = Lds<G, kDist>::kScrOff + (size_t)2 * 1 * 4 * 4096;                       // CT = 2, PS = 1, NTW = 4 (2 at small batches)
    constexpr size_t ldsV = Lds<G, 2>::kScrOff + (size_t)2 * 2 * 4096 * 2 + 36864;
    constexpr size_t lds = ldsP > ldsV ? ldsP : ldsV;
    static_assert(lds <= 160 * 1024, "LDS budget");
    static std::atomic<uint64_t> attr_devs{0};
    int dev = 0;
    FS_HIP_OK(hipGetDevice(&dev));
    const uint64_t bit = 1ull << (dev & 63);
    if (!(attr_devs.load(std::memory_order_relaxed) & bit)) {
        FS_HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(af_small_pair_f16s<G>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        FS_HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(af_big_pair_f16s<G>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_devs.fetch_or(bit, std::memory_order_relaxed);
    }
    const int gp = std::max(1, std::min(batch, n->ncu));          // workgroups of a class at full batch: one per CU
    if (small) {
        hipLaunchKernelGGL(af_small_pair_f16s<G>, dim3(3 * batch), dim3(256), lds, st, P, V);
    } else {
        P.gx0 = gp;
        hipLaunchKernelGGL(af_big_pair_f16s<G>, dim3(2 * gp), dim3(256), lds, st, P, V);
    }
    FS_HIP_OK(hipGetLastError());
    {
        F16sArgs Q;
        memset(&Q, 0, sizeof(Q));
        Q.in = n->g[3]; Q.in2 = n->o[1]; Q.w = n->w[7]; Q.bias = n->bias[7]; Q.out = n->o[3]; Q.inv_scale = n->inv_scale[7];
        Q.batch = batch; Q.abl = (n->abl & 0xf0ff) | (7 << 8); Q.stash = n->stash; Q.inv_scale2 = 1.0f; Q.inv_scale_h = 1.0f;
        const VfcArgs F = {n->hx[0], n->hfw[0], n->hfb[0], n->v2w, n->v2b, n->hf_inv[0], value};
        constexpr size_t lds7 = Lds<G, kDist>::kScrOff + (size_t)2 * 1 * 4 * 4096;
        static std::atomic<uint64_t> attr7{0};
        if (!(attr7.load(std::memory_order_relaxed) & bit)) {
            // (the kernels also have 256 bytes of static LDS — the value head's reduction buffer: dynamic + static must stay within 160 KB)
            FS_HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(af_small_l7v_f16s<G>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds7));
            FS_HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(af_big_l7v_f16s<G>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds7));
            attr7.fetch_or(bit, std::memory_order_relaxed);
        }
        if (small) {
            hipLaunchKernelGGL(af_small_l7v_f16s<G>, dim3(2 * batch + 1), dim3(256), lds7, st, Q, F);
        } else {
            Q.gx0 = gp;
            hipLaunchKernelGGL(af_big_l7v_f16s<G>, dim3(gp + (batch + 31) / 32), dim3(256), lds7, st, Q, F);
        }
        FS_HIP_OK(hipGetLastError());
    }
    int rc = launch_block<G, 2, 1, 4, true, 2>(st, block_args(n, 8, n->o[3], batch, 1), n->ncu);
    if (!rc) hipLaunchKernelGGL(af_policy_fc_f16s<G>, dim3((batch + kPfPos - 1) / kPfPos), dim3(512), kPfLds, st, n->hx[1], n->hfw[1], n->hfb[1],
                                n->hf_inv[1], policy, batch);
    FS_HIP_OK(hipGetLastError());
    return rc;
}

// The single-launch small-batch forward (af_small_forward_f16s) + the policy head's dense layer.
// (abl bit 11 = the multi-launch form, for A/B; it also yields to the older A/B bits of the small-batch path)
int f16s_small_forward_ok(const f16s_net* n, int batch) {
    return n && n->S == 11 && batch >= 1 && batch <= kSmallBatch && !(n->abl & (16 | 32 | 128 | 256 | 512 | 2048));
}
static F16sArgs layer_args(f16s_net* n, int li, const char* in, const char* in2, char* out, int batch) {
    F16sArgs a;
    memset(&a, 0, sizeof(a));
    a.in = in; a.in2 = in2; a.w = n->w[li]; a.bias = n->bias[li]; a.out = out; a.inv_scale = n->inv_scale[li]; a.batch = batch;
    a.abl = (n->abl & 0xf0ff) | (li << 8); a.stash = n->stash; a.inv_scale2 = 1.0f; a.inv_scale_h = 1.0f;
    a.pw = n->pw[li]; a.pbuf = n->pbuf[li / 2]; a.inv_scale_p = n->inv_scale_p[li];
    return a;
}
int f16s_small_forward(f16s_net* n, hipStream_t st, const float* planes, int batch, float* value, float* policy) {
    if (!f16s_small_forward_ok(n, batch) || !planes || !value || !policy) return -1;
    using G = Geo<11>;
    SmallFwdArgs S;
    memset(&S, 0, sizeof(S));
    S.planes = planes; S.stem_w = n->stem_wm; S.stem_b = n->stem_b; S.stem_inv = n->stem_inv_scale; S.f0 = n->f0;
    S.df = n->df; S.batch = batch;
    auto role = [&](F16sArgs a, int wait, int need, int done) { a.df_wait = n->df + wait; a.df_need = need; a.df_done = n->df + done; a.df_err = n->df + kDfErr; return a; };
    S.L0 = role(layer_args(n, 0, n->f0, nullptr, n->g[0], batch), kDfStem, 1, kDfL0);
    S.L1 = role(layer_args(n, 1, n->g[0], n->f0, n->o[0], batch), kDfL0, 2, kDfL1);
    S.L2 = role(layer_args(n, 2, n->o[0], nullptr, n->g[1], batch), kDfL1, 2, kDfL2);
    S.L3 = role(layer_args(n, 3, n->g[1], n->o[0], n->o[1], batch), kDfL2, 4, kDfL3);
    S.P = role(layer_args(n, 6, n->o[1], nullptr, n->g[3], batch), kDfL3, 4, kDfP);
    S.P.pw = nullptr; S.P.pbuf = nullptr;
    S.V = role(block_args(n, 4, n->o[1], batch, 0), kDfL3, 4, kDfV);
    S.Q = role(layer_args(n, 7, n->g[3], n->o[1], n->o[3], batch), kDfP, 2, kDfQ);
    S.Q.pw = nullptr; S.Q.pbuf = nullptr;
    S.B5 = role(block_args(n, 8, n->o[3], batch, 1), kDfQ, 2, kDfB5);
    S.F = VfcArgs{n->hx[0], n->hfw[0], n->hfb[0], n->v2w, n->v2b, n->hf_inv[0], value};
    constexpr size_t ldsC = Lds<G, kDist>::kScrOff + (size_t)2 * 1 * 4 * 4096;                 // the k-split layers (CT = 2, PS = 1)
    constexpr size_t ldsV = Lds<G, 2>::kScrOff + (size_t)2 * 2 * 4096 * 2 + 36864;           // the value block
    constexpr size_t ldsB = Lds<G, 2>::kScrOff + Lay<G>::kSlotL + 36864;                     // block 5
    constexpr size_t lds = std::max(std::max(ldsC, ldsV), std::max(ldsB, (size_t)StemLds<G>::kBytes));
    static_assert(lds + 256 <= 160 * 1024, "LDS budget (dynamic + the value head's static reduction buffer)");
    static std::atomic<uint64_t> attr_devs{0};
    int dev = 0;
    FS_HIP_OK(hipGetDevice(&dev));
    const uint64_t bit = 1ull << (dev & 63);
    if (!(attr_devs.load(std::memory_order_relaxed) & bit)) {
        FS_HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(af_small_forward_f16s<G>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_devs.fetch_or(bit, std::memory_order_relaxed);
    }
    S.xp = n->hx[1]; S.pfa = n->hfw[1]; S.pfb = n->hfb[1]; S.pf_inv = n->hf_inv[1]; S.policy = policy; S.pfx = n->pfx;
    hipLaunchKernelGGL(af_small_forward_f16s<G>, dim3(19 * batch + 3), dim3(256), lds, st, S);
    FS_HIP_OK(hipGetLastError());
    return 0;
}
// 1 if a role of af_small_forward_f16s ever gave up waiting (synchronises; tests)
int f16s_small_forward_error(f16s_net* n) {
    if (!n || !n->df) return -1;
    int e = 0;
    FS_HIP_OK(hipDeviceSynchronize());
    FS_HIP_OK(hipMemcpy(&e, n->df + kDfErr, sizeof(int), hipMemcpyDeviceToHost));
    return e;
}
// (abl bit 9: the two-stream form at small batches; bit 10: at full batch — for A/B)
int f16s_small_branches_ok(const f16s_net* n, int batch) {
    if (!n || n->S != 11 || (n->abl & 256)) return 0;
    return batch <= kSmallBatch ? !(n->abl & (128 | 512)) : !(n->abl & 1024);
}

// debug / tests: activation `which` (0 f0, 1 g1, 2 o1, 3 g2, 4 o2, 5 g3, 6 g4, 7 o4, 8 g5) of the first `batch` positions as
// fp32 [batch][C][S*S] on the host (hi + lo).  Returns the channel count or < 0.
int f16s_read_activation(f16s_net* n, int which, int batch, float* host) {
    if (!n || batch < 1 || batch > n->max_batch) return -1;
    const char* bufs[9] = {n->f0, n->g[0], n->o[0], n->g[1], n->o[1], n->g[2], n->g[3], n->o[3], n->g[4]};
    const int chans[9] = {32, 64, 64, 128, 128, 32, 64, 64, 32};
    if (which < 0 || which > 8) return -1;
    const int C = chans[which];
    const bool s11 = n->S == 11;
    const size_t slab_h = (s11 ? Lay<Geo<11>>::kSlabH : Lay<Geo<15>>::kSlabH) / 2, half_h = slab_h / 2;      // in fp16 elements
    const int rowu = s11 ? (int)Geo<11>::ROWU : (int)Geo<15>::ROWU, poff = s11 ? Geo<11>::POFF : Geo<15>::POFF, npix = n->S * n->S;
    std::vector<_Float16> h((size_t)batch * (C / 32) * slab_h);
    FS_HIP_OK(hipDeviceSynchronize());
    FS_HIP_OK(hipMemcpy(h.data(), bufs[which], h.size() * 2, hipMemcpyDeviceToHost));
    for (int b = 0; b < batch; ++b)
        for (int c = 0; c < C; ++c)
            for (int p = 0; p < npix; ++p) {
                const size_t slab = ((size_t)b * (C / 32) + c / 32) * slab_h;
                const size_t idx = slab + ((size_t)((c % 32) / 8) * rowu + p + poff) * 8 + c % 8;
                host[((size_t)b * C + c) * npix + p] = (float)h[idx] + (float)h[idx + half_h];
            }
    return C;
}

#ifdef AF_DF_TIMING
extern "C" int af_df_debug_wall(unsigned long long* host) {
    FS_HIP_OK(hipDeviceSynchronize());
    FS_HIP_OK(hipMemcpyFromSymbol(host, HIP_SYMBOL(g_df_wall), sizeof(unsigned long long) * 256 * 5));
    return 0;
}
#endif
#ifdef AF_F16S_TIMING
extern "C" int af_f16s_debug_cycles(unsigned long long* host) {
    FS_HIP_OK(hipDeviceSynchronize());
    FS_HIP_OK(hipMemcpyFromSymbol(host, HIP_SYMBOL(g_f16s_cycles), sizeof(unsigned long long) * 10 * 512 * 4 * 9));
    return 0;
}
extern "C" int af_f16s_debug_wall(unsigned long long* host) {
    FS_HIP_OK(hipDeviceSynchronize());
    FS_HIP_OK(hipMemcpyFromSymbol(host, HIP_SYMBOL(g_f16s_wall), sizeof(unsigned long long) * 10 * 512 * 2));
    return 0;
}
#endif
