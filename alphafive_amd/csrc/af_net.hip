// af_net.hip — hand-written gfx950 forward pass of the alphaFive policy/value net
// (genData/network.py:52-97,163-165), fp32 on the matrix cores (v_mfma_f32_32x32x2_f32: exact fp32
// products, fp32 accumulate).  C ABI: include/af_net.h.
//
// Activations live in zero-padded planes [batch][channel][WP*WP (+pad)], WP = 2*ceil(S/2)+2, so a
// convolution tap is a constant address shift and no kernel needs bounds checks; every layer writes
// straight into the next layer's padded layout.  Kernels:
//   af_conv_wino         (97 % of this path's algorithmic FLOPs) Winograd F(2x2,3x3) for every 3x3
//                        layer with the block's 1x1 projection folded into the Winograd domain; 16
//                        accumulators in AGPRs, on-the-fly input transform, fused output transform +
//                        bias + residual + ELU.  3.5 ms per 4096 positions at 11x11 (profiles/r6_02).
//   af_stem_conv, af_value_head, af_policy_head<PPB>, af_policy_head_mfma   5x5 stem and the two heads.
// This is the fp32-operand path (af_net_tune(0, 1)): every board size other than 11x11 / 15x15, and the 24-bit
// leg of bench.py.  The product path on 11x11 / 15x15 is af_conv_f16s.hip (af_net_tune(0, 5), default), which
// this file's af_net_forward dispatches to.  The variants that lost their A/B (U through LDS, LDS-DMA staging,
// persistent grid, direct implicit GEMM, sub-batch streams) were removed from the source in r6; their records
// are in DESIGN_HISTORY.md and profiles/.

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <map>
#include <string>
#include <vector>

#include "af_net.h"
#include "af_conv_f16s.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));   // 4 floats, only 4-byte aligned
typedef float f2u __attribute__((ext_vector_type(2), aligned(4)));
typedef float f2a __attribute__((ext_vector_type(2)));               // 8-byte aligned pair (LDS reads)

// ELU(alpha=1).  exp(x)-1 through the hardware exp2 (absolute error ~1e-7 at x -> 0-, far inside the
// 1e-5 parity budget); expm1f costs ~30 VALU ops per element and sat un-overlapped in the epilogue.
__device__ __forceinline__ float elu1(float x) { return x > 0.0f ? x : __expf(x) - 1.0f; }

// ----------------------------------------------------------------------------------------------
// Winograd F(2x2,3x3) on the matrix cores: Y = A^T [ (G g G^T) (.) (B^T d B) ] A per 2x2 output
// tile, the 16 element-wise products becoming 16 independent [cout x cin] x [cin x tile] GEMMs:
// 16 MFMA k-steps per cin pair cover 32 tiles = 128 output pixels (direct: 9 k-steps cover 32
// pixels) => 2.25x fewer MFMA instructions for the layers that hold 97 % of the FLOPs.
//   * one wave = 32 tiles x 32 couts; all 16 Winograd-domain accumulators M[xi][nu] stay in
//     registers (256 AGPRs), so nothing is staged through memory: the input transform B^T d B
//     (32 adds on the lane's own 4x4 patch, loaded as 16 dwords) happens between the loads and the
//     MFMAs, the output transform A^T M A (24 adds per cout row) in the epilogue, fused with bias,
//     the residual projection, ELU and the store into the next layer's padded layout.
//   * the block's 1x1 projection is a 3x3 kernel with only the centre tap set; in the Winograd
//     domain only (xi,nu) in {1,2}^2 are non-zero, so it is 4 extra k-steps per cin pair into
//     M[1][1], M[1][2], M[2][1], M[2][2] and needs only the 2x2 centre of the patch.
//   * U = G g G^T is computed once on the host (fp64, rounded to fp32) and packed k-pair-major
//     [cin/2][2][cout][16] (a lane's 16 values = 4 dwordx4 loads); the 4 waves of a workgroup take 4 tile blocks of the SAME cout tile so
//     the weight stream is shared in L1.
// Plane layout: tiles need rows/cols 2t-1 .. 2t+2, so planes are padded to (2*ceil(S/2)+2)^2.
struct WinoArgs {
    const float* in;     // [batch][cin][PP]   input of the 3x3 convolution
    const float* u;      // [cin/2][2][cout][16]
    const float* in2;    // [batch][cin2][PP]  input of the 1x1 projection (or nullptr)
    const float* u2;     // [cin2/2][2][cout][4]
    const float* bias;   // [cout]
    float* out;          // [batch][cout][PP]
    int cin, cin2, cout, ntiles, T, S, WP, PP;
};

// (r1-r3 A/B records, removed from the source in r6: U through LDS — 6 % slower; both operands through LDS by LDS-DMA; a persistent grid;
//  a direct implicit-GEMM kernel af_conv_mfma — 5.2 ms against 3.5 per 4096 positions: profiles/r6_02_arith_width.json, DESIGN_HISTORY.md)
__device__ __forceinline__ void wino_task(const WinoArgs& A, const int bid) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = lane & 31, kh = lane >> 5;
    const int nct = A.cout >> 5;
    // XCD-aware block -> task map: workgroup b is observed to run on XCD b % 8 and each XCD has its own L2.
    // The nct workgroups that read the SAME activations (same tile-block group, different cout tile) get the
    // block ids g8, g8+8, g8+16, ... so they share one XCD's L2 instead of fetching the planes nct times.
    const int xcd = bid & 7, rest = bid >> 3;
    const int ct = rest % nct;
    const int tb = ((rest / nct) * 8 + xcd) * 4 + wave;
    if (tb * 32 >= A.ntiles) return;
    const int q = tb * 32 + col;
    const bool valid = q < A.ntiles;
    const int qq = valid ? q : 0;
    const int TT = A.T * A.T;
    const int pos = qq / TT;
    const int t_ = qq - pos * TT;
    const int ty = t_ / A.T, tx = t_ - ty * A.T;
    const int WP = A.WP, PP = A.PP, cout = A.cout;
    const int poff0 = 2 * ty * WP + 2 * tx;

    f32x16 M[16];
#pragma unroll
    for (int x = 0; x < 16; ++x)
#pragma unroll
        for (int r = 0; r < 16; ++r) M[x][r] = 0.0f;

    float bias_r[16];                                    // requested now, consumed in the epilogue
#pragma unroll
    for (int r = 0; r < 16; ++r) bias_r[r] = A.bias[ct * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh];
    const uint32_t wlane = (uint32_t)(kh * cout + ct * 32 + col);
    // k loops: operands ride a 4-deep register ring — the loads of phase p+3 are requested while phase p
    // computes (16 MFMAs = 1024 cycles per phase => >= 3072 cycles to land) — and the wide loads + the
    // transform adds of a phase are spread between its MFMAs by sched_group_barrier.
    {   // ---- 3x3 segment: phase = one cin pair (16 MFMAs) ----
        const char* __restrict__ inb = reinterpret_cast<const char*>(A.in);
        const uint32_t boff = (uint32_t)(pos * A.cin * PP + poff0 + kh * PP) * 4u;
        const int npairs = A.cin / 2;                    // multiple of 4
        f4u d0[4] = {}, d1[4] = {}, d2[4] = {}, d3[4] = {};
        auto load_d = [&](int c, f4u (&d)[4]) {
            c = c < npairs ? c : npairs - 1;             // tail: harmless re-load instead of a branch
            const char* ip = inb + (ptrdiff_t)(2 * c) * PP * 4;
#pragma unroll
            for (int r = 0; r < 4; ++r)                  // one patch row = 4 consecutive floats (4-byte aligned)
                d[r] = *reinterpret_cast<const f4u*>(ip + (ptrdiff_t)(r * WP) * 4 + (size_t)boff);
        };
        {
            const char* __restrict__ ub = reinterpret_cast<const char*>(A.u);
            const uint32_t woff = wlane * 64u;           // 16 floats per (kh, cout)
            float4 u0[4] = {}, u1[4] = {}, u2[4] = {}, u3[4] = {};
            auto load_u = [&](int c, float4 (&u)[4]) {
                    c = c < npairs ? c : npairs - 1;
                const char* up = ub + (size_t)c * 2 * cout * 64;
#pragma unroll
                for (int r = 0; r < 4; ++r) u[r] = *reinterpret_cast<const float4*>(up + (size_t)r * 16 + (size_t)woff);
            };
            // the input transform runs ONE PHASE AHEAD of the MFMAs that consume it (v of pair c+1 is computed
            // while pair c multiplies), so no MFMA waits on a VALU result issued just before it
            auto transform = [&](f4u (&d)[4], float (&v)[16]) {
                float t[16];
#pragma unroll
                for (int s_ = 0; s_ < 4; ++s_) {
                    t[0 + s_] = d[0][s_] - d[2][s_];
                    t[4 + s_] = d[1][s_] + d[2][s_];
                    t[8 + s_] = d[2][s_] - d[1][s_];
                    t[12 + s_] = d[1][s_] - d[3][s_];
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    v[i * 4 + 0] = t[i * 4 + 0] - t[i * 4 + 2];
                    v[i * 4 + 1] = t[i * 4 + 1] + t[i * 4 + 2];
                    v[i * 4 + 2] = t[i * 4 + 2] - t[i * 4 + 1];
                    v[i * 4 + 3] = t[i * 4 + 1] - t[i * 4 + 3];
                }
            };
            auto mfma16 = [&](float4 (&u)[4], float (&v)[16]) {
#pragma unroll
                for (int x = 0; x < 4; ++x) {
                    M[4 * x + 0] = __builtin_amdgcn_mfma_f32_32x32x2f32(u[x].x, v[4 * x + 0], M[4 * x + 0], 0, 0, 0);
                    M[4 * x + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(u[x].y, v[4 * x + 1], M[4 * x + 1], 0, 0, 0);
                    M[4 * x + 2] = __builtin_amdgcn_mfma_f32_32x32x2f32(u[x].z, v[4 * x + 2], M[4 * x + 2], 0, 0, 0);
                    M[4 * x + 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(u[x].w, v[4 * x + 3], M[4 * x + 3], 0, 0, 0);
                }
            };
            float vA[16], vB[16];
            auto phase = [&](int c_load, f4u (&dl)[4], float4 (&ul)[4], f4u (&dn)[4], float (&vn)[16], float4 (&uc)[4],
                             float (&vc)[16]) {
                load_d(c_load + 1, dl);        // a patch slot is free one phase earlier than a U slot: 4-phase lead
                load_u(c_load, ul);
                transform(dn, vn);
                mfma16(uc, vc);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);   // VALU
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // MFMA
                    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // VMEM read
                    __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            };
            load_d(0, d0); load_u(0, u0);
            load_d(1, d1); load_u(1, u1);
            load_d(2, d2); load_u(2, u2);
            __builtin_amdgcn_sched_barrier(0);
            load_d(3, d3);
            transform(d0, vA);
            __builtin_amdgcn_sched_barrier(0);
            for (int c = 0; c < npairs; c += 4) {
                phase(c + 3, d0, u3, d1, vB, u0, vA);     // multiply pair c, transform pair c+1, load U c+3 / patch c+4
                phase(c + 4, d1, u0, d2, vA, u1, vB);
                phase(c + 5, d2, u1, d3, vB, u2, vA);
                phase(c + 6, d3, u2, d0, vA, u3, vB);
            }
        }
    }
    if (A.in2) {   // ---- 1x1 projection segment: phase = 4 cin pairs (16 MFMAs) ----
        const char* __restrict__ inb = reinterpret_cast<const char*>(A.in2);
        const char* __restrict__ ub = reinterpret_cast<const char*>(A.u2);
        const uint32_t boff = (uint32_t)(pos * A.cin2 * PP + poff0 + kh * PP + WP + 1) * 4u;   // patch centre (1,1)
        const uint32_t woff = wlane * 16u;               // 4 floats per (kh, cout)
        const int nph = A.cin2 / 8;                      // multiple of 4
        f2u d0[4][2], d1[4][2], d2[4][2], d3[4][2];
        float4 u0[4], u1[4], u2[4], u3[4];
        auto load_ph = [&](int q, f2u (&d)[4][2], float4 (&u)[4]) {
            q = q < nph ? q : nph - 1;
#pragma unroll
            for (int h = 0; h < 4; ++h) {
                const char* ip = inb + (ptrdiff_t)(2 * (4 * q + h)) * PP * 4;
                const char* up = ub + (size_t)(4 * q + h) * 2 * cout * 16;
                d[h][0] = *reinterpret_cast<const f2u*>(ip + (size_t)boff);
                d[h][1] = *reinterpret_cast<const f2u*>(ip + (ptrdiff_t)WP * 4 + (size_t)boff);
                u[h] = *reinterpret_cast<const float4*>(up + (size_t)woff);
            }
        };
        auto compute_ph = [&](f2u (&d)[4][2], float4 (&u)[4]) {
#pragma unroll
            for (int h = 0; h < 4; ++h) {
                const float t11 = d[h][0][0] + d[h][1][0], t12 = d[h][0][1] + d[h][1][1];   // (B^T d) rows 1,2, cols 1,2
                const float t21 = d[h][1][0] - d[h][0][0], t22 = d[h][1][1] - d[h][0][1];
                M[5] = __builtin_amdgcn_mfma_f32_32x32x2f32(u[h].x, t11 + t12, M[5], 0, 0, 0);
                M[6] = __builtin_amdgcn_mfma_f32_32x32x2f32(u[h].y, t12 - t11, M[6], 0, 0, 0);
                M[9] = __builtin_amdgcn_mfma_f32_32x32x2f32(u[h].z, t21 + t22, M[9], 0, 0, 0);
                M[10] = __builtin_amdgcn_mfma_f32_32x32x2f32(u[h].w, t22 - t21, M[10], 0, 0, 0);
            }
        };
        auto phase = [&](int q_load, f2u (&dl)[4][2], float4 (&ul)[4], f2u (&dc)[4][2], float4 (&uc)[4]) {
            load_ph(q_load, dl, ul);
            compute_ph(dc, uc);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 3, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        load_ph(0, d0, u0);
        load_ph(1, d1, u1);
        load_ph(2, d2, u2);
        __builtin_amdgcn_sched_barrier(0);
        for (int q = 0; q < nph; q += 4) {
            phase(q + 3, d3, u3, d0, u0);
            phase(q + 4, d0, u0, d1, u1);
            phase(q + 5, d1, u1, d2, u2);
            phase(q + 6, d2, u2, d3, u3);
        }
    }
    // ---- output transform Y = A^T M A, bias, ELU, store the 2x2 tile ----
    // (bias values were requested before the k loops; each tile row goes out as one 8-byte store, so the
    // lanes of a board row write one contiguous run)
    const bool ok1y = 2 * ty + 1 < A.S, ok1x = 2 * tx + 1 < A.S;      // rows/cols 2t are always inside the board
    float* const obase = A.out + (size_t)pos * cout * PP + poff0 + WP + 1;  // output pixel (2ty, 2tx) of cout 0
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int co = ct * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
        const float bv = bias_r[r];
        float s0[4], s1[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            s0[j] = M[0 + j][r] + M[4 + j][r] + M[8 + j][r];
            s1[j] = M[4 + j][r] - M[8 + j][r] - M[12 + j][r];
        }
        f2u y0, y1;
        y0[0] = elu1(s0[0] + s0[1] + s0[2] + bv); y0[1] = elu1(s0[1] - s0[2] - s0[3] + bv);
        y1[0] = elu1(s1[0] + s1[1] + s1[2] + bv); y1[1] = elu1(s1[1] - s1[2] - s1[3] + bv);
        float* o = obase + (size_t)co * PP;
        if (valid) {
            if (ok1x) {
                *reinterpret_cast<f2u*>(o) = y0;
                if (ok1y) *reinterpret_cast<f2u*>(o + WP) = y1;
            } else {
                o[0] = y0[0];
                if (ok1y) o[WP] = y1[0];
            }
        }
    }
}

__global__ __launch_bounds__(256, 1) void af_conv_wino(WinoArgs A) { wino_task(A, (int)blockIdx.x); }

// (r1 experiment, removed: af_conv_wino2 split the 16 Winograd-domain accumulators over a wave pair —
// 128 AGPRs per wave, two waves per SIMD, partial outputs swapped through LDS.  Correct, but hipcc spilled
// 288 B/lane at the 256-register budget and the pair doubles the patch loads: 4.94 ms vs 3.96 ms.)

// stem: conv 5x5 SAME 3->32 + ELU (network.py:63), one block per position, one thread per pixel
__global__ __launch_bounds__(256) void af_stem_conv(const float* __restrict__ planes, const float* __restrict__ w /*[75][32]*/,
                                                    const float* __restrict__ bias, float* __restrict__ out, int S, int WP, int PP) {
    __shared__ float sx[3 * 20 * 20];
    __shared__ float sw[75 * 32];
    __shared__ float sb[32];
    const int b = blockIdx.x, t = threadIdx.x, HW = S * S, XP = S + 4;
    for (int i = t; i < 3 * XP * XP; i += blockDim.x) sx[i] = 0.0f;
    for (int i = t; i < 75 * 32; i += blockDim.x) sw[i] = w[i];
    if (t < 32) sb[t] = bias[t];
    __syncthreads();
    for (int i = t; i < 3 * HW; i += blockDim.x) {
        const int c = i / HW, p = i - c * HW, y = p / S, x = p - y * S;
        sx[c * XP * XP + (y + 2) * XP + x + 2] = planes[(size_t)b * 3 * HW + i];
    }
    __syncthreads();
    if (t >= HW) return;
    const int y = t / S, x = t - y * S;
    float acc[32];
#pragma unroll
    for (int co = 0; co < 32; ++co) acc[co] = sb[co];
    for (int ky = 0; ky < 5; ++ky)
        for (int kx = 0; kx < 5; ++kx)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float xv = sx[c * XP * XP + (y + ky) * XP + x + kx];
                const float* wr = sw + ((ky * 5 + kx) * 3 + c) * 32;
#pragma unroll
                for (int co = 0; co < 32; ++co) acc[co] = fmaf(xv, wr[co], acc[co]);
            }
    float* o = out + (size_t)b * 32 * PP + (y + 1) * WP + x + 1;
#pragma unroll
    for (int co = 0; co < 32; ++co) o[(size_t)co * PP] = elu1(acc[co]);
}

// value head: 1x1 conv 32->4 + ELU, fc 4*HW->64 + ELU, fc 64->1, tanh(x/2)  (network.py:70-76,163).
// VPB positions per workgroup: the fc1 weights (4*HW x 64 floats = 124 KB at 11x11) are streamed from L2 once per
// workgroup, not once per position (r1: one position per block = 0.5 GB of L2 reads per forward for 0.2 GMAC).
constexpr int VPB = 8;
__global__ __launch_bounds__(256) void af_value_head(const float* __restrict__ in /*[b][32][PP]*/, const float* __restrict__ wc /*[32][4]*/,
                                                     const float* __restrict__ bc, const float* __restrict__ w1 /*[4HW][64]*/,
                                                     const float* __restrict__ b1, const float* __restrict__ w2 /*[64]*/,
                                                     const float* __restrict__ b2, float* __restrict__ value, int batch, int S, int WP, int PP) {
    __shared__ float sh[VPB][4 * 256];
    __shared__ float swc[32 * 4];
    __shared__ float s64[VPB][64];
    const int b0 = blockIdx.x * VPB, t = threadIdx.x, HW = S * S;
    if (t < 128) swc[t] = wc[t];
    __syncthreads();
    for (int it = t; it < VPB * HW; it += 256) {                     // 1x1 conv + ELU: item = (position q, pixel)
        const int q = it / HW, px = it - q * HW, y = px / S, x = px - y * S;
        const int b = b0 + q < batch ? b0 + q : batch - 1;
        const float* p = in + (size_t)b * 32 * PP + (y + 1) * WP + x + 1;
        float a0 = bc[0], a1 = bc[1], a2 = bc[2], a3 = bc[3];
#pragma unroll 8
        for (int c = 0; c < 32; ++c) {
            const float xv = p[(size_t)c * PP];
            a0 = fmaf(xv, swc[c * 4 + 0], a0); a1 = fmaf(xv, swc[c * 4 + 1], a1);
            a2 = fmaf(xv, swc[c * 4 + 2], a2); a3 = fmaf(xv, swc[c * 4 + 3], a3);
        }
        sh[q][0 * HW + px] = elu1(a0); sh[q][1 * HW + px] = elu1(a1); sh[q][2 * HW + px] = elu1(a2); sh[q][3 * HW + px] = elu1(a3);
    }
    __syncthreads();
    {                                                                // fc1: thread = (output o, pair of positions pg)
        const int o = t & 63, pg = t >> 6;
        float acc0 = b1[o], acc1 = acc0;
        const float* x0 = sh[2 * pg];
        const float* x1 = sh[2 * pg + 1];
        for (int k = 0; k < 4 * HW; ++k) {                           // same k order as one-position-per-block: same bits
            const float w = w1[(size_t)k * 64 + o];
            acc0 = fmaf(x0[k], w, acc0);
            acc1 = fmaf(x1[k], w, acc1);
        }
        s64[2 * pg][o] = elu1(acc0) * w2[o];
        s64[2 * pg + 1][o] = elu1(acc1) * w2[o];
    }
    __syncthreads();
    if (t < VPB && b0 + t < batch) {
        float s = b2[0];
        for (int j = 0; j < 64; ++j) s += s64[t][j];
        value[b0 + t] = tanhf(s * 0.5f);
    }
}

// policy head: 1x1 conv 32->16 + ELU, fc 16*HW->HW, softmax  (network.py:82-88); PPB positions per block
template <int PPB>
__global__ __launch_bounds__(256) void af_policy_head(const float* __restrict__ in /*[b][32][PP]*/, const float* __restrict__ wc /*[32][16]*/,
                                                      const float* __restrict__ bc, const float* __restrict__ wf /*[16HW][HW]*/,
                                                      const float* __restrict__ bf, float* __restrict__ policy, int batch, int S, int WP, int PP) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int HW = S * S, t = threadIdx.x, b0 = blockIdx.x * PPB;
    float* sh = smem;                       // [16*HW][PPB]
    float* swc = smem + 16 * HW * PPB;      // [32][16]
    float* red = swc + 512;                 // [4 waves][PPB]
    for (int i = t; i < 512; i += blockDim.x) swc[i] = wc[i];
    __syncthreads();
    if (t < HW) {
        const int y = t / S, x = t - y * S;
        for (int p = 0; p < PPB; ++p) {
            const int b = b0 + p < batch ? b0 + p : batch - 1;
            const float* src = in + (size_t)b * 32 * PP + (y + 1) * WP + x + 1;
            float a[16];
#pragma unroll
            for (int co = 0; co < 16; ++co) a[co] = bc[co];
            for (int c = 0; c < 32; ++c) {
                const float xv = src[(size_t)c * PP];
#pragma unroll
                for (int co = 0; co < 16; ++co) a[co] = fmaf(xv, swc[c * 16 + co], a[co]);
            }
#pragma unroll
            for (int co = 0; co < 16; ++co) sh[(co * HW + t) * PPB + p] = elu1(a[co]);
        }
    }
    __syncthreads();
    // dense layer: thread (ks, tj) accumulates logit tj over the ks-th slice of the 16*HW inputs for PPB
    // positions at once (weights streamed once per block); slices are summed through LDS in fixed order
    const int JW = HW <= 128 ? 128 : 256, KS = blockDim.x / JW;
    const int ks = t / JW, tj = t - ks * JW;
    const bool owner = ks == 0 && tj < HW;
    float acc[PPB];
    const int j = tj < HW ? tj : HW - 1;
#pragma unroll
    for (int p = 0; p < PPB; ++p) acc[p] = ks == 0 ? bf[j] : 0.0f;
    const int K = 16 * HW, k0 = ks * (K / KS), k1 = ks == KS - 1 ? K : k0 + K / KS;
#pragma unroll 8
    for (int k = k0; k < k1; ++k) {
        const float wv = wf[(size_t)k * HW + j];
        const float4* xr = reinterpret_cast<const float4*>(sh + (size_t)k * PPB);
#pragma unroll
        for (int q = 0; q < PPB / 4; ++q) {
            const float4 xv = xr[q];
            acc[4 * q + 0] = fmaf(xv.x, wv, acc[4 * q + 0]); acc[4 * q + 1] = fmaf(xv.y, wv, acc[4 * q + 1]);
            acc[4 * q + 2] = fmaf(xv.z, wv, acc[4 * q + 2]); acc[4 * q + 3] = fmaf(xv.w, wv, acc[4 * q + 3]);
        }
    }
    if (KS > 1) {
        __syncthreads();                                  // everyone is done reading sh: reuse it for the partials
        if (ks > 0) {
#pragma unroll
            for (int p = 0; p < PPB; ++p) sh[((ks - 1) * PPB + p) * JW + tj] = acc[p];
        }
        __syncthreads();
        if (ks == 0) {
            for (int s_ = 1; s_ < KS; ++s_)
#pragma unroll
                for (int p = 0; p < PPB; ++p) acc[p] += sh[((s_ - 1) * PPB + p) * JW + tj];
        }
    }
    // softmax over the HW logits of each position
    const int wave = t >> 6, lane = t & 63, nw = blockDim.x >> 6;
    float m[PPB];
#pragma unroll
    for (int p = 0; p < PPB; ++p) {
        float v = owner ? acc[p] : -3.0e38f;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) v = fmaxf(v, __shfl_xor(v, off));
        if (lane == 0) red[wave * PPB + p] = v;
    }
    __syncthreads();
#pragma unroll
    for (int p = 0; p < PPB; ++p) {
        float v = red[p];
        for (int w_ = 1; w_ < nw; ++w_) v = fmaxf(v, red[w_ * PPB + p]);
        m[p] = v;
    }
    __syncthreads();
    float e[PPB];
#pragma unroll
    for (int p = 0; p < PPB; ++p) {
        e[p] = owner ? expf(acc[p] - m[p]) : 0.0f;
        float v = e[p];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) v = v + __shfl_xor(v, off);
        if (lane == 0) red[wave * PPB + p] = v;
    }
    __syncthreads();
    if (owner) {
#pragma unroll
        for (int p = 0; p < PPB; ++p) {
            float s_ = red[p];
            for (int w_ = 1; w_ < nw; ++w_) s_ += red[w_ * PPB + p];
            if (b0 + p < batch) policy[(size_t)(b0 + p) * HW + tj] = e[p] / s_;
        }
    }
}

// policy head on the matrix cores (boards up to 11x11), 16 positions per workgroup = one workgroup per CU
// at 4096 leaves.  All three phases are matrix shaped, v_mfma_f32_16x16x4_f32 throughout:
//   phase 1  1x1 conv 32->16 + ELU: D[co][item] = sum_ci wc[ci][co] x[ci][item] over tiles of 16 items
//            (item = position*HW + pixel); A = the lane's 8 weights (registers), B = one dword per k-step
//            straight from the padded planes, two tiles in flight; result -> X[pos][co*HW + pix] in LDS
//            (the NCHW flatten order of network.py:83-84; row stride = 2 mod 32: conflict-free B reads later)
//   phase 2  logits^T[j][pos] = sum_k Wfc[k][j] X[pos][k]: A = 16 logits x 4 k from global (64-byte runs,
//            requested 16 k-steps = 1024 MFMA cycles ahead), B = 4 k x 16 positions from LDS; each wave owns
//            two 16-logit tiles
//   phase 3  logits -> LDS, softmax by 16-lane groups, store
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void af_policy_head_mfma(const float* __restrict__ in /*[b][32][PP]*/, const float* __restrict__ wc /*[32][16]*/,
                                                           const float* __restrict__ bc, const float* __restrict__ wf /*[16HW][HW]*/,
                                                           const float* __restrict__ bf, float* __restrict__ policy, int batch, int S, int WP, int PP) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int HW = S * S, K = 16 * HW, t = threadIdx.x, b0 = blockIdx.x * 16;
    const int KS = K + ((2 - K % 32) + 32) % 32;
    float* X = smem;                                       // [16][KS]
    const int lane = t & 63, wave = t >> 6, i16 = lane & 15, kq = lane >> 4;
    {   // ---- phase 1 ----
        float aw[8];                                       // A fragments: wc[ci = 4s+kq][co = i16]
#pragma unroll
        for (int s_ = 0; s_ < 8; ++s_) aw[s_] = wc[(4 * s_ + kq) * 16 + i16];
        float bias4[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) bias4[r] = bc[kq * 4 + r];
        const int nitems = 16 * HW, ntiles = (nitems + 15) / 16;
        auto item_src = [&](int tile, int& p, int& pix) -> const float* {
            int it = tile * 16 + i16;
            it = it < nitems ? it : nitems - 1;
            p = it / HW;
            pix = it - p * HW;
            const int y = pix / S, x = pix - y * S;
            const int b = b0 + p < batch ? b0 + p : batch - 1;
            return in + (size_t)b * 32 * PP + (size_t)kq * PP + (y + 1) * WP + x + 1;
        };
        for (int tile = wave; tile < ntiles; tile += 8) {   // two tiles (this one and tile+4) in flight
            int p0, x0, p1, x1;
            const float* s0 = item_src(tile, p0, x0);
            const bool has1 = tile + 4 < ntiles;
            const float* s1 = item_src(has1 ? tile + 4 : tile, p1, x1);
            float bx0[8], bx1[8];
#pragma unroll
            for (int s_ = 0; s_ < 8; ++s_) { bx0[s_] = s0[(size_t)(4 * s_) * PP]; bx1[s_] = s1[(size_t)(4 * s_) * PP]; }
            f32x4 d0 = {0, 0, 0, 0}, d1 = {0, 0, 0, 0};
#pragma unroll
            for (int s_ = 0; s_ < 8; ++s_) {
                d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(aw[s_], bx0[s_], d0, 0, 0, 0);
                d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(aw[s_], bx1[s_], d1, 0, 0, 0);
            }
            const bool ok0 = tile * 16 + i16 < nitems, ok1 = has1 && (tile + 4) * 16 + i16 < nitems;
#pragma unroll
            for (int r = 0; r < 4; ++r) {                  // D: col = lane&15 (item), row = kq*4 + r (co)
                if (ok0) X[p0 * KS + (kq * 4 + r) * HW + x0] = elu1(d0[r] + bias4[r]);
                if (ok1) X[p1 * KS + (kq * 4 + r) * HW + x1] = elu1(d1[r] + bias4[r]);
            }
        }
    }
    __syncthreads();
    // ---- phase 2 ----
    const int j0 = wave * 32 + i16, j1 = j0 + 16;
    const float* w0 = wf + (size_t)kq * HW + (j0 < HW ? j0 : HW - 1);
    const float* w1 = wf + (size_t)kq * HW + (j1 < HW ? j1 : HW - 1);
    const float* xb = X + i16 * KS + kq;
    f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
    constexpr int PF = 16;
    float a0[PF], a1[PF];
    const int nsteps = K / 4;                              // = 4*HW (484 at 11x11: not a multiple of PF -> tail below)
#pragma unroll
    for (int q = 0; q < PF; ++q) { a0[q] = w0[(size_t)q * 4 * HW]; a1[q] = w1[(size_t)q * 4 * HW]; }
    int s0 = 0;
    for (; s0 + PF <= nsteps; s0 += PF) {
#pragma unroll
        for (int q = 0; q < PF; ++q) {
            const int s_ = s0 + q;
            const float bx = xb[s_ * 4];
            const float ca0 = a0[q], ca1 = a1[q];
            const int sn = s_ + PF < nsteps ? s_ + PF : nsteps - 1;   // ring slot q now waits for step s_+PF
            a0[q] = w0[(size_t)sn * 4 * HW];
            a1[q] = w1[(size_t)sn * 4 * HW];
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(ca0, bx, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(ca1, bx, acc1, 0, 0, 0);
        }
    }
#pragma unroll
    for (int q = 0; q < PF; ++q) {                         // tail: slots 0..rem-1 hold steps s0..nsteps-1
        if (s0 + q < nsteps) {
            const float bx = xb[(s0 + q) * 4];
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[q], bx, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[q], bx, acc1, 0, 0, 0);
        }
    }
    __syncthreads();                                       // X is dead: reuse it for the logits [16 pos][128]
    float* Lg = smem;
#pragma unroll
    for (int r = 0; r < 4; ++r) {                          // D: col = lane&15 (position), row = kq*4 + r
        const int ja = wave * 32 + kq * 4 + r, jb = ja + 16;
        Lg[i16 * 128 + ja] = acc0[r] + (ja < HW ? bf[ja] : 0.0f);
        Lg[i16 * 128 + jb] = acc1[r] + (jb < HW ? bf[jb] : 0.0f);
    }
    __syncthreads();
    // ---- phase 3 ----
    const int p = t >> 4, sub = t & 15;                    // 16 threads per position, 8 logits each
    float v[8], m = -3.0e38f;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int j = sub + 16 * q;
        v[q] = j < HW ? Lg[p * 128 + j] : -3.0e38f;
        m = fmaxf(m, v[q]);
    }
#pragma unroll
    for (int off = 8; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    float sum = 0.0f;
#pragma unroll
    for (int q = 0; q < 8; ++q) { v[q] = (sub + 16 * q) < HW ? expf(v[q] - m) : 0.0f; sum += v[q]; }
#pragma unroll
    for (int off = 8; off >= 1; off >>= 1) sum += __shfl_xor(sum, off);
    if (b0 + p < batch) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int j = sub + 16 * q;
            if (j < HW) policy[(size_t)(b0 + p) * HW + j] = v[q] / sum;
        }
    }
}

// ----------------------------------------------------------------------------------------------
// host side
// ----------------------------------------------------------------------------------------------
#define NET_HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "[af_net] %s failed: %s\n", #x, hipGetErrorString(e_)); return AF_NET_ERR_HIP; } } while (0)

struct Block { const char* name; int cin, cout; };
static const Block kBlocks[5] = {{"bone/block1", 32, 64}, {"bone/block2", 64, 128}, {"value/block3", 128, 32},
                                 {"policy/block4", 128, 64}, {"policy/block5", 64, 32}};

struct af_net {
    int S, HW, WP, PP, max_batch, device;
    bool ready = false;
    std::map<std::string, std::vector<float>> vars;
    std::map<std::string, size_t> expect;
    std::vector<void*> allocs;
    // device weights
    float *stem_w, *stem_b;
    float *conv1_b[5], *sum_b[5];
    float *wino1_u[5], *wino2_u[5], *winor_u[5];
    int T;
    hipStream_t branch_stream = nullptr;
    hipEvent_t ev_trunk = nullptr, ev_value = nullptr;
    float *vc_w, *vc_b, *v1_w, *v1_b, *v2_w, *v2_b, *pc_w, *pc_b, *pf_w, *pf_b;
    // activations [max_batch][C][PP]
    float *f0, *g[5], *o[5];
    f16s_net* f16s = nullptr;     // 11x11 boards: the fp16 split-operand convolution path (af_conv_f16s.hip)
};

static int pad32(int c) { return (c + 31) / 32 * 32; }

template <typename T>
static int net_alloc(af_net* n, T** p, size_t count, bool zero) {
    void* q = nullptr;
    NET_HIP_OK(hipMalloc(&q, count * sizeof(T)));
    if (zero) NET_HIP_OK(hipMemset(q, 0, count * sizeof(T)));
    n->allocs.push_back(q);
    *p = (T*)q;
    return AF_NET_OK;
}
static int net_upload(af_net* n, float** p, const std::vector<float>& h) {
    int rc = net_alloc(n, p, h.size(), false);
    if (rc) return rc;
    NET_HIP_OK(hipMemcpy(*p, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    return AF_NET_OK;
}

// Winograd weights U = G g G^T per (cin, cout), G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]], packed
// [cin/2][2][cout][npos] (one lane's npos values contiguous).  taps == 9: all 16 (xi,nu); taps == 1: the centre-tap kernel of a 1x1
// projection, whose only non-zero positions are (1,1),(1,2),(2,1),(2,2).
static std::vector<float> pack_wino(const std::vector<float>& w, int taps, int cin, int cout) {
    static const double G[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
    static const int kCentre[4] = {5, 6, 9, 10};
    const int npos = taps == 9 ? 16 : 4;
    std::vector<float> out((size_t)cin * npos * cout, 0.0f);
    for (int c = 0; c < cin; ++c)
        for (int co = 0; co < cout; ++co) {
            double g[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
            if (taps == 9) {
                for (int t = 0; t < 9; ++t) g[t / 3][t % 3] = w[((size_t)t * cin + c) * cout + co];
            } else {
                g[1][1] = w[(size_t)c * cout + co];
            }
            for (int p = 0; p < npos; ++p) {
                const int x = taps == 9 ? p : kCentre[p];
                const int xi = x / 4, nu = x % 4;
                double u = 0.0;
                for (int i = 0; i < 3; ++i)
                    for (int j = 0; j < 3; ++j) u += G[xi][i] * g[i][j] * G[nu][j];
                out[((((size_t)(c / 2) * 2) + (c & 1)) * cout + co) * npos + p] = (float)u;
            }
        }
    return out;
}
static std::vector<float> pad_bias(const std::vector<float>& a, const std::vector<float>* b, int cout) {
    std::vector<float> out(pad32(cout), 0.0f);
    for (int i = 0; i < cout; ++i) out[i] = a[i] + (b ? (*b)[i] : 0.0f);
    return out;
}

extern "C" {

const char* af_net_strerror(int code) {
    switch (code) {
        case AF_NET_OK: return "ok";
        case AF_NET_ERR_ARG: return "bad argument";
        case AF_NET_ERR_HIP: return "HIP runtime error";
        case AF_NET_ERR_NAME: return "unknown variable name or wrong element count";
        case AF_NET_ERR_STATE: return "network not finalized or variables missing";
        default: return "unknown error";
    }
}

int af_net_create(int32_t S, int32_t max_batch, int32_t device, af_net** out) {
    if (!out || S < 3 || S > 16 || max_batch < 1) return AF_NET_ERR_ARG;
    NET_HIP_OK(hipSetDevice(device));
    af_net* n = new af_net();
    n->S = S; n->HW = S * S;
    n->T = (S + 1) / 2;                      // 2x2 output tiles per board side (Winograd)
    n->WP = 2 * n->T + 2;                    // plane padded so every tile's 4x4 patch is in bounds
    n->PP = (n->WP * n->WP + 15) / 16 * 16;
    n->max_batch = max_batch; n->device = device;
    // the conv kernels address activations with 32-bit byte offsets: the widest buffer must stay below 4 GiB
    if ((uint64_t)max_batch * 128u * (uint64_t)n->PP * 4u >= (1ull << 32)) { delete n; return AF_NET_ERR_ARG; }
    const size_t HW = n->HW;
    n->expect["bone/conv1/kernel"] = 75 * 32; n->expect["bone/conv1/bias"] = 32;
    for (const Block& b : kBlocks) {
        const std::string s = b.name;
        n->expect[s + "_res/kernel"] = (size_t)b.cin * b.cout; n->expect[s + "_res/bias"] = b.cout;
        n->expect[s + "_conv1/kernel"] = (size_t)9 * b.cin * b.cout; n->expect[s + "_conv1/bias"] = b.cout;
        n->expect[s + "_conv2/kernel"] = (size_t)9 * b.cout * b.cout; n->expect[s + "_conv2/bias"] = b.cout;
    }
    n->expect["value/conv/kernel"] = 32 * 4; n->expect["value/conv/bias"] = 4;
    n->expect["value/fc1/kernel"] = 4 * HW * 64; n->expect["value/fc1/bias"] = 64;
    n->expect["value/fc2/kernel"] = 64; n->expect["value/fc2/bias"] = 1;
    n->expect["policy/conv/kernel"] = 32 * 16; n->expect["policy/conv/bias"] = 16;
    n->expect["policy/fc/kernel"] = 16 * HW * HW; n->expect["policy/fc/bias"] = HW;
    *out = n;
    return AF_NET_OK;
}

void af_net_destroy(af_net* n) {
    if (!n) return;
    (void)hipSetDevice(n->device);
    for (void* p : n->allocs) (void)hipFree(p);
    f16s_destroy(n->f16s);
    if (n->branch_stream) (void)hipStreamDestroy(n->branch_stream);
    if (n->ev_trunk) (void)hipEventDestroy(n->ev_trunk);
    if (n->ev_value) (void)hipEventDestroy(n->ev_value);
    delete n;
}

int af_net_set_variable(af_net* n, const char* name, const float* data, int64_t count) {
    if (!n || !name || !data) return AF_NET_ERR_ARG;
    auto it = n->expect.find(name);
    if (it == n->expect.end() || (int64_t)it->second != count) return AF_NET_ERR_NAME;
    n->vars[name].assign(data, data + count);
    n->ready = false;
    return AF_NET_OK;
}

int af_net_finalize(af_net* n) {
    if (!n) return AF_NET_ERR_ARG;
    for (auto& kv : n->expect) if (!n->vars.count(kv.first)) return AF_NET_ERR_STATE;
    NET_HIP_OK(hipSetDevice(n->device));
    for (void* p : n->allocs) (void)hipFree(p);
    n->allocs.clear();
    f16s_destroy(n->f16s);
    n->f16s = nullptr;
    if (f16s_supported(n->S) && f16s_create(&n->f16s, n->S, n->max_batch, n->device, n->vars) != 0) return AF_NET_ERR_HIP;
    auto& V = n->vars;
    int rc = AF_NET_OK;
#define UP(dst, vec) if (!rc) rc = net_upload(n, &n->dst, (vec))
    UP(stem_w, V["bone/conv1/kernel"]); UP(stem_b, V["bone/conv1/bias"]);
    for (int i = 0; i < 5; ++i) {
        const Block& b = kBlocks[i];
        const std::string s = b.name;
        UP(conv1_b[i], pad_bias(V[s + "_conv1/bias"], nullptr, b.cout));
        UP(sum_b[i], pad_bias(V[s + "_conv2/bias"], &V[s + "_res/bias"], b.cout));
        UP(wino1_u[i], pack_wino(V[s + "_conv1/kernel"], 9, b.cin, b.cout));
        UP(wino2_u[i], pack_wino(V[s + "_conv2/kernel"], 9, b.cout, b.cout));
        UP(winor_u[i], pack_wino(V[s + "_res/kernel"], 1, b.cin, b.cout));
    }
    UP(vc_w, V["value/conv/kernel"]); UP(vc_b, V["value/conv/bias"]); UP(v1_w, V["value/fc1/kernel"]);
    UP(v1_b, V["value/fc1/bias"]); UP(v2_w, V["value/fc2/kernel"]); UP(v2_b, V["value/fc2/bias"]);
    UP(pc_w, V["policy/conv/kernel"]); UP(pc_b, V["policy/conv/bias"]); UP(pf_w, V["policy/fc/kernel"]);
    UP(pf_b, V["policy/fc/bias"]);
#undef UP
    const size_t plane = (size_t)n->max_batch * n->PP;
    if (!rc) rc = net_alloc(n, &n->f0, plane * 32, true);
    for (int i = 0; i < 5 && !rc; ++i) {
        rc = net_alloc(n, &n->g[i], plane * kBlocks[i].cout, true);
        if (!rc) rc = net_alloc(n, &n->o[i], plane * kBlocks[i].cout, true);
    }
    if (rc) return rc;
    NET_HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(af_policy_head_mfma), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   160 * 1024));
    if (!n->branch_stream) {
        NET_HIP_OK(hipStreamCreateWithFlags(&n->branch_stream, hipStreamNonBlocking));
        NET_HIP_OK(hipEventCreateWithFlags(&n->ev_trunk, hipEventDisableTiming));
        NET_HIP_OK(hipEventCreateWithFlags(&n->ev_value, hipEventDisableTiming));
    }
    n->ready = true;
    return AF_NET_OK;
}

}  // extern "C"

static int g_f16s_abl = 0;
static int g_wino = 5;   // 5: fp16 split-operand implicit GEMM on 11x11 / 15x15 boards (default; other sizes run 1); 1: af_conv_wino (fp32 MFMA, Winograd)

static void launch_wino(hipStream_t st, const af_net* n, int batch, const float* in, const float* u, int cin,
                        const float* in2, const float* u2, int cin2, const float* bias, float* out, int cout) {
    WinoArgs a;
    a.in = in; a.u = u; a.in2 = in2; a.u2 = u2; a.bias = bias; a.out = out;
    a.cin = cin; a.cin2 = cin2; a.cout = cout; a.T = n->T; a.S = n->S; a.WP = n->WP; a.PP = n->PP;
    a.ntiles = batch * n->T * n->T;
    const int ntb = (a.ntiles + 31) / 32;
    const int ngrp8 = (((ntb + 3) / 4) + 7) / 8 * 8;      // tile-block groups, padded to the 8-XCD interleave
    hipLaunchKernelGGL(af_conv_wino, dim3(ngrp8 * (cout / 32)), dim3(256), 0, st, a);
}


static int g_phead = 1;          // 1: af_policy_head_mfma for boards up to 11x11; 0: VALU head
static int g_branch = 1;         // 1: value branch on a side stream
static int g_fhead = 1;          // 1: split-operand path computes the heads itself (fused 1x1 conv + MFMA dense layers); 0: af_value_head / af_policy_head_mfma

// forward pass of positions [b0, b0+batch) on stream st
static int forward_range(af_net* n, hipStream_t st, const float* planes_all, int b0, int batch, float* policy_all, float* value_all) {
    const int S = n->S, HW = n->HW, WP = n->WP, PP = n->PP;
    const float* planes = planes_all + (size_t)b0 * 3 * HW;
    float* policy = policy_all + (size_t)b0 * HW;
    float* value = value_all + b0;
    const size_t po = (size_t)b0 * PP;
    float* f0 = n->f0 + po * 32;
    float* g[5];
    float* o[5];
    for (int i = 0; i < 5; ++i) { g[i] = n->g[i] + po * kBlocks[i].cout; o[i] = n->o[i] + po * kBlocks[i].cout; }
    const bool split16 = g_wino == 5 && n->f16s != nullptr;     // (board sizes without the split-operand path run the fp32 Winograd path)
    const bool fhead = split16 && g_fhead;                 // heads fused into the split-operand path
    if (split16) {
        f16s_set_ablation(n->f16s, g_f16s_abl);
        if (fhead && f16s_small_forward_ok(n->f16s, batch)) {
            // <= 8 positions on 11x11: one launch of dataflow roles instead of nine dependent launches (af_conv_f16s.hip, r6)
            if (f16s_small_forward(n->f16s, st, planes, batch, value, policy)) return AF_NET_ERR_HIP;
            return AF_NET_OK;
        }
        if (f16s_trunk(n->f16s, st, planes, batch)) return AF_NET_ERR_HIP;
    } else {
        hipLaunchKernelGGL(af_stem_conv, dim3(batch), dim3(256), 0, st, planes, n->stem_w, n->stem_b, f0, S, WP, PP);
    }
    if (fhead && f16s_small_branches_ok(n->f16s, batch)) {
        // small batches: fork and join of the value branch's side stream cost more than its kernels (af_conv_f16s.hip)
        if (f16s_small_branches(n->f16s, st, batch, value, policy)) return AF_NET_ERR_HIP;
        NET_HIP_OK(hipGetLastError());
        return AF_NET_OK;
    }
    const float* block_in[5] = {f0, o[0], o[1], o[1], o[3]};
    // the value branch (block3 + head) only depends on the trunk output o[1]: it runs on a side stream,
    // concurrently with the policy branch (blocks 4,5 + head), filling the SIMDs the 32/64-wide layers leave idle
    // (r5: on the split-operand path the side stream is used by neither board size any more — 11x11 runs the value branch as workgroup
    //  classes of the policy branch's launches (f16s_small_branches above); 15x15 serialised is 0.7 % faster than forked, 2.653 vs 2.672 ms:
    //  fork and join cost more than the tail filling returns.  af_net_tune(4, 2) forces the side stream for A/B.)
    hipStream_t vs = st;
    if (g_branch && n->branch_stream && (!split16 || g_branch == 2)) {
        vs = n->branch_stream;
    }
    for (int i = split16 ? 2 : 0; i < 5; ++i) {
        const Block& b = kBlocks[i];
        hipStream_t st_main = st;
        if (i == 2 && vs != st_main) {
            NET_HIP_OK(hipEventRecord(n->ev_trunk, st_main));
            NET_HIP_OK(hipStreamWaitEvent(vs, n->ev_trunk, 0));
        }
        hipStream_t st = (i == 2) ? vs : st_main;
        if (split16) {
            if (i == 2 && f16s_value_branch(n->f16s, st, batch, o[2], WP, PP, fhead ? value : nullptr)) return AF_NET_ERR_HIP;
            if (i == 4 && f16s_policy_branch(n->f16s, st, batch, o[4], WP, PP, fhead ? policy : nullptr)) return AF_NET_ERR_HIP;
        } else {
            // conv1 3x3 + ELU (network.py:54); conv2 3x3 (+) 1x1 projection, add, ELU (network.py:53,55,56)
            launch_wino(st, n, batch, block_in[i], n->wino1_u[i], b.cin, nullptr, nullptr, 0, n->conv1_b[i], g[i], b.cout);
            launch_wino(st, n, batch, g[i], n->wino2_u[i], b.cout, block_in[i], n->winor_u[i], b.cin, n->sum_b[i], o[i], b.cout);
        }
        if (i == 2) {
            if (!fhead)
            hipLaunchKernelGGL(af_value_head, dim3((batch + VPB - 1) / VPB), dim3(256), 0, st, o[2], n->vc_w, n->vc_b, n->v1_w, n->v1_b,
                               n->v2_w, n->v2_b, value, batch, S, WP, PP);
            if (vs != st_main) NET_HIP_OK(hipEventRecord(n->ev_value, vs));
        }
    }
    if (fhead) {
        // (the policy head ran inside f16s_policy_branch)
    } else if (HW <= 128 && g_phead) {
        const int K = 16 * HW, KS = K + ((2 - K % 32) + 32) % 32;
        const size_t lds = (size_t)16 * KS * 4;
        hipLaunchKernelGGL(af_policy_head_mfma, dim3((batch + 15) / 16), dim3(256), lds, st, o[4], n->pc_w, n->pc_b, n->pf_w,
                           n->pf_b, policy, batch, S, WP, PP);
    } else if (HW <= 128) {
        const size_t lds = ((size_t)16 * HW * 8 + 512 + 32) * 4;
        hipLaunchKernelGGL((af_policy_head<8>), dim3((batch + 7) / 8), dim3(256), lds, st, o[4], n->pc_w, n->pc_b, n->pf_w,
                           n->pf_b, policy, batch, S, WP, PP);
    } else {
        const size_t lds = ((size_t)16 * HW * 4 + 512 + 16) * 4;
        hipLaunchKernelGGL((af_policy_head<4>), dim3((batch + 3) / 4), dim3(256), lds, st, o[4], n->pc_w, n->pc_b, n->pf_w,
                           n->pf_b, policy, batch, S, WP, PP);
    }
    if (vs != st) NET_HIP_OK(hipStreamWaitEvent(st, n->ev_value, 0));
    NET_HIP_OK(hipGetLastError());
    return AF_NET_OK;
}

extern "C" {

int af_net_forward(af_net* n, void* stream, const float* planes, int32_t batch, float* policy, float* value) {
    if (!n || !planes || !policy || !value || batch < 1 || batch > n->max_batch) return AF_NET_ERR_ARG;
    if (!n->ready) return AF_NET_ERR_STATE;
    // (r1-r5 A/B records, removed from the source in r6: concurrent sub-batches on side streams and sequential sub-batches that keep a
    //  layer's output inside the Infinity Cache — 1.415 / 1.621 ms per 4096 positions as 2 / 4 sub-batches against 1.330: profiles/r5_41)
    return forward_range(n, (hipStream_t)stream, planes, 0, batch, policy, value);
}

int af_net_tune(int32_t key, int32_t value) {
    if (key == 0) { if (value != 1 && value != 5) return AF_NET_ERR_ARG; g_wino = value; return AF_NET_OK; }   // 0: conv path (5 split fp16, 1 fp32 Winograd)
    if (key == 5) { g_phead = value; return AF_NET_OK; }                            // 5: MFMA policy head of the fp32 path (1/0)
    if (key == 4) { g_branch = value; return AF_NET_OK; }                           // 4: value branch on a side stream (fp32 path: 1/0; 2 forces it on path 5)
    if (key == 9) { g_fhead = value ? 1 : 0; return AF_NET_OK; }                    // 9: heads on the split-operand path (1/0)
    if (key == 7) { g_f16s_abl = value; return AF_NET_OK; }                         // 7: A/B and profiling bits of af_conv_f16s.hip
    return AF_NET_ERR_ARG;
}

int af_net_debug_activation(af_net* n, int32_t which, int32_t batch, float* host_out) {
    if (!n || !host_out || !n->f16s) return AF_NET_ERR_ARG;
    const int c = f16s_read_activation(n->f16s, which, batch, host_out);
    return c < 0 ? AF_NET_ERR_ARG : c;
}

int af_net_small_forward_error(af_net* n) {
    if (!n || !n->f16s) return AF_NET_ERR_ARG;
    return f16s_small_forward_error(n->f16s);
}

int64_t af_net_flops_per_position(const af_net* n) {
    const int64_t HW = n->HW;
    int64_t mac = 75 * 32 * HW;
    for (const Block& b : kBlocks) mac += ((int64_t)9 * b.cin * b.cout + (int64_t)9 * b.cout * b.cout + (int64_t)b.cin * b.cout) * HW;
    mac += 32 * 4 * HW + 4 * HW * 64 + 64 + 32 * 16 * HW + 16 * HW * HW;
    return 2 * mac;
}

}  // extern "C"
