// Where does the forward's power go?  A bare chain of v_mfma_f32_32x32x16_f16 on random operands sustains ~1.68 PFLOP/s under the
// 1,400-W cap (tools/probes/mfma_shape_power.hip); af_conv_f16s issues 1.11.  This probe adds the forward's OTHER activities to the
// bare chain one at a time, at the forward's own rates per MFMA, and reports the MFMA rate the cap then allows:
//   L : B fragments from LDS, 2 x ds_read_b128 per 3 MFMAs (one cout tile per wave: CT across waves) / per 6 MFMAs
//   D : the slab stream, LDS-DMA (global_load_lds_dwordx4) of fresh HBM lines at ~2.4 TB/s chip-wide
//   V : the epilogue's VALU work at about twice its rate: 4 VALU instructions per 3 MFMAs, one of them v_exp_f32 (the 128 -> 128 layer runs 0.63 per MFMA)
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_budget tools/probes/mfma_power_budget.hip && /tmp/mfma_budget
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// MODE bits: 1 = LDS B fragments per 3 MFMAs, 2 = LDS B fragments per 6 MFMAs, 4 = LDS-DMA stream, 8 = VALU work
template <int MODE>
__global__ __launch_bounds__(256, 1) void burn(const h8* __restrict__ ops, const char* __restrict__ stream, size_t stream_bytes,
                                               float* sink, int iters, unsigned long long* clk) {
    extern __shared__ char smem[];
    h8 a[8], b[8];
    for (int i = 0; i < 8; ++i) { a[i] = ops[(i * 256 + threadIdx.x)]; b[i] = ops[((8 + i) * 256 + threadIdx.x)]; }
    // LDS image of random fragments (64 KB), read with the conflict-free lane*16 pattern of a ring slot
    for (int u = threadIdx.x; u < 4096; u += 256) *reinterpret_cast<h8*>(smem + u * 16) = ops[u & 4095];
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    f32x16 acc[4];
    for (int q = 0; q < 4; ++q) for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
    float x[6];
    for (int k = 0; k < 6; ++k) x[k] = threadIdx.x * 0.01f + k;
    // this workgroup's share of the stream: fresh 1-KB pieces, wrapping inside a window far larger than every cache
    const size_t per_wg = stream_bytes / gridDim.x;
    const char* src = stream + (size_t)blockIdx.x * per_wg;
    size_t off = (size_t)wv * 1024;
    const uint32_t lds_dma = (uint32_t)(uintptr_t)smem + 65536u + (uint32_t)wv * 1024u;
    const unsigned long long c0 = __builtin_readcyclecounter(), w0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            h8 bh = b[i], bl = b[(i + 3) & 7];
            if (MODE & 3) {
                const uint32_t o = (uint32_t)(((it * 8 + i) & 31) * 2048 + lane * 16);
                if ((MODE & 1) || (i & 1) == 0) {
                    bh = *reinterpret_cast<const h8*>(smem + o);
                    bl = *reinterpret_cast<const h8*>(smem + o + 1024);
                }
            }
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], bh, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], bl, acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(i + 1) & 7], bh, acc[2], 0, 0, 0);
            if (MODE & 8) {
                x[i % 3] = x[i % 3] * 0.999f + 0.001f;
                x[3 + i % 3] = x[3 + i % 3] * 0.999f + 0.001f;
                x[i % 6] = __builtin_amdgcn_exp2f(x[i % 6] * 0.01f);
            }
            if ((MODE & 4) && (i & 3) == 0) {
                // 2 pieces of 1 KB per wave per 24 MFMAs: 4 waves x 2 KB per 24 x 32 cycles ~ 10.7 B/clk/CU ~ 2.4 TB/s at 256 CUs x ~0.9 GHz-equivalent issue
                unsigned keep;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep) : "v"(src + off + lane * 16), "s"(__builtin_amdgcn_readfirstlane((int)lds_dma)) : "memory");
                off += 4096;
                if (off + 4096 > per_wg) off = (size_t)wv * 1024;
            }
        }
        if (MODE & 4) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long c1 = __builtin_readcyclecounter(), w1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0 && blockIdx.x == 17) { clk[0] = c1 - c0; clk[1] = w1 - w0; }      // shader cycles | 100-MHz ticks of one wave's loop
    float s = 0.f;
    for (int q = 0; q < 4; ++q) for (int r = 0; r < 16; ++r) s += acc[q][r];
    for (int k = 0; k < 6; ++k) s += x[k];
    sink[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
static void run(const char* name, h8* d, char* stream, size_t sb, float* sink, int iters, int launches, unsigned long long* clk) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(burn<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 5; ++w) hipLaunchKernelGGL(burn<MODE>, dim3(256), dim3(256), 72 * 1024, 0, d, stream, sb, sink, iters, clk);
    hipEventRecord(e0);
    for (int l = 0; l < launches; ++l) hipLaunchKernelGGL(burn<MODE>, dim3(256), dim3(256), 72 * 1024, 0, d, stream, sb, sink, iters, clk);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)launches * 1024.0 * iters * 24.0 * 32768.0;
    const double gbs = (MODE & 4) ? (double)launches * 1024.0 * iters * 2.0 * 1024.0 / ms / 1e6 : 0.0;
    unsigned long long hc[2] = {1, 1};
    hipMemcpy(hc, clk, 16, hipMemcpyDeviceToHost);
    const double ghz = (double)hc[0] / ((double)hc[1] * 10.0), busy = (double)iters * 24.0 * 32.0 / (double)hc[0];
    printf("%-44s %7.1f ms  %7.1f TFLOP/s MFMA (%.3f of 2500)  clock %.2f GHz  MFMA-busy %.2f  stream %.0f GB/s\n", name, ms,
           flop / ms / 1e9, flop / ms / 1e9 / 2500.0, ghz, busy, gbs);
    hipEventDestroy(e0); hipEventDestroy(e1);
}

int main() {
    const int nops = 4096;
    h8* h = (h8*)malloc(nops * sizeof(h8));
    srand(7);
    for (int i = 0; i < nops; ++i) for (int e = 0; e < 8; ++e) h[i][e] = (_Float16)(((rand() & 0xffff) / 32768.0f - 1.0f) * 2.0f);
    h8* d; float* sink; char* stream; unsigned long long* clk;
    const size_t sb = (size_t)3 << 30;                          // 3 GB of random halves: nothing of it stays in a cache
    hipMalloc(&d, nops * sizeof(h8)); hipMalloc(&sink, 256 * 256 * 4); hipMalloc(&stream, sb); hipMalloc(&clk, 16);
    hipMemcpy(d, h, nops * sizeof(h8), hipMemcpyHostToDevice);
    for (size_t o = 0; o < sb; o += nops * sizeof(h8)) hipMemcpyAsync(stream + o, d, nops * sizeof(h8), hipMemcpyDeviceToDevice, 0);
    hipDeviceSynchronize();
    const int iters = 20000, launches = 50;
    for (int round = 0; round < 3; ++round) {
        printf("round %d\n", round);
        run<0>("bare MFMA chain", d, stream, sb, sink, iters, launches, clk);
        run<1>("+ LDS B fragments, 2 per 3 MFMAs", d, stream, sb, sink, iters, launches, clk);
        run<2>("+ LDS B fragments, 2 per 6 MFMAs", d, stream, sb, sink, iters, launches, clk);
        run<4>("+ LDS-DMA stream from HBM", d, stream, sb, sink, iters, launches, clk);
        run<8>("+ VALU (4 per 3 MFMAs, 1 of them v_exp)", d, stream, sb, sink, iters, launches, clk);
        run<1 | 4 | 8>("+ all three (LDS 2 per 3)", d, stream, sb, sink, iters, launches, clk);
    }
    return 0;
}
