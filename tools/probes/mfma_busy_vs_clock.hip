// How does the clock the power cap allows depend on MFMA-busy?  A bare chain of v_mfma_f32_32x32x16_f16 (random operands, one
// 512-register wave per SIMD, whole chip) with idle gaps (s_sleep) after every 24 MFMAs: busy b -> sustained clock f(b), and
// b x f = the MFMA work done.  If b x f did not depend on b, filling idle MFMA slots (hiding an epilogue) could not pay at all; the
// slope of b x f against b is what such a change can return at best.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/busyclk tools/probes/mfma_busy_vs_clock.hip && /tmp/busyclk
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int SLEEP>      // s_sleep units (64 cycles each) per 24 MFMAs (768 cycles)
__global__ __launch_bounds__(256, 1) void burn(const h8* __restrict__ ops, float* sink, int iters, unsigned long long* clk) {
    h8 a[8], b[8];
    for (int i = 0; i < 8; ++i) { a[i] = ops[(i * 256 + threadIdx.x)]; b[i] = ops[((8 + i) * 256 + threadIdx.x)]; }
    f32x16 acc[4];
    for (int q = 0; q < 4; ++q) for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
    const unsigned long long c0 = __builtin_readcyclecounter(), w0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], b[i], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], b[(i + 3) & 7], acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(i + 1) & 7], b[i], acc[2], 0, 0, 0);
        }
        if (SLEEP > 0) __builtin_amdgcn_s_sleep(SLEEP);
    }
    const unsigned long long c1 = __builtin_readcyclecounter(), w1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0 && blockIdx.x == 17) { clk[0] = c1 - c0; clk[1] = w1 - w0; }
    float s = 0.f;
    for (int q = 0; q < 4; ++q) for (int r = 0; r < 16; ++r) s += acc[q][r];
    sink[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int SLEEP>
static void run(h8* d, float* sink, unsigned long long* clk) {
    const int iters = 20000, launches = 40;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 5; ++w) hipLaunchKernelGGL(burn<SLEEP>, dim3(256), dim3(256), 0, 0, d, sink, iters, clk);
    hipEventRecord(e0);
    for (int l = 0; l < launches; ++l) hipLaunchKernelGGL(burn<SLEEP>, dim3(256), dim3(256), 0, 0, d, sink, iters, clk);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long hc[2] = {1, 1};
    hipMemcpy(hc, clk, 16, hipMemcpyDeviceToHost);
    const double ghz = (double)hc[0] / ((double)hc[1] * 10.0), busy = (double)iters * 24.0 * 32.0 / (double)hc[0];
    const double tf = (double)launches * 1024.0 * iters * 24.0 * 32768.0 / ms / 1e9;
    printf("s_sleep %2d per 24 MFMAs: MFMA-busy %.2f  clock %.2f GHz  busy x clock %.2f  %7.1f TFLOP/s (%.3f of 2500)\n", SLEEP, busy, ghz,
           busy * ghz, tf, tf / 2500.0);
}

int main() {
    const int nops = 4096;
    h8* h = (h8*)malloc(nops * sizeof(h8));
    srand(7);
    for (int i = 0; i < nops; ++i) for (int e = 0; e < 8; ++e) h[i][e] = (_Float16)(((rand() & 0xffff) / 32768.0f - 1.0f) * 2.0f);
    h8* d; float* sink; unsigned long long* clk;
    hipMalloc(&d, nops * sizeof(h8)); hipMalloc(&sink, 256 * 256 * 4); hipMalloc(&clk, 16);
    hipMemcpy(d, h, nops * sizeof(h8), hipMemcpyHostToDevice);
    for (int round = 0; round < 2; ++round) {
        run<0>(d, sink, clk); run<1>(d, sink, clk); run<2>(d, sink, clk); run<3>(d, sink, clk); run<5>(d, sink, clk);
        run<8>(d, sink, clk); run<12>(d, sink, clk);
    }
    return 0;
}
