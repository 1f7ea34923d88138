// Which fp16 MFMA shape does more work under the board's power cap?  The forward is power-bound, so a shape that
// needs fewer joules per FLOP would be worth a re-tiling.  Whole chip, one 512-register wave per SIMD (as the forward runs), a bare
// chain of MFMAs on random fp16 operands (8 A x 8 B fragments in rotation, 4 independent accumulators), ~2 s per variant, variants
// interleaved: v_mfma_f32_32x32x16_f16 (what af_conv_f16s issues) against v_mfma_f32_16x16x32_f16 at the same FLOP count.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_shape tools/probes/mfma_shape_power.hip && /tmp/mfma_shape
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE>
__global__ __launch_bounds__(256, 1) void burn(const h8* __restrict__ ops, float* sink, int iters) {
    h8 a[8], b[8];
    for (int i = 0; i < 8; ++i) { a[i] = ops[(i * 256 + threadIdx.x)]; b[i] = ops[((8 + i) * 256 + threadIdx.x)]; }
    float s = 0.f;
    if (SHAPE == 32) {
        f32x16 acc[4];
        for (int q = 0; q < 4; ++q) for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], b[(i + 2 * q) & 7], acc[q], 0, 0, 0);
        }
        for (int q = 0; q < 4; ++q) for (int r = 0; r < 16; ++r) s += acc[q][r];
    } else {
        f32x4 acc[8];
        for (int q = 0; q < 8; ++q) for (int r = 0; r < 4; ++r) acc[q][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int q = 0; q < 8; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], b[(i + q) & 7], acc[q], 0, 0, 0);
        }
        for (int q = 0; q < 8; ++q) for (int r = 0; r < 4; ++r) s += acc[q][r];
    }
    sink[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
    const int nops = 16 * 256;
    h8* h = (h8*)malloc(nops * sizeof(h8));
    srand(7);
    for (int i = 0; i < nops; ++i) for (int e = 0; e < 8; ++e) h[i][e] = (_Float16)(((rand() & 0xffff) / 32768.0f - 1.0f) * 2.0f);
    h8* d; float* sink;
    hipMalloc(&d, nops * sizeof(h8)); hipMalloc(&sink, 256 * 256 * 4);
    hipMemcpy(d, h, nops * sizeof(h8), hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;                                   // per launch: 32 (64) MFMAs per iteration
    for (int round = 0; round < 4; ++round) {
        for (int shape = 32; shape >= 16; shape -= 16) {
            const int launches = 60;
            for (int w = 0; w < 5; ++w) {
                if (shape == 32) hipLaunchKernelGGL(burn<32>, dim3(256), dim3(256), 0, 0, d, sink, iters);
                else hipLaunchKernelGGL(burn<16>, dim3(256), dim3(256), 0, 0, d, sink, iters);
            }
            hipEventRecord(e0);
            for (int l = 0; l < launches; ++l) {
                if (shape == 32) hipLaunchKernelGGL(burn<32>, dim3(256), dim3(256), 0, 0, d, sink, iters);
                else hipLaunchKernelGGL(burn<16>, dim3(256), dim3(256), 0, 0, d, sink, iters);
            }
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms = 0; hipEventElapsedTime(&ms, e0, e1);
            const double flop = (double)launches * 1024.0 * iters * (shape == 32 ? 32.0 * 32768.0 : 64.0 * 16384.0);
            printf("round %d  v_mfma_f32_%s_f16: %.1f ms for %d launches, %.1f TFLOP/s (%.3f of 2500)\n", round,
                   shape == 32 ? "32x32x16" : "16x16x32", ms, launches, flop / ms / 1e9, flop / ms / 1e9 / 2500.0);
        }
    }
    return 0;
}
