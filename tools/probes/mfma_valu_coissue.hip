// Does a wave's own VALU work run in the shadow of its MFMAs on gfx950?  One wave per SIMD issues a chain of
// v_mfma_f32_32x32x16_f16 (two accumulators, alternating) with K independent VALU instructions behind each MFMA, in program order
// (inline asm).  Prints shader cycles per MFMA for K = 0..12 (v_fma_f32) and for v_exp_f32.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/coissue tools/probes/mfma_valu_coissue.hip && /tmp/coissue
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int K, bool EXP>
__global__ __launch_bounds__(256, 1) void probe(unsigned long long* out, float* sink) {
    h8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(threadIdx.x * 0.001f + e); b[e] = (_Float16)(0.5f + e); }
    f32x16 acc0, acc1;
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    float x[12];
    for (int k = 0; k < 12; ++k) x[k] = threadIdx.x * 0.01f + k;
    const float y = 0.999f, z = 0.001f;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < 512; ++it) {
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc0) : "v"(a), "v"(b));
#pragma unroll
        for (int k = 0; k < K; ++k) {
            if (EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(x[k % 12]));
            else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[k % 12]) : "v"(y), "v"(z));
        }
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc1) : "v"(a), "v"(b));
#pragma unroll
        for (int k = 0; k < K; ++k) {
            if (EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(x[k % 12]));
            else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[k % 12]) : "v"(y), "v"(z));
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r];
    for (int k = 0; k < 12; ++k) s += x[k];
    sink[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}

template <int K, bool EXP>
void run(unsigned long long* d, float* sink) {
    hipLaunchKernelGGL((probe<K, EXP>), dim3(256), dim3(256), 0, 0, d, sink);
    hipLaunchKernelGGL((probe<K, EXP>), dim3(256), dim3(256), 0, 0, d, sink);
    unsigned long long h = 0;
    hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
    printf("%s K=%2d: %.1f cycles per MFMA\n", EXP ? "v_exp_f32" : "v_fma_f32", K, (double)h / 1024.0);
}

int main() {
    unsigned long long* d; float* sink;
    hipMalloc(&d, 8); hipMalloc(&sink, 256 * 256 * 4);
    run<0, false>(d, sink); run<2, false>(d, sink); run<4, false>(d, sink); run<6, false>(d, sink); run<7, false>(d, sink);
    run<8, false>(d, sink); run<10, false>(d, sink); run<12, false>(d, sink);
    run<1, true>(d, sink); run<2, true>(d, sink); run<3, true>(d, sink);
    return 0;
}
