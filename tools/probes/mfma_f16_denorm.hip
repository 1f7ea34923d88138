// probe: does v_mfma_f32_32x32x16_f16 keep fp16 denormal inputs, and does the float->half conversion produce them?
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
__global__ void k(float* out, float small) {
    h8 a, b;
    const _Float16 d = (_Float16)small;             // 1e-6 -> fp16 denormal
    for (int e = 0; e < 8; ++e) { a[e] = (e == 0 && threadIdx.x < 32) ? d : (_Float16)0.0f; b[e] = (e == 0 && threadIdx.x < 32) ? (_Float16)1024.0f : (_Float16)0.0f; }
    f16v c;
    for (int i = 0; i < 16; ++i) c[i] = 0.0f;
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    out[threadIdx.x] = c[0];
    if (threadIdx.x == 0) out[64] = (float)d;
}
int main() {
    float* o; hipMalloc(&o, 65 * 4);
    k<<<1, 64>>>(o, 1.0e-6f);
    float h[65]; hipMemcpy(h, o, 65 * 4, hipMemcpyDeviceToHost);
    printf("cvt(1e-6)->half->float = %g (denormal kept by cvt: %s)\n", h[64], h[64] != 0.0f ? "yes" : "NO");
    printf("mfma(denormal * 1024) = %g expected %g (denormal kept by MFMA: %s)\n", h[0], h[64] * 1024.0f, h[0] != 0.0f ? "yes" : "NO");
    return 0;
}
