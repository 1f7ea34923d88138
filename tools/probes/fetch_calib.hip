// Calibration of rocprofv3's FETCH_SIZE for the tick kernel's access pattern (MI355X_MICROARCH.md, HBM section: "other access widths
// are uncalibrated: calibrate on a known byte count in your own access pattern").  Three kernels read a known number of bytes from a
// 4 GiB buffer (far beyond L2 + Infinity Cache), each wave walking pseudo-random 512-byte rows:
//   rows_dword   : a row = 2 x global_load_dword per lane (lane l reads words l and l + 64)    <- af_tick_kernel's N / W / P / C rows
//   rows_dwordx4 : a row = 1 x global_load_dwordx4 per lane of the first 32 lanes              (the same bytes, wide loads)
//   stream_x4    : contiguous 16 B / lane streaming read                                       <- the guide's calibrated case
// build: hipcc -O3 --offload-arch=gfx950 -o tools/probes/_bin/fetch_calib tools/probes/fetch_calib.hip
// run:   rocprofv3 --kernel-trace --pmc FETCH_SIZE -d out -o p -- tools/probes/_bin/fetch_calib   (prints the known bytes per kernel)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#define OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
constexpr size_t kBytes = 4ull << 30;
constexpr int kRows = 64;            // rows per wave
constexpr int kWaves = 1 << 18;      // 262,144 waves x 64 rows x 512 B = 8.6 GB per launch

__global__ __launch_bounds__(64) void rows_dword(const uint32_t* __restrict__ p, uint32_t* out) {
    const uint64_t nrows = kBytes / 512;
    uint64_t r = (uint64_t)blockIdx.x * 0x9E3779B97F4A7C15ull;
    uint32_t acc = 0;
    for (int i = 0; i < kRows; ++i) {
        r = r * 6364136223846793005ull + 1442695040888963407ull;
        const uint32_t* row = p + ((r >> 11) % nrows) * 128;
        acc += row[threadIdx.x] + row[threadIdx.x + 64];
    }
    if (acc == 0x12345678u) out[0] = acc;
}
__global__ __launch_bounds__(64) void rows_dwordx4(const uint4* __restrict__ p, uint32_t* out) {
    const uint64_t nrows = kBytes / 512;
    uint64_t r = (uint64_t)blockIdx.x * 0x9E3779B97F4A7C15ull;
    uint32_t acc = 0;
    for (int i = 0; i < kRows; ++i) {
        r = r * 6364136223846793005ull + 1442695040888963407ull;
        const uint4* row = p + ((r >> 11) % nrows) * 32;
        if (threadIdx.x < 32) { const uint4 v = row[threadIdx.x]; acc += v.x + v.y + v.z + v.w; }
    }
    if (acc == 0x12345678u) out[0] = acc;
}
__global__ __launch_bounds__(256) void stream_x4(const uint4* __restrict__ p, uint32_t* out, size_t n16) {
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) { const uint4 v = p[i]; acc += v.x + v.y + v.z + v.w; }
    if (acc == 0x12345678u) out[0] = acc;
}
int main() {
    void* buf; uint32_t* out;
    OK(hipMalloc(&buf, kBytes)); OK(hipMalloc((void**)&out, 4));
    OK(hipMemset(buf, 1, kBytes));
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(rows_dword, dim3(kWaves), dim3(64), 0, 0, (const uint32_t*)buf, out);
        hipLaunchKernelGGL(rows_dwordx4, dim3(kWaves), dim3(64), 0, 0, (const uint4*)buf, out);
        hipLaunchKernelGGL(stream_x4, dim3(4096), dim3(256), 0, 0, (const uint4*)buf, out, kBytes / 16);
    }
    OK(hipDeviceSynchronize());
    printf("known bytes per launch: rows_dword %zu rows_dwordx4 %zu stream_x4 %zu\n", (size_t)kWaves * kRows * 512, (size_t)kWaves * kRows * 512, kBytes);
    return 0;
}
