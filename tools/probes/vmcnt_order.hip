// vmcnt_order.hip — does gfx950 retire a wave's vector-memory operations in issue order across loads AND stores?
// (gfx9 has ONE counter, vmcnt, for loads, LDS-DMA loads and stores; a counted `s_waitcnt vmcnt(N)` only names "all but the N
// youngest" if retirement is in order.)  af_conv_f16s wants to leave its epilogue's stores in flight across the next
// LDS-DMA wait; that is only sound if a YOUNGER store can never retire before an OLDER LDS-DMA load.
//
// Test per wave and iteration: LDS destination cleared; one LDS-DMA load (1 KB) from a cold, pseudo-random place of a 4 GB
// buffer (HBM miss, slow); then NST stores to a small L2-hot buffer (fast); `s_waitcnt vmcnt(NST)`; read the LDS
// destination at once.  In-order retirement: the load has landed, every lane sees the pattern.  Out-of-order retirement:
// the stores can bring the counter down to NST while the load is still in flight -> stale zeros are seen (counted).
// Also times vmcnt(NST) against vmcnt(0) and the reverse order (stores to cold lines, then a hot load, vmcnt(1)).
//
//   hipcc -O2 --offload-arch=gfx950 -o tools/probes/_bin/vmcnt_order tools/probes/vmcnt_order.hip && tools/probes/_bin/vmcnt_order
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

#define OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ void glds16(const void* gsrc, uint32_t lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

template <int NST>
__global__ __launch_bounds__(256) void probe_load_then_stores(const uint4* __restrict__ cold, size_t cold_units, uint4* __restrict__ hot,
                                                              int iters, unsigned long long* out /*[blocks*4][4]*/) {
    __shared__ __attribute__((aligned(16))) uint4 dst[4][64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint32_t lds = (uint32_t)(uintptr_t)&dst[wv][0];
    unsigned long long stale = 0, t_cnt = 0, t_zero = 0;
    uint32_t rng = 0x9E3779B9u * (blockIdx.x * 4 + wv + 1);
    uint4* hp = hot + ((size_t)(blockIdx.x * 4 + wv) * NST) * 64 + lane;
    for (int it = 0; it < iters; ++it) {
        rng = rng * 1664525u + 1013904223u;
        const size_t unit = ((size_t)rng * 64u) % (cold_units - 64);               // a fresh 1 KB piece somewhere in the cold buffer
        dst[wv][lane] = uint4{0, 0, 0, 0};
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        const unsigned long long t0 = __builtin_readcyclecounter();
        glds16(cold + unit + lane, (uint32_t)__builtin_amdgcn_readfirstlane((int)lds));
#pragma unroll
        for (int s = 0; s < NST; ++s) {
            const u32x4 v = {(uint32_t)it, (uint32_t)s, (uint32_t)lane, 7u};
            asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 0" ::"v"(hp + s * 64), "v"(v) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NST) : "memory");
        const uint4 got = dst[wv][lane];
        const unsigned long long t1 = __builtin_readcyclecounter();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long t2 = __builtin_readcyclecounter();
        // the cold buffer holds x = unit index (low word), y = ~x
        const uint32_t want = (uint32_t)(unit + lane);
        if (got.x != want || got.y != ~want) stale++;
        t_cnt += t1 - t0;
        t_zero += t2 - t0;
    }
    // wave totals
    for (int off = 32; off >= 1; off >>= 1) stale += __shfl_xor(stale, off);
    if (lane == 0) {
        unsigned long long* o = out + (size_t)(blockIdx.x * 4 + wv) * 4;
        o[0] = stale; o[1] = t_cnt; o[2] = t_zero; o[3] = (unsigned long long)iters;
    }
}

// reverse: NST stores to cold lines (slow), then one L2-hot load, vmcnt(1) vs vmcnt(0): in order => both wait for the stores
template <int NST>
__global__ __launch_bounds__(256) void probe_stores_then_load(uint4* __restrict__ cold, size_t cold_units, const uint4* __restrict__ hot,
                                                              int iters, unsigned long long* out) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    unsigned long long t_one = 0, t_zero = 0, t_load = 0;
    uint32_t rng = 0x85EBCA6Bu * (blockIdx.x * 4 + wv + 1);
    const uint4* hp = hot + (size_t)(blockIdx.x * 4 + wv) * 64 + lane;
    uint32_t sink = 0;
    for (int it = 0; it < iters; ++it) {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        {   // the hot load alone
            const unsigned long long a = __builtin_readcyclecounter();
            u32x4 v;
            asm volatile("global_load_dwordx4 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(hp) : "memory");
            t_load += __builtin_readcyclecounter() - a;
            sink += v[0];
        }
        const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll
        for (int s = 0; s < NST; ++s) {
            rng = rng * 1664525u + 1013904223u;
            const size_t unit = ((size_t)rng * 64u) % (cold_units - 64);
            const u32x4 v = {(uint32_t)(unit + lane), ~(uint32_t)(unit + lane), 0u, 0u};          // keeps the buffer's pattern intact
            asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 0" ::"v"(cold + unit + lane), "v"(v) : "memory");
        }
        u32x4 v;
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(hp) : "memory");
        asm volatile("s_waitcnt vmcnt(1)" ::: "memory");       // in order: everything but the load -> the stores are done
        const unsigned long long t1 = __builtin_readcyclecounter();
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(v)::"memory");
        const unsigned long long t2 = __builtin_readcyclecounter();
        sink += v[0];
        t_one += t1 - t0;
        t_zero += t2 - t0;
    }
    if (lane == 0) {
        unsigned long long* o = out + (size_t)(blockIdx.x * 4 + wv) * 4;
        o[0] = t_load; o[1] = t_one; o[2] = t_zero; o[3] = (unsigned long long)iters + (sink == 0xFFFFFFFFu);
    }
}

__global__ void fill_pattern(uint4* p, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        p[i] = uint4{(uint32_t)i, ~(uint32_t)i, 0u, 0u};
}

template <int NST>
int run(const uint4* cold, size_t units, uint4* hot, unsigned long long* dout, int blocks, int iters) {
    OK(hipMemset(dout, 0, (size_t)blocks * 4 * 4 * 8));
    hipLaunchKernelGGL(probe_load_then_stores<NST>, dim3(blocks), dim3(256), 0, 0, cold, units, hot, iters, dout);
    OK(hipDeviceSynchronize());
    unsigned long long* h = (unsigned long long*)malloc((size_t)blocks * 4 * 4 * 8);
    OK(hipMemcpy(h, dout, (size_t)blocks * 4 * 4 * 8, hipMemcpyDeviceToHost));
    unsigned long long stale = 0, tc = 0, tz = 0, n = 0;
    for (int w = 0; w < blocks * 4; ++w) { stale += h[4 * w]; tc += h[4 * w + 1]; tz += h[4 * w + 2]; n += h[4 * w + 3]; }
    printf("LDS-DMA load (cold) then %2d stores (hot), vmcnt(%2d): %llu wave-iterations, stale lanes %llu, cycles to vmcnt(N) %.0f, to vmcnt(0) %.0f\n",
           NST, NST, n, stale, (double)tc / n, (double)tz / n);
    free(h);
    return stale ? 2 : 0;
}

template <int NST>
int run_rev(uint4* cold, size_t units, const uint4* hot, unsigned long long* dout, int blocks, int iters) {
    OK(hipMemset(dout, 0, (size_t)blocks * 4 * 4 * 8));
    hipLaunchKernelGGL(probe_stores_then_load<NST>, dim3(blocks), dim3(256), 0, 0, cold, units, hot, iters, dout);
    OK(hipDeviceSynchronize());
    unsigned long long* h = (unsigned long long*)malloc((size_t)blocks * 4 * 4 * 8);
    OK(hipMemcpy(h, dout, (size_t)blocks * 4 * 4 * 8, hipMemcpyDeviceToHost));
    unsigned long long tl = 0, t1 = 0, t0 = 0, n = 0;
    for (int w = 0; w < blocks * 4; ++w) { tl += h[4 * w]; t1 += h[4 * w + 1]; t0 += h[4 * w + 2]; n += h[4 * w + 3]; }
    printf("%2d stores (cold) then load (hot): hot load alone %.0f cycles; to vmcnt(1) %.0f; to vmcnt(0) %.0f\n", NST, (double)tl / n,
           (double)t1 / n, (double)t0 / n);
    free(h);
    return 0;
}

int main() {
    const size_t units = (size_t)4 << 26;                  // 4 GiB of uint4
    uint4 *cold = nullptr, *hot = nullptr;
    unsigned long long* dout = nullptr;
    const int blocks = 1024, iters = 2000;
    OK(hipMalloc(&cold, units * 16));
    OK(hipMalloc(&hot, (size_t)blocks * 4 * 16 * 64 * 16));
    OK(hipMalloc(&dout, (size_t)blocks * 4 * 4 * 8));
    hipLaunchKernelGGL(fill_pattern, dim3(4096), dim3(256), 0, 0, cold, units);
    OK(hipMemset(hot, 0, (size_t)blocks * 4 * 16 * 64 * 16));
    OK(hipDeviceSynchronize());
    int bad = 0;
    bad |= run<1>(cold, units, hot, dout, blocks, iters);
    bad |= run<4>(cold, units, hot, dout, blocks, iters);
    bad |= run<8>(cold, units, hot, dout, blocks, iters);
    bad |= run<16>(cold, units, hot, dout, blocks, iters);
    run_rev<8>(cold, units, hot, dout, blocks, iters);
    printf(bad ? "RESULT: a younger store retired before an older LDS-DMA load (out of order)\n"
               : "RESULT: no stale read: vmcnt retires LDS-DMA loads and stores of a wave in issue order\n");
    return 0;
}
