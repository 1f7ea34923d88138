// Can one workgroup hand a tensor to ANOTHER workgroup of the same XCD through L2 without the bytes going to HBM?
// (the open question behind DESIGN's "what is left" item 5: a forward whose wide blocks pass their intermediate tensor on chip needs a
//  multi-workgroup exchange, because the second convolution's weights live in other workgroups' registers.)
// 128 producer / consumer pairs, workgroup b produces for workgroup b + 8 (blockIdx % 8 = XCD, so a pair shares an L2); per hand-over the
// producer writes 62 KB (one 128-channel 11x11 split-fp16 activation tensor), publishes a sequence number (release, agent scope), the
// consumer waits for it (acquire), reads the 62 KB and acknowledges, so that the producer may reuse the slot.  Two modes:
//   ring   : every pair reuses TWO 62-KB slots for ever (128 pairs x 124 KB = 16 MB, within the 8 x 4 MB of L2)
//   stream : every hand-over goes to fresh addresses of a 32-GB buffer (what the layer-by-layer forward does today)
//   ringL2 : the ring with a hand-made "XCD scope": sc0 stores waited for with vmcnt(0), relaxed flags, sc0 loads — no agent-scope release /
//            acquire (on a multi-XCD device those write back / invalidate L2); stale_threads tells whether the consumer saw the data
// Every spin is bounded: a pair that waits too long sets an error flag and everybody leaves (no hang).
// build: hipcc -O3 --offload-arch=gfx950 -o tools/probes/_bin/l2_handover tools/probes/l2_handover.hip
// run:   tools/probes/_bin/l2_handover            (prints us per hand-over and GB/s per mode)
//        rocprofv3 --kernel-trace --pmc FETCH_SIZE WRITE_SIZE -d out -o p -- tools/probes/_bin/l2_handover   (bytes that left L2)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
typedef uint32_t u4 __attribute__((ext_vector_type(4)));
#define OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

constexpr int kPairs = 128, kThreads = 256;
constexpr size_t kSlot = 62 * 1024;                  // bytes per hand-over
constexpr int kVec = kSlot / 16 / kThreads;          // 16-B vectors per thread (15.5 -> 15: 61,440 B moved per hand-over)
constexpr int kIters = 4000;
constexpr long long kSpinLimit = 2000000ll;          // a few seconds of polling before a pair gives up

__device__ __forceinline__ bool wait_ge(const unsigned* flag, unsigned want, unsigned* err) {
    long long spins = 0;
    while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < want) {
        if (++spins > kSpinLimit || __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
            __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return false;
        }
        __builtin_amdgcn_s_sleep(2);
    }
    return true;
}

template <bool RING, bool XCD = false>
__global__ __launch_bounds__(kThreads) void handover(u4* __restrict__ buf, unsigned* __restrict__ flags, unsigned* err, uint32_t* sink, size_t stride16, unsigned* xcc, int* dbg) {
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;          // 32 workgroups per XCD: 16 pairs
    if (threadIdx.x == 0) xcc[blockIdx.x] = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 15u;      // HW_REG_XCC_ID: where this workgroup really runs
    const bool producer = (idx & 1) == 0;
    const int pair = (idx >> 1) * 8 + xcd;                          // 0 .. 127
    unsigned* ready = flags + pair * 64;                            // (separate cache lines)
    unsigned* acked = flags + pair * 64 + 32;
    uint32_t acc = 0;
    for (int it = 0; it < kIters; ++it) {
        const size_t slot = RING ? (size_t)(it & 1) : (size_t)it;
        u4* p = buf + ((size_t)pair * (RING ? 2 : kIters) + slot) * stride16;
        if (producer) {
            if (RING && it >= 2) {                                  // the slot must have been read before it is written again
                if (threadIdx.x == 0 && !wait_ge(acked, (unsigned)(it - 1), err)) acc = 0xdead;
                __syncthreads();
                if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
            }
            if constexpr (XCD) {
                // hand-made "XCD scope" (the memory model has none): stores that are complete in the XCD's L2 (vmcnt = 0: acknowledged by
                // L2; the L1 is write-through) and NO agent-scope release, which on a multi-XCD device writes the dirty L2 lines back to memory
#pragma unroll
                for (int v = 0; v < kVec; ++v) {
                    const u4 val = u4{(uint32_t)it, (uint32_t)v, threadIdx.x, (uint32_t)pair};
                    // (s_nop: a VALU write to the data registers of a > 64-bit VMEM store needs a wait state the compiler only inserts for its own stores)
                    asm volatile("global_store_dwordx4 %0, %1, off sc0\n\ts_nop 1" : : "v"(&p[v * kThreads + threadIdx.x]), "v"(val) : "memory");
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (threadIdx.x == 0) __hip_atomic_store(ready, (unsigned)(it + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
#pragma unroll
            for (int v = 0; v < kVec; ++v) p[v * kThreads + threadIdx.x] = u4{(uint32_t)it, (uint32_t)v, threadIdx.x, (uint32_t)pair};
            __threadfence();                                        // every thread's stores visible at agent scope ...
            __syncthreads();
            if (threadIdx.x == 0) __hip_atomic_store(ready, (unsigned)(it + 1), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);   // ... before the flag
            }
        } else {
            if (threadIdx.x == 0 && !wait_ge(ready, (unsigned)(it + 1), err)) acc = 0xdead;
            __syncthreads();
            if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
            if constexpr (XCD) {
                u4 q[kVec];                                         // loads at agent scope (sc1: the L1 is bypassed — sc0 = workgroup scope may hit it — and the shared L2 answers)
                // (five loads and their wait in ONE asm statement: the compiler does not know that an asm load's destination is written
                //  later, and with the wait in a statement of its own it reused the registers in between)
#pragma unroll
                for (int v = 0; v < kVec; v += 5) {
                    const u4* a = &p[v * kThreads + threadIdx.x];
                    asm volatile("global_load_dwordx4 %0, %5, off sc1\n\tglobal_load_dwordx4 %1, %6, off sc1\n\tglobal_load_dwordx4 %2, %7, off sc1\n\t"
                                 "global_load_dwordx4 %3, %8, off sc1\n\tglobal_load_dwordx4 %4, %9, off sc1\n\ts_waitcnt vmcnt(0)"
                                 : "=&v"(q[v]), "=&v"(q[v + 1]), "=&v"(q[v + 2]), "=&v"(q[v + 3]), "=&v"(q[v + 4])
                                 : "v"(a), "v"(a + kThreads), "v"(a + 2 * kThreads), "v"(a + 3 * kThreads), "v"(a + 4 * kThreads) : "memory");
                }
#pragma unroll
                for (int v = 0; v < kVec; ++v) {
                    if ((q[v].x != (uint32_t)it || q[v].y != (uint32_t)v) && atomicCAS(&dbg[blockIdx.x * 4], 0, it + 1) == 0) {
                        dbg[blockIdx.x * 4 + 1] = (int)q[v].x; dbg[blockIdx.x * 4 + 2] = v * 1000 + (int)q[v].y; dbg[blockIdx.x * 4 + 3] = (int)threadIdx.x;
                    }
                    acc += q[v].x - (uint32_t)it + q[v].y - (uint32_t)v;
                }
                __syncthreads();
                if (threadIdx.x == 0) __hip_atomic_store(acked, (unsigned)(it + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
            __threadfence();
#pragma unroll
            for (int v = 0; v < kVec; ++v) {
                const u4 q = __builtin_nontemporal_load(&p[v * kThreads + threadIdx.x]);
                acc += q.x - (uint32_t)it + q.y - (uint32_t)v;      // 0 when the data are the producer's
            }
            __syncthreads();
            if (threadIdx.x == 0) __hip_atomic_store(acked, (unsigned)(it + 1), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    if (acc != 0) atomicAdd(sink, 1u);                              // counts threads that saw stale data (or gave up)
}

int main(int argc, char** argv) {
    const int only = argc > 1 ? atoi(argv[1]) : -1;                 // run one mode only (0 ring, 1 stream, 2 ringL2)
    const size_t stride16 = kSlot / 16;
    const size_t ring_bytes = (size_t)kPairs * 2 * kSlot, stream_bytes = (size_t)kPairs * kIters * kSlot;
    u4* buf;
    unsigned *flags, *err;
    uint32_t* sink;
    unsigned* xcc;
    unsigned hx[256];
    int* dbg;
    int hd[1024];
    OK(hipMalloc(&buf, only == 0 || only == 2 ? ring_bytes : stream_bytes));
    OK(hipMalloc(&flags, kPairs * 64 * 4));
    OK(hipMalloc(&err, 4));
    OK(hipMalloc(&sink, 4));
    OK(hipMalloc(&xcc, 256 * 4));
    OK(hipMalloc(&dbg, 1024 * 4));
    hipEvent_t e0, e1;
    OK(hipEventCreate(&e0));
    OK(hipEventCreate(&e1));
    printf("per hand-over %d B (61,440 moved), %d pairs, %d hand-overs each; ring footprint %.1f MB, stream footprint %.1f GB\n", (int)kSlot, kPairs, kIters,
           ring_bytes / 1e6, stream_bytes / 1e9);
    for (int mode = 0; mode < 3; ++mode) {
        if (only >= 0 && mode != only) continue;
        for (int rep = 0; rep < 2; ++rep) {
            OK(hipMemset(flags, 0, kPairs * 64 * 4));
            OK(hipMemset(err, 0, 4));
            OK(hipMemset(sink, 0, 4));
            OK(hipMemset(dbg, 0, 1024 * 4));
            OK(hipEventRecord(e0));
            if (mode == 0) hipLaunchKernelGGL(handover<true>, dim3(256), dim3(kThreads), 0, 0, buf, flags, err, sink, stride16, xcc, dbg);
            else if (mode == 1) hipLaunchKernelGGL(handover<false>, dim3(256), dim3(kThreads), 0, 0, buf, flags, err, sink, stride16, xcc, dbg);
            else hipLaunchKernelGGL((handover<true, true>), dim3(256), dim3(kThreads), 0, 0, buf, flags, err, sink, stride16, xcc, dbg);
            OK(hipEventRecord(e1));
            OK(hipEventSynchronize(e1));
            float ms = 0;
            OK(hipEventElapsedTime(&ms, e0, e1));
            unsigned herr = 0, hsink = 0;
            OK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
            OK(hipMemcpy(&hsink, sink, 4, hipMemcpyDeviceToHost));
            OK(hipMemcpy(hx, xcc, 256 * 4, hipMemcpyDeviceToHost));
            int split = 0;
            for (int b = 0; b < 256; ++b) if (((b >> 3) & 1) == 0 && hx[b] != hx[b + 8]) ++split;
            OK(hipMemcpy(hd, dbg, 1024 * 4, hipMemcpyDeviceToHost));
            if (mode == 2 && rep == 0)
                for (int b = 0, shown = 0; b < 256 && shown < 12; ++b)
                    if (hd[b * 4]) { printf("  consumer wg %3d (xcc %u): first stale at hand-over %d: saw it=%d (v*1000+y=%d) thread %d\n", b, hx[b], hd[b * 4] - 1, hd[b * 4 + 1], hd[b * 4 + 2], hd[b * 4 + 3]); ++shown; }
            const double moved = (double)kPairs * kIters * (kVec * kThreads * 16.0);
            printf("%-6s rep %d: %.3f ms  = %.2f us per hand-over per pair, %.1f GB/s written + the same read; gave_up=%u stale_threads=%u pairs_on_two_xcds=%d\n",
                   mode == 0 ? "ring" : (mode == 1 ? "stream" : "ringL2"), rep, ms, ms * 1e3 / kIters, moved / (ms * 1e-3) / 1e9, herr, hsink, split);
        }
    }
    return 0;
}
