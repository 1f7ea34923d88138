// af_replay.hip — device-resident replay ring + get_data gather/augmentation kernel (C ABI: include/af_replay.h).
// HBM-bound byte/index work: per sample the kernel reads 121 B of board + 484 B of policy and writes 1452 B of
// planes + 484 B of policy (2.5 KB of algorithmic traffic per sample), one 256-thread workgroup per sample.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <vector>

#include "af_replay.h"

#define RP_HIP_OK(expr)                                      \
    do {                                                     \
        hipError_t err_ = (expr);                            \
        if (err_ != hipSuccess) return AF_REPLAY_ERR_HIP;    \
    } while (0)

struct af_replay {
    int S = 0, C = 0, cap = 0, device = 0;
    int64_t head = 0;            // physical slot of the oldest position
    int count = 0;
    int8_t* boards = nullptr;    // [cap][C]
    float* policies = nullptr;   // [cap][C]
    int32_t* last = nullptr;     // [cap]
    float* values = nullptr;     // [cap]
    float* weights = nullptr;    // [cap]
    int32_t* sel = nullptr;      // [3][sel_cap] device copy of (slot, quarter turns, flip)
    int sel_cap = 0;
    void* pinned = nullptr;      // staging for append / sample arguments
    size_t pinned_bytes = 0;
};

struct SampleArgs {
    const int8_t* boards;
    const float* policies;
    const int32_t* last;
    const float* values;
    const float* weights;
    const int32_t* sel;          // [3][num]: physical slot, quarter turns, flip
    float* out_boards;
    float* out_weights;
    float* out_values;
    float* out_policies;
    int S, C, num;
};

// Output cell (oi, oj) of the augmented board comes from source cell (si, sj): undo the flip, then undo k quarter
// turns.  Forward maps (utils.py:133-140): one np.rot90 turn sends (i, j) -> (S-1-j, i); the flip sends (i, j) -> (S-1-i, j).
__device__ __forceinline__ int source_cell(int oi, int oj, int k, int flip, int S) {
    int si = flip ? S - 1 - oi : oi, sj = oj;
    for (int t = 0; t < k; ++t) {
        const int ni = sj, nj = S - 1 - si;
        si = ni; sj = nj;
    }
    return si * S + sj;
}

__global__ __launch_bounds__(256) void af_replay_sample_kernel(SampleArgs A) {
    const int b = blockIdx.x;
    const int slot = A.sel[b], k = A.sel[A.num + b], flip = A.sel[2 * A.num + b];
    const int S = A.S, C = A.C;
    int la = A.last[slot];
    if (la >= 0) {                                     // forward map of the last move
        int i = la / S, j = la - i * S;
        for (int t = 0; t < k; ++t) {
            const int ni = S - 1 - j, nj = i;
            i = ni; j = nj;
        }
        if (flip) i = S - 1 - i;
        la = i * S + j;
    }
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        const int oi = c / S, oj = c - oi * S;
        const int src = source_cell(oi, oj, k, flip, S);
        const int v = A.boards[(size_t)slot * C + src];
        float* ob = A.out_boards + (size_t)b * 3 * C;
        ob[c] = v == 1 ? 1.0f : 0.0f;                  // utils.py:256-272 board_to_inputs
        ob[C + c] = v == -1 ? 1.0f : 0.0f;
        ob[2 * C + c] = c == la ? 1.0f : 0.0f;
        A.out_policies[(size_t)b * C + c] = A.policies[(size_t)slot * C + src];
    }
    if (threadIdx.x == 0) {
        A.out_weights[b] = A.weights[slot];
        A.out_values[b] = A.values[slot];
    }
}

static int ensure_pinned(af_replay* r, size_t bytes) {
    if (bytes <= r->pinned_bytes) return AF_REPLAY_OK;
    if (r->pinned) (void)hipHostFree(r->pinned);
    r->pinned = nullptr; r->pinned_bytes = 0;
    RP_HIP_OK(hipHostMalloc(&r->pinned, bytes, hipHostMallocDefault));
    r->pinned_bytes = bytes;
    return AF_REPLAY_OK;
}

extern "C" {

const char* af_replay_strerror(int code) {
    switch (code) {
        case AF_REPLAY_OK: return "ok";
        case AF_REPLAY_ERR_ARG: return "bad argument";
        case AF_REPLAY_ERR_HIP: return "HIP runtime error";
        case AF_REPLAY_ERR_FULL: return "replay ring full";
        case AF_REPLAY_ERR_RANGE: return "index outside the stored positions";
        default: return "unknown error";
    }
}

int af_replay_create(int32_t S, int32_t capacity, int32_t device, af_replay** out) {
    if (!out || S < 3 || S > 16 || capacity < 1) return AF_REPLAY_ERR_ARG;
    RP_HIP_OK(hipSetDevice(device));
    af_replay* r = new af_replay();
    r->S = S; r->C = S * S; r->cap = capacity; r->device = device;
    const size_t n = (size_t)capacity;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&r->boards), n * r->C);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&r->policies), n * r->C * 4);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&r->last), n * 4);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&r->values), n * 4);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&r->weights), n * 4);
    if (e != hipSuccess) { af_replay_destroy(r); return AF_REPLAY_ERR_HIP; }
    *out = r;
    return AF_REPLAY_OK;
}

void af_replay_destroy(af_replay* r) {
    if (!r) return;
    (void)hipSetDevice(r->device);
    if (r->boards) (void)hipFree(r->boards);
    if (r->policies) (void)hipFree(r->policies);
    if (r->last) (void)hipFree(r->last);
    if (r->values) (void)hipFree(r->values);
    if (r->weights) (void)hipFree(r->weights);
    if (r->sel) (void)hipFree(r->sel);
    if (r->pinned) (void)hipHostFree(r->pinned);
    delete r;
}

int32_t af_replay_size(const af_replay* r) { return r ? r->count : 0; }

int af_replay_drop_front(af_replay* r, int32_t n) {
    if (!r || n < 0) return AF_REPLAY_ERR_ARG;
    if (n > r->count) return AF_REPLAY_ERR_RANGE;
    r->head = (r->head + n) % r->cap;
    r->count -= n;
    return AF_REPLAY_OK;
}

int af_replay_append(af_replay* r, void* stream, int32_t n, const int8_t* boards, const float* policies,
                     const int32_t* last_cell, const float* values, const float* weights) {
    if (!r || n < 0 || (n > 0 && (!boards || !policies || !last_cell || !values || !weights))) return AF_REPLAY_ERR_ARG;
    if (n == 0) return AF_REPLAY_OK;
    if (r->count + n > r->cap) return AF_REPLAY_ERR_FULL;
    RP_HIP_OK(hipSetDevice(r->device));
    hipStream_t st = static_cast<hipStream_t>(stream);
    const size_t C = r->C;
    // the ring wraps: copy in up to two runs
    int done = 0;
    while (done < n) {
        const int64_t slot = (r->head + r->count + done) % r->cap;
        const int run = (int)((int64_t)(n - done) < r->cap - slot ? (n - done) : r->cap - slot);
        RP_HIP_OK(hipMemcpyAsync(r->boards + slot * C, boards + (size_t)done * C, (size_t)run * C, hipMemcpyHostToDevice, st));
        RP_HIP_OK(hipMemcpyAsync(r->policies + slot * C, policies + (size_t)done * C, (size_t)run * C * 4, hipMemcpyHostToDevice, st));
        RP_HIP_OK(hipMemcpyAsync(r->last + slot, last_cell + done, (size_t)run * 4, hipMemcpyHostToDevice, st));
        RP_HIP_OK(hipMemcpyAsync(r->values + slot, values + done, (size_t)run * 4, hipMemcpyHostToDevice, st));
        RP_HIP_OK(hipMemcpyAsync(r->weights + slot, weights + done, (size_t)run * 4, hipMemcpyHostToDevice, st));
        done += run;
    }
    RP_HIP_OK(hipStreamSynchronize(st));               // pageable host memory: the caller may reuse its arrays on return
    r->count += n;
    return AF_REPLAY_OK;
}

int af_replay_sample(af_replay* r, void* stream, int32_t num, const int32_t* idx, const int32_t* quarter_turns,
                     const int32_t* flip, float* boards_dev, float* weights_dev, float* values_dev, float* policies_dev) {
    if (!r || num < 0 || (num > 0 && (!idx || !quarter_turns || !flip || !boards_dev || !weights_dev || !values_dev || !policies_dev)))
        return AF_REPLAY_ERR_ARG;
    if (num == 0) return AF_REPLAY_OK;
    RP_HIP_OK(hipSetDevice(r->device));
    hipStream_t st = static_cast<hipStream_t>(stream);
    int rc = ensure_pinned(r, (size_t)3 * num * 4);
    if (rc) return rc;
    if (num > r->sel_cap) {
        if (r->sel) (void)hipFree(r->sel);
        r->sel = nullptr; r->sel_cap = 0;
        RP_HIP_OK(hipMalloc(reinterpret_cast<void**>(&r->sel), (size_t)3 * num * 4));
        r->sel_cap = num;
    }
    RP_HIP_OK(hipStreamSynchronize(st));               // a previous sample's staging copy must have left the pinned buffer
    int32_t* h = static_cast<int32_t*>(r->pinned);
    for (int i = 0; i < num; ++i) {
        if (idx[i] < 0 || idx[i] >= r->count || quarter_turns[i] < 0 || quarter_turns[i] > 3) return AF_REPLAY_ERR_RANGE;
        h[i] = (int32_t)((r->head + idx[i]) % r->cap);
        h[num + i] = quarter_turns[i];
        h[2 * num + i] = flip[i] ? 1 : 0;
    }
    RP_HIP_OK(hipMemcpyAsync(r->sel, h, (size_t)3 * num * 4, hipMemcpyHostToDevice, st));
    SampleArgs a;
    a.boards = r->boards; a.policies = r->policies; a.last = r->last; a.values = r->values; a.weights = r->weights;
    a.sel = r->sel; a.out_boards = boards_dev; a.out_weights = weights_dev; a.out_values = values_dev; a.out_policies = policies_dev;
    a.S = r->S; a.C = r->C; a.num = num;
    hipLaunchKernelGGL(af_replay_sample_kernel, dim3(num), dim3(256), 0, st, a);
    RP_HIP_OK(hipGetLastError());
    return AF_REPLAY_OK;
}

}  // extern "C"
