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
    float* wtab = nullptr;       // [max_T + 1][max_T] construct_weights rows (af_replay_set_weights)
    int max_T = 0;
    int32_t* err = nullptr;      // device flag: a packed append whose T did not match the buffer
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

// One workgroup per ply of episode `ep` of a packed hand-off buffer (layout: include/af_engine.h af_engine_pack_episodes:
// header {episodes, plies, K, C}, meta [max_eps][4] = {game, seq, T, first ply}, final values [max_eps], then per ply
// K u64 key words (mine bitboard, theirs bitboard) | C policy floats | C visit counts | last cell | action).
struct PackedArgs {
    const int32_t* buf;
    int8_t* boards;
    float* policies;
    int32_t* last;
    float* values;
    float* weights;
    const float* wtab;
    int32_t* err;
    int64_t slot0;               // ring slot of ply 0
    int cap, C, max_eps, ep, T, max_T;
};

__global__ __launch_bounds__(256) void af_replay_append_packed_kernel(PackedArgs A) {
    const int32_t* buf = A.buf;
    const int n_eps = buf[0], K = buf[2], Cc = buf[3];
    const int32_t* meta = buf + 4 + 4 * A.ep;
    const int T = meta[2], p0 = meta[3];
    if (A.ep >= n_eps || T != A.T || Cc != A.C) {                 // the host's view of the buffer is stale: append nothing
        if (blockIdx.x == 0 && threadIdx.x == 0) atomicExch(A.err, 1);
        return;
    }
    const int t = blockIdx.x, R = 2 * K + 2 * Cc + 2, KW = K / 2;
    const int32_t* rec = buf + 4 + 5 * A.max_eps + (size_t)(p0 + t) * R;
    const int64_t slot = (A.slot0 + t) % A.cap;
    const uint32_t* key = reinterpret_cast<const uint32_t*>(rec);      // u64 word w = key[2w] | key[2w+1] << 32
    for (int c = threadIdx.x; c < Cc; c += blockDim.x) {
        const int w = c >> 6, b = c & 63;
        const uint32_t mine = key[2 * w + (b >> 5)] >> (b & 31), theirs = key[2 * (KW + w) + (b >> 5)] >> (b & 31);
        A.boards[slot * Cc + c] = (int8_t)((mine & 1u) ? 1 : ((theirs & 1u) ? -1 : 0));
        A.policies[slot * Cc + c] = __int_as_float(rec[2 * K + c]);
    }
    if (threadIdx.x == 0) {
        const float fv = __int_as_float(buf[4 + 4 * A.max_eps + A.ep]);
        float v = (T & 1) ? -fv : fv;                               // player.py:74-76, then the sign alternates ply by ply (:80-81)
        if (t & 1) v = -v;
        A.last[slot] = rec[2 * K + 2 * Cc];
        A.values[slot] = v;
        A.weights[slot] = A.wtab[(size_t)T * A.max_T + t];
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
    if (r->wtab) (void)hipFree(r->wtab);
    if (r->err) (void)hipFree(r->err);
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

int af_replay_set_weights(af_replay* r, const float* table_host, int32_t max_T) {
    if (!r || !table_host || max_T < 1) return AF_REPLAY_ERR_ARG;
    RP_HIP_OK(hipSetDevice(r->device));
    RP_HIP_OK(hipDeviceSynchronize());                 // no append may still be reading the old table
    if (r->wtab) (void)hipFree(r->wtab);
    r->wtab = nullptr; r->max_T = 0;
    const size_t bytes = (size_t)(max_T + 1) * max_T * 4;
    RP_HIP_OK(hipMalloc(reinterpret_cast<void**>(&r->wtab), bytes));
    RP_HIP_OK(hipMemcpy(r->wtab, table_host, bytes, hipMemcpyHostToDevice));
    if (!r->err) {
        RP_HIP_OK(hipMalloc(reinterpret_cast<void**>(&r->err), 4));
        RP_HIP_OK(hipMemset(r->err, 0, 4));
    }
    r->max_T = max_T;
    return AF_REPLAY_OK;
}

int af_replay_append_packed(af_replay* r, void* stream, const int32_t* packed_dev, int32_t max_episodes, int32_t episode, int32_t T) {
    if (!r || !packed_dev || max_episodes < 1 || episode < 0 || episode >= max_episodes || T < 1) return AF_REPLAY_ERR_ARG;
    if (!r->wtab || T > r->max_T) return AF_REPLAY_ERR_ARG;      // af_replay_set_weights first
    if (r->count + T > r->cap) return AF_REPLAY_ERR_FULL;
    RP_HIP_OK(hipSetDevice(r->device));
    PackedArgs a;
    a.buf = packed_dev; a.boards = r->boards; a.policies = r->policies; a.last = r->last; a.values = r->values; a.weights = r->weights;
    a.wtab = r->wtab; a.err = r->err; a.slot0 = (r->head + r->count) % r->cap; a.cap = r->cap; a.C = r->C;
    a.max_eps = max_episodes; a.ep = episode; a.T = T; a.max_T = r->max_T;
    hipLaunchKernelGGL(af_replay_append_packed_kernel, dim3(T), dim3(256), 0, static_cast<hipStream_t>(stream), a);
    RP_HIP_OK(hipGetLastError());
    r->count += T;
    return AF_REPLAY_OK;
}

/* 1 if a packed append found the buffer not to hold what the host said (and appended nothing); clears the flag */
int af_replay_check(af_replay* r, void* stream) {
    if (!r) return AF_REPLAY_ERR_ARG;
    if (!r->err) return 0;
    int32_t h = 0;
    RP_HIP_OK(hipSetDevice(r->device));
    RP_HIP_OK(hipMemcpyAsync(&h, r->err, 4, hipMemcpyDeviceToHost, static_cast<hipStream_t>(stream)));
    RP_HIP_OK(hipStreamSynchronize(static_cast<hipStream_t>(stream)));
    if (h) { RP_HIP_OK(hipMemset(r->err, 0, 4)); return AF_REPLAY_ERR_RANGE; }
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
