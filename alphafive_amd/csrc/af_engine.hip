// af_engine.hip — MI355X (gfx950) self-play engine: thousands of concurrent Gomoku
// MCTS games resident in HBM, one 64-lane wavefront per game.
//
// What the reference does per simulation in Python (genData/player.py:204-228
// MCTS_search, :230 select_action_q_and_u, :186 evaluate_and_expand, :166
// update_tree, :84 calc_policy; utils.py:199 is_game_over, :238 get_legal_actions,
// :256 board_to_inputs, :275 step) is done here by ONE kernel launch per tick for
// all games: the wave that owns a game consumes the evaluation of the leaf it
// parked on, then keeps descending / backing up / finishing moves until it parks
// on the next unseen position, whose input planes it writes for the batched net.
//
// Layout (SoA, one slice per game, everything stays resident between ticks):
//   position key   2*KW u64 : "mine" bitboard, "theirs" bitboard (bit c = cell i*S+j)
//   node stats     dense by cell, CP = 64*KW slots per node: N (int32, bit31 = "W is
//                  fp32-typed", SURVEY §8a rule 2), W (fp32 running sum), P (fp32 prior)
//                  lane l owns cells l, l+64, ...  -> every stats access is one
//                  coalesced 256-byte row per array
//   store          per-game open-addressing table (u32 slot = node index + 1) over an
//                  exact-key node pool; unreachable nodes (stones not a superset of
//                  the root's) are compacted away under capacity pressure
// Arithmetic follows SURVEY §8a to the bit: fp32 sequential prior sum, fp32 W, fp64
// PUCT score rounded to fp32, ties on the fp32 score.  Compiled -ffp-contract=off.

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <type_traits>
#include <vector>

#include "af_engine.h"
#include "af_noise.h"

typedef unsigned long long u64;

enum { PH_IDLE = 0, PH_MOVE_START = 1, PH_SEARCH = 2, PH_MOVE_DONE = 3, PH_ERROR = 4, PH_DESCENT = 5 /* yielded inside a simulation */ };
enum { CT_SIMS = 0, CT_SELECTS, CT_EXPANDS, CT_TERMINALS, CT_PLIES, CT_EPISODES, CT_LSUM, CT_LEXP,
       CT_GC, CT_GC_SCANNED, CT_YIELDS, CT_STALLS, CT_N };
enum { HIST_WORK = 0, HIST_TIME = 64, HIST_MAXTIME = 96, HIST_N = 97 };   // selects per launch | 8-us wave-time bins | max wave time (10 ns units)

struct EngineParams {
    int G, S, C, goal, sims, upper, training, mode, node_cap, max_ply, budget, budget_hard;
    uint32_t hash_mask;
    double init_temp, tau_decay, tau_decay_r, alpha, c_puct;
    float c_puct32;
    uint32_t k0, seed_hi, first_game_id;
    u64 colmask[4], boardmask[4];
    // per-game state
    int32_t *phase, *pending, *sims_left, *ply, *root_last, *nodes, *nfree, *depth, *leaf_last, *leaf_slot, *status, *random_a;
    int32_t *action, *has_policy, *visits;
    float* policy;
    u64 *root, *leaf;
    double* tau;
    uint32_t *episode, *sel, *plyctr;
    int32_t *path_node, *path_cell;
    // transposition store
    uint32_t* hash;
    u64* node_key;
    int32_t* node_sum;    // sum_n; -1 = slot is free (dropped by the collector)
    int32_t* free_idx;    // [G][cap] stack of free slots below the high-water mark `nodes`
    int32_t* edge_n;
    float *edge_w, *edge_p;
    int32_t* edge_c;      // [G][cap][CP] child cache: node index + 1 of the position reached by playing that cell, 0 = not known yet
    double* edge_w64;     // AF_MODE_VALUE_F64 (pipe path, networkAPI.py:72): W as a python float; edge_w is then unused
    int32_t w64;
    // finished-episode double buffers
    uint32_t *ep_seq, *ep_popped;
    int32_t* ep_len;
    float* ep_final;
    u64* rec_key;
    float* rec_policy;
    int32_t *rec_visits, *rec_last, *rec_action;
    u64* counters;
    u64* progress;   // [0] plies committed, [1] episodes finished (all games)
    u64* hist;       // HIST_N words, all games
    // evaluation memo (ABI v5, off unless af_engine_memo_enable was called): (position, last move) -> the net's policy and value,
    // shared by all games of the engine.  4-way buckets; a launch only reads it (the tick kernel) or only writes it (af_memo_insert)
    u64* memo_key;   // [4 * buckets][4 KW]: 2 KW key words, (epoch << 32 | last + 1), zeros; an entry whose epoch is not the
                     // launch's memo_epoch is empty (all zeros = never written; older epochs = another weight set's bits)
    float* memo_pv;  // [4 * buckets][CP]: C policy floats as the net wrote them
    float* memo_val; // [4 * buckets]: the value (an array of its own: C == CP on a 16x16 board leaves no spare slot in the row)
    u64* memo_lock;  // [4 * buckets]: serial of the launch that last wrote the entry (one writer per entry and launch)
    u64* memo_stats; // MEMO_SERIAL, MEMO_PROBES, MEMO_HITS, MEMO_INSERTS, MEMO_REPLACED
    uint32_t memo_bucket_mask;
    int memo_max_stones;
    uint32_t memo_epoch; // >= 1; af_engine_memo_clear bumps it (O(1): stale entries simply stop matching)
    int memo_budget;     // selects per launch after which a game that has consumed a memo hit yields (AF_MEMO_BUDGET; default min(budget, 6))
};
enum { MEMO_SERIAL = 0, MEMO_PROBES, MEMO_HITS, MEMO_INSERTS, MEMO_REPLACED, MEMO_N };

// ----------------------------------------------------------------------------------------------
// device helpers
// ----------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t rfl32(uint32_t x) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)x); }
__device__ __forceinline__ int rfli(int x) { return __builtin_amdgcn_readfirstlane(x); }
__device__ __forceinline__ u64 rfl64(u64 x) {
    return ((u64)rfl32((uint32_t)(x >> 32)) << 32) | (u64)rfl32((uint32_t)x);
}
__device__ __forceinline__ float rflf(float x) { return __int_as_float(rfli(__float_as_int(x))); }
// Wave-wide maximum, the same value in every lane, on the DPP cross-lane paths of the VALU (quad permutes, row mirrors, row
// broadcasts) + one v_readlane: ~10 instructions where six __shfl_xor rounds cost six ds_bpermute round trips through the LDS
// crossbar (~100 cycles each) on the critical path of every select.  max is exact and order-free, so the result is the same bits.
template <int CTRL, int ROWS = 0xF>
__device__ __forceinline__ int dpp_i(int v) { return __builtin_amdgcn_update_dpp(v, v, CTRL, ROWS, 0xF, false); }
__device__ __forceinline__ int wave_max_i32(int v) {
    int o;
    o = dpp_i<0xB1>(v); v = o > v ? o : v;            // quad_perm [1,0,3,2]
    o = dpp_i<0x4E>(v); v = o > v ? o : v;            // quad_perm [2,3,0,1]
    o = dpp_i<0x141>(v); v = o > v ? o : v;           // row_half_mirror
    o = dpp_i<0x140>(v); v = o > v ? o : v;           // row_mirror: every lane of a 16-lane row holds the row's maximum
    o = dpp_i<0x142, 0xA>(v); v = o > v ? o : v;      // row_bcast:15 into rows 1 and 3
    o = dpp_i<0x143, 0xC>(v); v = o > v ? o : v;      // row_bcast:31 into rows 2 and 3: lane 63 holds the wave's maximum
    return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ float wave_max_f32(float v) {
    float o;
    o = __int_as_float(dpp_i<0xB1>(__float_as_int(v))); v = o > v ? o : v;
    o = __int_as_float(dpp_i<0x4E>(__float_as_int(v))); v = o > v ? o : v;
    o = __int_as_float(dpp_i<0x141>(__float_as_int(v))); v = o > v ? o : v;
    o = __int_as_float(dpp_i<0x140>(__float_as_int(v))); v = o > v ? o : v;
    o = __int_as_float(dpp_i<0x142, 0xA>(__float_as_int(v))); v = o > v ? o : v;
    o = __int_as_float(dpp_i<0x143, 0xC>(__float_as_int(v))); v = o > v ? o : v;
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// Wave-wide fp64 sum with the summation tree of the tier-B specification (include/af_noise.h, oracle afo_noise_philox_dirichlet:
// acc = acc + partner(lane ^ off) for off = 32, 16, 8, 4, 2, 1), on VALU cross-lane paths instead of six ds_bpermute round trips per
// half: v_permlane32_swap / v_permlane16_swap (gfx950) hand every lane its own and its partner's value as the pair (r[0], r[1]) in one
// order or the other — fp64 addition is commutative, so r[0] + r[1] has the bits of own + partner; row_ror:8 is lane ^ 8 inside a
// 16-lane row; after that step the values are 8-periodic, so row_ror:4 delivers the value of lane ^ 4; quad permutes do ^ 2 and ^ 1.
// (A/B records, removed from the source in r6: clean-up rounds with two attempts side by side — same bits, +5.5 % per launch,
//  profiles/r5_03_tick_ab.txt; round 0 of a lane's cells one after the other instead of side by side — r4_15)

template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
    const int lo = __double2loint(v), hi = __double2hiint(v);
    return __hiloint2double(__builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xF, 0xF, false), __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xF, 0xF, false));
}
__device__ __forceinline__ double wave_sum_f64_tree(double acc) {
    {
        const int lo = __double2loint(acc), hi = __double2hiint(acc);
        const auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false), b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
        acc = __hiloint2double(b[0], a[0]) + __hiloint2double(b[1], a[1]);
    }
    {
        const int lo = __double2loint(acc), hi = __double2hiint(acc);
        const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false), b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
        acc = __hiloint2double(b[0], a[0]) + __hiloint2double(b[1], a[1]);
    }
    acc = acc + dpp_f64<0x128>(acc);      // row_ror:8
    acc = acc + dpp_f64<0x124>(acc);      // row_ror:4
    acc = acc + dpp_f64<0x4E>(acc);       // quad_perm [2,3,0,1]
    acc = acc + dpp_f64<0xB1>(acc);       // quad_perm [1,0,3,2]
    return acc;
}

template <int KW>
__device__ __forceinline__ void bb_shr(const u64* a, int d, u64* o) {   // 0 < d < 64
#pragma unroll
    for (int k = 0; k < KW; ++k) o[k] = (a[k] >> d) | (k + 1 < KW ? (a[k + 1] << (64 - d)) : 0ull);
}
template <int KW>
__device__ __forceinline__ void bb_shl(const u64* a, int d, u64* o) {
#pragma unroll
    for (int k = KW - 1; k >= 0; --k) o[k] = (a[k] << d) | (k > 0 ? (a[k - 1] >> (64 - d)) : 0ull);
}
template <int KW>
__device__ __forceinline__ int bb_first(const u64* a) {   // lowest set bit index or -1
#pragma unroll
    for (int k = 0; k < KW; ++k)
        if (a[k]) return 64 * k + __builtin_ctzll(a[k]);
    return -1;
}
template <int KW>
__device__ __forceinline__ int bb_count(const u64* a) {
    int n = 0;
#pragma unroll
    for (int k = 0; k < KW; ++k) n += __popcll(a[k]);
    return n;
}
template <int KW>
__device__ __forceinline__ int bb_select(const u64* a, int r) {   // index of the r-th set bit (ascending)
#pragma unroll
    for (int k = 0; k < KW; ++k) {
        int pc = __popcll(a[k]);
        if (r < pc) {
            u64 x = a[k];
            for (int i = 0; i < r; ++i) x &= x - 1;
            return 64 * k + __builtin_ctzll(x);
        }
        r -= pc;
    }
    return -1;
}
// The same on the lanes (r4): lane l asks "is bit l of word k set and are exactly r bits set below it?" — v_mbcnt counts a mask's
// bits below the lane — and a ballot names the one lane that says yes.  The scalar loop above clears r bits one by one: ~5 SALU
// instructions per bit, and a forced root visit (player.py:264-276, half of all root selects) picks among up to 121 candidates.
template <int KW>
__device__ __forceinline__ int bb_select_wave(const u64* a, int r, int lane) {   // a and r wave-uniform; -1 if r >= popcount
    int base = 0, out = -1;
#pragma unroll
    for (int k = 0; k < KW; ++k) {
        const int below = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(a[k] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)a[k], 0u));
        const u64 hit = __ballot(((a[k] >> lane) & 1ull) && base + below == r);
        if (hit) out = 64 * k + __builtin_ctzll(hit);
        base += __popcll(a[k]);
    }
    return out;
}
template <int KW>
__device__ __forceinline__ bool bb_test(const u64* a, int c) {
    bool r = false;
#pragma unroll
    for (int k = 0; k < KW; ++k) r = (k == (c >> 6)) ? (((a[k] >> (c & 63)) & 1ull) != 0) : r;
    return r;
}
// register-array friendly forms (no dynamic indexing => no scratch): set bit c / pick key word `lane`
template <int KW>
__device__ __forceinline__ void bb_set(u64* a, int c) {
#pragma unroll
    for (int k = 0; k < KW; ++k) a[k] |= (k == (c >> 6)) ? (1ull << (c & 63)) : 0ull;
}
template <int KW>
__device__ __forceinline__ u64 key_word(const u64* mine, const u64* theirs, int lane) {
    u64 v = 0;
#pragma unroll
    for (int k = 0; k < KW; ++k) v = lane == k ? mine[k] : (lane == KW + k ? theirs[k] : v);
    return v;
}

// run masks: bit c set iff `goal` consecutive stones start at c along the direction
template <int KW, bool LEFT>
__device__ __forceinline__ void run_mask(const u64* b, int d, int goal, const u64* valid, u64* out) {
    u64 cur[KW], nxt[KW];
#pragma unroll
    for (int k = 0; k < KW; ++k) { cur[k] = b[k]; out[k] = b[k]; }
    for (int t = 1; t < goal; ++t) {
        if (LEFT) bb_shl<KW>(cur, d, nxt); else bb_shr<KW>(cur, d, nxt);
#pragma unroll
        for (int k = 0; k < KW; ++k) { cur[k] = nxt[k]; out[k] &= nxt[k]; }
    }
    if (valid) {
#pragma unroll
        for (int k = 0; k < KW; ++k) out[k] &= valid[k];
    }
}

// utils.py:199-235 is_game_over on bitboards.  Scan order of the reference (cell-major,
// then down / right / diag-down-right / diag-up-right) decides when both colours have a line.
// MOVED: the position was reached by a move from a position that was not terminal (every position of a descent: its parent was
// tested one level up, the root at the start of the move), so the side to move cannot own a line — only the stones of the side
// that has just moved are scanned (half the work; the result is the same)
template <int KW, bool MOVED = false>
__device__ int terminal_test(const EngineParams& P, const u64* mine, const u64* theirs, float* value) {
    int best_key = 0x7fffffff;
    float best_v = 0.0f;
    const int S = P.S;
#pragma unroll
    for (int dir = 0; dir < 4; ++dir) {
        u64 wm[KW], wt[KW], any[KW];
        const int d = dir == 0 ? S : (dir == 1 ? 1 : (dir == 2 ? S + 1 : S - 1));
        const u64* valid = dir == 0 ? nullptr : P.colmask;
        if (MOVED) {
#pragma unroll
            for (int k = 0; k < KW; ++k) wm[k] = 0;
            if (dir == 3) run_mask<KW, true>(theirs, d, P.goal, valid, wt);
            else run_mask<KW, false>(theirs, d, P.goal, valid, wt);
        } else if (dir == 3) {
            run_mask<KW, true>(mine, d, P.goal, valid, wm);
            run_mask<KW, true>(theirs, d, P.goal, valid, wt);
        } else {
            run_mask<KW, false>(mine, d, P.goal, valid, wm);
            run_mask<KW, false>(theirs, d, P.goal, valid, wt);
        }
#pragma unroll
        for (int k = 0; k < KW; ++k) any[k] = wm[k] | wt[k];
        const int f = bb_first<KW>(any);
        if (f >= 0) {
            const int key = f * 4 + dir;
            if (key < best_key) { best_key = key; best_v = bb_test<KW>(wm, f) ? 1.0f : -1.0f; }
        }
    }
    if (best_key != 0x7fffffff) { *value = best_v; return 1; }
    bool full = true;
#pragma unroll
    for (int k = 0; k < KW; ++k) full = full && ((mine[k] | theirs[k]) == P.boardmask[k]);
    *value = 0.0f;
    return full ? 1 : 0;
}

// The same test inside a descent (depth > 0), where the position was reached by playing `last` from a position that was NOT
// terminal: a line can only have been completed by that stone, so the 64 lanes look at the cells of the four lines through
// `last` instead of the scalar unit shifting whole bitboards (~400 SALU instructions per select, r4): 16 lanes per direction, lane
// j of a group takes the cell j - (goal - 1) steps along it, a ballot packs the mover's stones into four 16-bit windows (bits
// 2*goal-1.. of a window stay 0) and `goal` of them in a row is w & w>>1 & ... & w>>(goal-1) != 0 on the 64-bit mask (a window's
// upper zeros stop runs from crossing into the next).  Only the mover can own a line (value -1 for the side to move); a full board
// without one is a draw: the same (terminal, value) as terminal_test<KW, true> — utils.py:199-235 on such positions.  goal <= 8.
template <int KW>
__device__ __forceinline__ int terminal_through_last(const EngineParams& P, const u64* mine, const u64* theirs, int last, uint32_t inv_s16,
                                                     int lane, float* value) {
    const int S = P.S, g1 = P.goal - 1;
    const int y0 = (int)(((uint32_t)last * inv_s16) >> 16), x0 = last - y0 * S;        // (exact for last < 256, S <= 16)
    const int dir = lane >> 4, j = (lane & 15) - g1;
    const int dy = dir == 1 ? 0 : (dir == 3 ? -1 : 1), dx = dir == 0 ? 0 : 1;         // down | right | down-right | up-right
    const int y = y0 + j * dy, x = x0 + j * dx;
    const bool inb = (lane & 15) <= 2 * g1 && (unsigned)y < (unsigned)S && (unsigned)x < (unsigned)S;
    const int c = inb ? y * S + x : 0;
    u64 word = theirs[0];
#pragma unroll
    for (int k = 1; k < KW; ++k) word = (c >> 6) == k ? theirs[k] : word;
    const u64 w = __ballot(inb && ((word >> (c & 63)) & 1ull));
    u64 r = w;
    for (int t = 1; t <= g1; ++t) r &= w >> t;
    if (r) { *value = -1.0f; return 1; }
    bool full = true;
#pragma unroll
    for (int k = 0; k < KW; ++k) full = full && ((mine[k] | theirs[k]) == P.boardmask[k]);
    *value = 0.0f;
    return full ? 1 : 0;
}

template <int KW>
__host__ __device__ __forceinline__ uint32_t key_hash(const u64* mine, const u64* theirs) {
    u64 h = 0x9E3779B97F4A7C15ull;
#pragma unroll
    for (int k = 0; k < KW; ++k) {
        h = (h ^ mine[k]) * 0xff51afd7ed558ccdull;
        h ^= h >> 29;
        h = (h ^ theirs[k]) * 0xc4ceb9fe1a85ec53ull;
        h ^= h >> 32;
    }
    return (uint32_t)h;
}

// Wave-cooperative exact-key lookup: 64 consecutive slots per round.
// Returns node index or -1; *slot_out = slot where the key would be inserted.
template <int KW>
__device__ int tree_lookup(const EngineParams& P, int g, const u64* mine, const u64* theirs, int lane, uint32_t* slot_out) {
    const uint32_t* slots = P.hash + (size_t)g * (P.hash_mask + 1);
    const u64* keys = P.node_key + (size_t)g * P.node_cap * (2 * KW);
    const uint32_t h = key_hash<KW>(mine, theirs);
    for (uint32_t round = 0; round <= P.hash_mask / 64 + 1; ++round) {
        const uint32_t s = (h + round * 64 + lane) & P.hash_mask;
        const uint32_t v = slots[s];
        const u64 empties = __ballot(v == 0);
        const int first_empty = empties ? __builtin_ctzll(empties) : 64;
        bool match = false;
        if (v != 0 && lane < first_empty) {
            const u64* kk = keys + (size_t)(v - 1) * (2 * KW);
            match = true;
#pragma unroll
            for (int k = 0; k < KW; ++k) match = match && kk[k] == mine[k] && kk[KW + k] == theirs[k];
        }
        const u64 mm = __ballot(match);
        if (mm) {
            const int src = __builtin_ctzll(mm);
            return __builtin_amdgcn_readlane((int)v, src) - 1;          // (src is wave-uniform: it comes from a ballot)
        }
        if (first_empty < 64) {
            *slot_out = (h + round * 64 + first_empty) & P.hash_mask;
            return -1;
        }
    }
    *slot_out = 0xffffffffu;
    return -1;
}

// ---- evaluation memo ----
// The net is a pure function of (stones, last move) and the forward is independent of batch slot and batch size to the bit
// (tests/test_gpu_net.py), so the evaluation of a position ANY game of the engine has parked on before can be consumed at once
// instead of waiting a tick for the same bits.  Every game still owns its tree (player.py:38: one tree per Player): only the
// pipe round trip of player.py:194-197 is short-cut.  One bucket = 4 entries of 4 KW words = one word per lane.
template <int KW>
__device__ __forceinline__ uint32_t memo_bucket(const EngineParams& P, const u64* mine, const u64* theirs, int last) {
    uint32_t h = key_hash<KW>(mine, theirs) ^ ((uint32_t)(last + 1) * 0x9E3779B1u);
    h ^= h >> 15; h *= 0x85EBCA6Bu; h ^= h >> 13;
    return h & P.memo_bucket_mask;
}
template <int KW>
__device__ __forceinline__ u64 memo_word(const EngineParams& P, const u64* mine, const u64* theirs, int last, int w) {   // word w of the entry's key row
    const u64 kw = key_word<KW>(mine, theirs, w);                                                // (0 for w >= 2 KW)
    return w == 2 * KW ? (((u64)P.memo_epoch << 32) | (u64)(uint32_t)(last + 1)) : kw;
}
// -> entry index or -1; *empty_ways = bit v set iff way v of the bucket holds nothing of this epoch (never written, or cleared)
template <int KW>
__device__ __forceinline__ int memo_probe(const EngineParams& P, uint32_t bucket, const u64* mine, const u64* theirs, int last, int lane,
                                          uint32_t* empty_ways) {
    constexpr int MK = 4 * KW;
    const int w = lane & (MK - 1);
    const u64 expect = memo_word<KW>(P, mine, theirs, last, w);
    u64 have = expect;
    if (lane < 4 * MK) have = P.memo_key[(size_t)bucket * (4 * MK) + lane];
    const u64 bad = __ballot(have != expect);
    const u64 blank = __ballot(lane < 4 * MK && w == 2 * KW && (uint32_t)(have >> 32) != P.memo_epoch);
    uint32_t em = 0;
    int hit = -1;
#pragma unroll
    for (int v = 3; v >= 0; --v) {
        const u64 wm = (MK == 16 ? 0xffffull : 0xffull) << (v * MK);
        if ((bad & wm) == 0ull) hit = (int)bucket * 4 + v;
        if (blank & wm) em |= 1u << v;
    }
    *empty_ways = em;
    return hit;
}


// numpy float32 add.reduce restated (pairwise, 8 accumulators) over an LDS array
__device__ float pairwise_block(const float* a, int n) {
    if (n < 8) {
        float r = 0.0f;
        for (int i = 0; i < n; ++i) r = r + a[i];
        return r;
    }
    float r0 = a[0], r1 = a[1], r2 = a[2], r3 = a[3], r4 = a[4], r5 = a[5], r6 = a[6], r7 = a[7];
    int i;
    for (i = 8; i < n - (n % 8); i += 8) {
        r0 = r0 + a[i]; r1 = r1 + a[i + 1]; r2 = r2 + a[i + 2]; r3 = r3 + a[i + 3];
        r4 = r4 + a[i + 4]; r5 = r5 + a[i + 5]; r6 = r6 + a[i + 6]; r7 = r7 + a[i + 7];
    }
    float res = ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7));
    for (; i < n; ++i) res = res + a[i];
    return res;
}
__device__ float pairwise_sum(const float* a, int n) {   // n <= 256
    if (n <= 128) return pairwise_block(a, n);
    int n2 = n / 2;
    n2 -= n2 % 8;
    return pairwise_block(a, n2) + pairwise_block(a + n2, n - n2);
}

// ----------------------------------------------------------------------------------------------
// the tick kernel: one wavefront (= one workgroup) per game
// ----------------------------------------------------------------------------------------------
#ifndef AF_TICK_MIN_WAVES
#define AF_TICK_MIN_WAVES 4   // 4 one-wave workgroups per SIMD = 16 games per CU in flight (measured 0.075 vs 0.088 ms per tick at 3)
#endif
#ifdef AF_TICK_TIMING
// profiling build only (tools/probe_tick_timing.py): shader cycles of the last launch per game and phase —
// 0 state load, 1 consume (expand + backup), 2 move boundary (calc_policy, record, collector), 3 terminal test + store lookup,
// 4 select (rows, noise, score, argmax, step), 5 park / yield + state store, 6 total, 7 selects
__device__ unsigned long long g_tick_cycles[8192][13];      // 8..11: inside the select: rows + noise rounds | noise sum + normalise | scores | argmax, pick, prefetch, step
#define TK_T(var) const unsigned long long var = __builtin_readcyclecounter()
#define TK_ACC(slot, a, b) tk[slot] += (b) - (a)
#else
#define TK_T(var)
#define TK_ACC(slot, a, b)
#endif

template <int KW, bool W64, bool MEMO = false>
__global__ __launch_bounds__(64, AF_TICK_MIN_WAVES) void af_tick_kernel(EngineParams P, const float* __restrict__ policy_in,
                                                     const float* __restrict__ value_in, float* __restrict__ planes_out) {
    constexpr int CP = 64 * KW;
    __shared__ float s_pv[CP];
    __shared__ double s_cdf[CP];
    const int g = blockIdx.x;
    const int lane = threadIdx.x;
    const int C = P.C, NCAP = P.node_cap;
    if constexpr (MEMO) {                            // the launch serial af_memo_insert claims entries with
        if (g == 0 && lane == 0) atomicAdd(reinterpret_cast<unsigned long long*>(&P.memo_stats[MEMO_SERIAL]), 1ull);
    }

    int phase = rfli(P.phase[g]);
    if (phase == PH_IDLE || phase == PH_MOVE_DONE || phase == PH_ERROR) return;
    const u64 t_start = wall_clock64();
#ifdef AF_TICK_TIMING
    unsigned long long tk[13] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};      // 12: rejection-loop iterations
#endif
    TK_T(tk_entry);

    // ---- load game state (wave-uniform) ----
    u64 root_m[KW], root_t[KW];
#pragma unroll
    for (int k = 0; k < KW; ++k) {
        root_m[k] = rfl64(P.root[(size_t)g * 2 * KW + k]);
        root_t[k] = rfl64(P.root[(size_t)g * 2 * KW + KW + k]);
    }
    int root_last = rfli(P.root_last[g]);
    int sims_left = rfli(P.sims_left[g]);
    int ply = rfli(P.ply[g]);
    int nodes = rfli(P.nodes[g]);      // high-water mark of the node pool
    int nfree = rfli(P.nfree[g]);      // free slots below it (left by the collector)
    double tau = P.tau[g];
    uint32_t episode = rfl32(P.episode[g]);
    uint32_t sel = rfl32(P.sel[g]);
    uint32_t plyctr = rfl32(P.plyctr[g]);
    const int random_a = rfli(P.random_a[g]);
    const uint32_t inv_s16 = (65536u + (uint32_t)P.S - 1u) / (uint32_t)P.S;          // n / S = (n * inv_s16) >> 16 for n < 256
    const uint32_t k0 = P.k0, k1 = P.seed_hi + (P.first_game_id + (uint32_t)g);
    uint32_t ct[CT_N];   // this launch's share of the per-game counters
#pragma unroll
    for (int i = 0; i < CT_N; ++i) ct[i] = 0;
    int err = 0;

    u64* const nkeys = P.node_key + (size_t)g * NCAP * (2 * KW);
    int32_t* const nsum = P.node_sum + (size_t)g * NCAP;
    int32_t* const fre = P.free_idx + (size_t)g * NCAP;
    int32_t* const en = P.edge_n + (size_t)g * NCAP * CP;
    constexpr bool w64 = W64;                        // fp64 W rows (pipe path) are a separate instantiation: no live range of the other kind
    float* const ew = P.edge_w + (w64 ? (size_t)0 : (size_t)g * NCAP * CP);
    double* const ew64 = P.edge_w64 + (w64 ? (size_t)g * NCAP * CP : (size_t)0);
    float* const ep = P.edge_p + (size_t)g * NCAP * CP;
    int32_t* const ec = P.edge_c + (size_t)g * NCAP * CP;
    uint32_t* const slots = P.hash + (size_t)g * (P.hash_mask + 1);
    int32_t* const pnode = P.path_node + (size_t)g * CP;
    int32_t* const pcell = P.path_cell + (size_t)g * CP;

    // backup along the stored path: player.py:166-184 update_tree.  Path element d receives
    // v * (-1)^(depth-d); is_f32 marks net values (np.float32) vs terminal values (python float).
    auto backup = [&](int depth, float v, int is_f32) {
        for (int d = lane; d < depth; d += 64) {
            const size_t off = (size_t)pnode[d] * CP + pcell[d];
            const float vv = ((depth - d) & 1) ? -v : v;
            const int32_t raw = en[off];
            if (w64) {                                      // float(v): every addend is a python float
                en[off] = raw + 1;
                ew64[off] = ew64[off] + (double)vv;
            } else {
                en[off] = (raw + 1) | (is_f32 ? (int32_t)0x80000000 : 0);
                ew[off] = ew[off] + vv;
            }
        }
        __syncthreads();
    };

    TK_T(tk_loaded);
    TK_ACC(0, tk_entry, tk_loaded);
    // ---- 1. consume the evaluation of the parked leaf: player.py:186-202 + :166 ----
    // (MEMO: also of a leaf whose evaluation the memo holds — found in the descent below, consumed here, and the game goes on)
    u64 lm[KW], lt[KW];
    int c_depth = 0;
    uint32_t c_slot = 0;
    const float* c_pol = policy_in + (size_t)g * C;
    const float* c_val = value_in + g;
    bool have_eval = rfli(P.pending[g]) != 0;
    if (have_eval) {
#pragma unroll
        for (int k = 0; k < KW; ++k) {
            lm[k] = rfl64(P.leaf[(size_t)g * 2 * KW + k]);
            lt[k] = rfl64(P.leaf[(size_t)g * 2 * KW + KW + k]);
        }
        c_depth = rfli(P.depth[g]);
        c_slot = rfl32((uint32_t)P.leaf_slot[g]);
    }
    int status = AF_STATUS_IDLE;
    bool parked = false;
    int work = 0;   // selects done in this launch
    uint32_t memo_probes = 0, memo_hits = 0;
  for (;;) {
    if (have_eval) {
        have_eval = false;
        u64 legal[KW];
#pragma unroll
        for (int k = 0; k < KW; ++k) legal[k] = ~(lm[k] | lt[k]) & P.boardmask[k];
        const int depth = c_depth;
        const uint32_t slot = c_slot;
        float pk[KW];
#pragma unroll
        for (int k = 0; k < KW; ++k) {
            const int c = lane + 64 * k;
            pk[k] = ((legal[k] >> lane) & 1ull) ? c_pol[c] : 0.0f;
        }
        // all_p: left-to-right fp32 sum in row-major legal order (SURVEY §8a rule 1; the zeros of the occupied cells change nothing).
        // Every lane adds the same broadcast LDS values in the same order: a chain of 64 KW dependent adds with the operands
        // fetched four at a time, instead of a v_readlane + hazard + add per cell
#pragma unroll
        for (int k = 0; k < KW; ++k) s_pv[lane + 64 * k] = pk[k];
        __syncthreads();
        float s = 0.0f;
#pragma unroll
        for (int i = 0; i < CP; i += 4) {
            const float4 v4 = *reinterpret_cast<const float4*>(&s_pv[i]);
            s = s + v4.x; s = s + v4.y; s = s + v4.z; s = s + v4.w;
        }
        s = rflf(s);
        if (!((double)s >= 1e-5)) s = (float)1e-5;
        if ((nfree == 0 && nodes >= NCAP) || slot == 0xffffffffu) {
            err = AF_ERR_NODE_CAP;
        } else {
            int idx;
            if (nfree > 0) idx = rfli(fre[--nfree]); else idx = nodes++;
            if (lane < 2 * KW) nkeys[(size_t)idx * 2 * KW + lane] = key_word<KW>(lm, lt, lane);
            if (lane == 0) {
                nsum[idx] = 0; slots[slot] = (uint32_t)idx + 1u;
                // child cache of the parent edge (see the descent below): the move that led here now has a node
                if (depth > 0) ec[(size_t)pnode[depth - 1] * CP + pcell[depth - 1]] = idx + 1;
            }
#pragma unroll
            for (int k = 0; k < KW; ++k) {
                const size_t off = (size_t)idx * CP + lane + 64 * k;
                en[off] = 0;
                ec[off] = 0;
                if (w64) ew64[off] = 0.0; else ew[off] = 0.0f;
                ep[off] = ((legal[k] >> lane) & 1ull) ? pk[k] / s : 0.0f;
            }
            __syncthreads();
            backup(depth, rflf(*c_val), 1);
            ct[CT_EXPANDS]++;
            ct[CT_LEXP] += (uint32_t)bb_count<KW>(legal);
        }
        ct[CT_SIMS]++;
        sims_left--;
    }

    TK_T(tk_consumed);
    TK_ACC(1, tk_loaded, tk_consumed);
    // ---- 2. advance until the game parks ----
    while (!parked && !err && !(MEMO && have_eval)) {
        TK_T(tk_iter);
        // A launch lasts as long as its slowest game.  Simulations that end in a terminal position need no
        // evaluation, so a game whose root has a decided child could run hundreds of them back to back while
        // every other wave has long parked; after `budget` selects the game yields at the next simulation
        // boundary instead (its slot of the leaf batch idles for one tick; the order of simulations inside
        // the game, and so every result, is unchanged).
        if (work >= ((MEMO && memo_hits) ? P.memo_budget : P.budget) && phase != PH_DESCENT) { status = AF_STATUS_YIELD; parked = true; ct[CT_YIELDS]++; break; }
        if (phase == PH_MOVE_START) {
            // SELFPLAY back-pressure (the reference's Queue(50), main.py:51,94): an episode may only start
            // when its record buffer has been popped; until then the game waits here.
            if (P.mode == AF_MODE_SELFPLAY && ply == 0 && rfl32(P.ep_seq[g]) - rfl32(P.ep_popped[g]) >= 2u) {
                status = AF_STATUS_YIELD; parked = true; ct[CT_STALLS]++;
                break;
            }
            // capacity pressure: drop nodes whose stones are not a superset of the root's (they can
            // never be reached again: play only adds stones; the reference never revisits them either).
            // Wave-parallel mark phase, 64 nodes per step: nothing moves — dropped slots go on the
            // game's free stack (reused by later expands) and the hash is rebuilt from the survivors.
            bool collected = false;
            if (nodes - nfree + P.sims + 2 > NCAP && nodes > 0) {
                for (uint32_t s_ = lane; s_ <= P.hash_mask; s_ += 64) slots[s_] = 0;
                __syncthreads();
                const int rc = bb_count<KW>(root_m) + bb_count<KW>(root_t);
                int nf = 0;
                for (int base = 0; base < nodes; base += 64) {
                    const int i = base + lane;
                    const bool valid = i < nodes;
                    bool keep = false;
                    if (valid && nsum[i] >= 0) {
                        u64 nm[KW], nt[KW];
#pragma unroll
                        for (int k = 0; k < KW; ++k) { nm[k] = nkeys[(size_t)i * 2 * KW + k]; nt[k] = nkeys[(size_t)i * 2 * KW + KW + k]; }
                        const int nc = bb_count<KW>(nm) + bb_count<KW>(nt);
                        keep = true;
                        // colour of "mine" is fixed by stone-count parity
#pragma unroll
                        for (int k = 0; k < KW; ++k) {
                            const u64 a = ((nc ^ rc) & 1) ? nt[k] : nm[k];
                            const u64 b = ((nc ^ rc) & 1) ? nm[k] : nt[k];
                            keep = keep && ((a & root_m[k]) == root_m[k]) && ((b & root_t[k]) == root_t[k]);
                        }
                        if (keep) {
                            uint32_t h = key_hash<KW>(nm, nt);
                            for (;;) {
                                h &= P.hash_mask;
                                if (atomicCAS(&slots[h], 0u, (uint32_t)i + 1u) == 0u) break;
                                ++h;
                            }
                        } else {
                            nsum[i] = -1;
                        }
                    }
                    const u64 dm = __ballot(valid && !keep);
                    if (valid && !keep) fre[nf + __popcll(dm & ((1ull << lane) - 1ull))] = i;
                    nf += __popcll(dm);
                }
                nfree = nf;
                __syncthreads();
                __threadfence();   // the inserts were L2 atomics: drop this CU's cached copies of the table
                ct[CT_GC]++; ct[CT_GC_SCANNED] += (uint32_t)nodes;
                collected = true;
            }
            uint32_t slot;
            const int ridx = tree_lookup<KW>(P, g, root_m, root_t, lane, &slot);
            int num = P.sims;                                           // player.py:140-143
            if (ridx >= 0) { const int rem = P.upper - rfli(nsum[ridx]); num = rem < num ? rem : num; }
            sims_left = num;
            phase = PH_SEARCH;
            // the collector is this launch's work: search resumes at the next tick
            if (collected) { status = AF_STATUS_YIELD; parked = true; ct[CT_YIELDS]++; break; }
        }

        if (sims_left <= 0) {
            // ------------- player.py:84-126 calc_policy -------------
            uint32_t slot;
            const int ridx = tree_lookup<KW>(P, g, root_m, root_t, lane, &slot);
            if (ridx < 0) { err = AF_ERR_NO_ROOT; break; }
            u64 legal[KW];
#pragma unroll
            for (int k = 0; k < KW; ++k) legal[k] = ~(root_m[k] | root_t[k]) & P.boardmask[k];
            const int L = bb_count<KW>(legal);
            int nv[KW];
            int most = -1;
#pragma unroll
            for (int k = 0; k < KW; ++k) {
                const bool lg = (legal[k] >> lane) & 1ull;
                nv[k] = lg ? (en[(size_t)ridx * CP + lane + 64 * k] & 0x7fffffff) : -1;
                most = nv[k] > most ? nv[k] : most;
            }
            most = wave_max_i32(most);
            u64 best[KW];
#pragma unroll
            for (int k = 0; k < KW; ++k) best[k] = __ballot(nv[k] == most && nv[k] >= 0);
            const int nb = bb_count<KW>(best);
            const uint32_t plyc = plyctr++;
            const int bi = (int)af_pick((uint32_t)nb, plyc, episode, AF_STREAM_BEST, k0, k1);   // :102
            const int best_cell = bb_select<KW>(best, bi);
            int action = best_cell, has_policy = 1;
            float pol[KW];
#pragma unroll
            for (int k = 0; k < KW; ++k) pol[k] = 0.0f;
            ct[CT_PLIES]++;
            if (lane == 0) atomicAdd(&P.progress[0], 1ull);
            if (!P.training && !random_a) {
                has_policy = 0;                                                                  // :106-107
            } else {
                tau = tau * (random_a ? P.tau_decay_r : P.tau_decay);                            // :108-111
                if (tau <= 0.01) {                                                               // :112-115
                    const float u = (float)(1.0 / (double)nb);
#pragma unroll
                    for (int k = 0; k < KW; ++k) pol[k] = ((best[k] >> lane) & 1ull) ? u : 0.0f;
                } else {
                    const float mx = (float)most;
                    const float inv_tau = (float)(1.0 / tau);
                    int e_idx[KW];
                    int base = 0;
#pragma unroll
                    for (int k = 0; k < KW; ++k) {
                        const bool lg = (legal[k] >> lane) & 1ull;
                        e_idx[k] = base + __popcll(legal[k] & ((1ull << lane) - 1ull));
                        base += __popcll(legal[k]);
                        if (lg) {
                            float x = (float)nv[k] / mx;                                         // :116
                            x = af_powf(x, inv_tau);                                             // :117
                            pol[k] = x;
                            s_pv[e_idx[k]] = x;
                        }
                    }
                    __syncthreads();
                    const float sum = pairwise_sum(s_pv, L);                                     // :118 np.sum
                    __syncthreads();
#pragma unroll
                    for (int k = 0; k < KW; ++k) {
                        if ((legal[k] >> lane) & 1ull) { pol[k] = pol[k] / sum; s_pv[e_idx[k]] = pol[k]; }
                    }
                    __syncthreads();
                    // :125 np.random.choice(L, p): fp64 cumsum, / last, searchsorted(side='right')
                    if (lane == 0) {
                        double acc = 0.0;
                        for (int e = 0; e < L; ++e) { acc = acc + (double)s_pv[e]; s_cdf[e] = acc; }
                    }
                    __syncthreads();
                    const double last = s_cdf[L - 1];
                    const double u = af_uniform(plyc, episode, AF_STREAM_MOVE, k0, k1);
                    int cnt = 0;
#pragma unroll
                    for (int k = 0; k < KW; ++k) {
                        const bool lg = (legal[k] >> lane) & 1ull;
                        const bool le = lg && (s_cdf[e_idx[k]] / last <= u);
                        cnt += __popcll(__ballot(le));
                    }
                    __syncthreads();
                    if (cnt >= L) cnt = L - 1;
                    action = bb_select<KW>(legal, cnt);
                }
            }
            // publish the move result
#pragma unroll
            for (int k = 0; k < KW; ++k) {
                P.policy[(size_t)g * CP + lane + 64 * k] = pol[k];
                P.visits[(size_t)g * CP + lane + 64 * k] = nv[k] < 0 ? 0 : nv[k];
            }
            if (lane == 0) { P.action[g] = action; P.has_policy[g] = has_policy; }

            if (P.mode == AF_MODE_EXTERNAL) {
                phase = PH_MOVE_DONE;
                status = AF_STATUS_MOVE_DONE;
                parked = true;
                break;
            }
            // ------------- player.py:53-82 run: record, step, game over? -------------
            if (ply >= P.max_ply) { err = AF_ERR_STATE; break; }
            const uint32_t seq = rfl32(P.ep_seq[g]);
            const size_t rslot = ((size_t)g * 2 + (seq & 1u)) * P.max_ply + ply;
            if (lane < 2 * KW) P.rec_key[rslot * 2 * KW + lane] = key_word<KW>(root_m, root_t, lane);
#pragma unroll
            for (int k = 0; k < KW; ++k) {
                const int c = lane + 64 * k;
                if (c < C) { P.rec_policy[rslot * C + c] = pol[k]; P.rec_visits[rslot * C + c] = nv[k] < 0 ? 0 : nv[k]; }
            }
            if (lane == 0) { P.rec_last[rslot] = root_last; P.rec_action[rslot] = action; }
            // utils.py:275 step: place my stone, flip perspective
            {
                u64 nm[KW], nt[KW];
#pragma unroll
                for (int k = 0; k < KW; ++k) { nm[k] = root_t[k]; nt[k] = root_m[k]; }
                bb_set<KW>(nt, action);
#pragma unroll
                for (int k = 0; k < KW; ++k) { root_m[k] = nm[k]; root_t[k] = nt[k]; }
            }
            root_last = action;
            ++ply;
            float fv;
            if (terminal_test<KW>(P, root_m, root_t, &fv)) {
                // episode finished: publish, then Player.reset() (:73) and start the next game
                if (seq - rfl32(P.ep_popped[g]) >= 2u) { err = AF_ERR_EP_OVERRUN; break; }
                if (lane == 0) {
                    P.ep_len[(size_t)g * 2 + (seq & 1u)] = ply;
                    P.ep_final[(size_t)g * 2 + (seq & 1u)] = fv;
                    P.ep_seq[g] = seq + 1u;
                }
                ct[CT_EPISODES]++;
                if (lane == 0) atomicAdd(&P.progress[1], 1ull);
                for (uint32_t s_ = lane; s_ <= P.hash_mask; s_ += 64) slots[s_] = 0;
                __syncthreads();
                nodes = 0; nfree = 0;
#pragma unroll
                for (int k = 0; k < KW; ++k) { root_m[k] = 0; root_t[k] = 0; }
                root_last = -1;
                ply = 0;
                tau = P.init_temp;
                episode += 1; sel = 0; plyctr = 0;
            }
            phase = PH_MOVE_START;
            continue;
        }

        TK_T(tk_sim);
        TK_ACC(2, tk_iter, tk_sim);
        // ------------- one simulation: player.py:204-228 MCTS_search -------------
        u64 cm[KW], ctb[KW];
        int last, depth;
        if (phase == PH_DESCENT) {       // resume the descent this game yielded in (hard cap below): position, depth and last move
#pragma unroll
            for (int k = 0; k < KW; ++k) {
                cm[k] = rfl64(P.leaf[(size_t)g * 2 * KW + k]);
                ctb[k] = rfl64(P.leaf[(size_t)g * 2 * KW + KW + k]);
            }
            last = rfli(P.leaf_last[g]); depth = rfli(P.depth[g]);
            phase = PH_SEARCH;
        } else {
#pragma unroll
            for (int k = 0; k < KW; ++k) { cm[k] = root_m[k]; ctb[k] = root_t[k]; }
            last = root_last; depth = 0;
        }
        // Child cache: an edge remembers the node its move leads to, so a descent follows indices instead of probing the hash and
        // comparing keys at every level (two dependent memory round trips per select).  A cached index stays valid for as long
        // as its parent lives: the collector only drops nodes whose stones are not a superset of the root's, a child has its
        // parent's stones plus one, so a surviving parent's children all survive; dropped slots are reused only for new nodes,
        // whose rows (cache included) are initialised at expansion; a whole-tree reset drops every node.
        int hint = 0, pidx = -1, pcl = 0;
        // ... and the child's rows are requested as soon as its index is known (right after the argmax), so that they travel
        // while the move is played and the terminal test runs
        typedef typename std::conditional<W64, double, float>::type wrow_t;
        int32_t r_raw[KW], r_c[KW];
        float r_p[KW];
        wrow_t r_w[KW];
        int32_t r_sum = 0;
        bool r_have = false;
        for (;;) {
            TK_T(tk_a);
            float tv;
            // The root of a simulation (depth 0).  Self-play: it was tested when the move that led to it was played (both sides, the
            // episode ends there) or is the empty board, so the per-simulation scan of it can only say "not terminal": skipped (r5).
            // EXTERNAL mode: a caller may hand over ANY position with no last move, also one in which the side to move already owns a
            // line — utils.py:199-235 returns (True, 1.0) there, so the scan is the two-sided one (the one-sided scan let
            // such a root be searched).  Inside a descent: through the last move only.
            bool term;
            if (depth == 0)
                term = P.mode == AF_MODE_EXTERNAL ? (bool)terminal_test<KW, false>(P, cm, ctb, &tv) : false;
            else
                term = (depth > 0 && P.goal <= 8) ? (bool)terminal_through_last<KW>(P, cm, ctb, last, inv_s16, lane, &tv)
                                                                       : (bool)terminal_test<KW, true>(P, cm, ctb, &tv);
            if (term) {                                                                                                   // :213-217
                backup(depth, tv, 0);
                ct[CT_TERMINALS]++; ct[CT_SIMS]++;
                sims_left--;
                break;
            }
            uint32_t slot = 0xffffffffu;
            int idx;
            if (hint) {
                idx = hint - 1;
            } else {
                idx = tree_lookup<KW>(P, g, cm, ctb, lane, &slot);
                if (idx >= 0 && pidx >= 0 && lane == 0) ec[(size_t)pidx * CP + pcl] = idx + 1;   // reached through another move order before
            }
            if (idx < 0) {                                                  // :218 unseen -> park for the net
                if constexpr (MEMO) {
                    // ... unless the memo holds this (position, last move): expand from it now, at the top of the outer loop
                    if (slot != 0xffffffffu && bb_count<KW>(cm) + bb_count<KW>(ctb) <= P.memo_max_stones) {
                        uint32_t unused;
                        const int me = memo_probe<KW>(P, memo_bucket<KW>(P, cm, ctb, last), cm, ctb, last, lane, &unused);
                        ++memo_probes;
                        if (me >= 0) {
                            ++memo_hits;
#pragma unroll
                            for (int k = 0; k < KW; ++k) { lm[k] = cm[k]; lt[k] = ctb[k]; }
                            c_depth = depth; c_slot = slot;
                            c_pol = P.memo_pv + (size_t)me * CP;
                            c_val = P.memo_val + me;
                            have_eval = true;
                            break;
                        }
                    }
                }
                if (lane < 2 * KW) P.leaf[(size_t)g * 2 * KW + lane] = key_word<KW>(cm, ctb, lane);
                if (lane == 0) { P.depth[g] = depth; P.leaf_last[g] = last; P.leaf_slot[g] = (int32_t)slot; }
                // utils.py:256 board_to_inputs -> float32[3][S][S]
                float* out = planes_out + (size_t)g * 3 * C;
#pragma unroll
                for (int k = 0; k < KW; ++k) {
                    const int c = lane + 64 * k;
                    if (c < C) {
                        out[c] = ((cm[k] >> lane) & 1ull) ? 1.0f : 0.0f;
                        out[C + c] = ((ctb[k] >> lane) & 1ull) ? 1.0f : 0.0f;
                        out[2 * C + c] = c == last ? 1.0f : 0.0f;
                    }
                }
                status = AF_STATUS_NEED_EVAL;
                parked = true;
                break;
            }
            // hard cap: a single simulation can descend 30 levels late in a game; past `budget_hard` selects in this launch the
            // game yields in front of its next select and resumes here at the next tick (path and position are in memory)
            if (work >= P.budget_hard) {
                if (lane < 2 * KW) P.leaf[(size_t)g * 2 * KW + lane] = key_word<KW>(cm, ctb, lane);
                if (lane == 0) { P.depth[g] = depth; P.leaf_last[g] = last; }
                phase = PH_DESCENT; status = AF_STATUS_YIELD; parked = true; ct[CT_YIELDS]++;
                break;
            }
            TK_T(tk_b);
            TK_ACC(3, tk_a, tk_b);
            // ------------- player.py:230-279 select_action_q_and_u -------------
            const bool is_root = depth == 0;
            u64 legal[KW];
#pragma unroll
            for (int k = 0; k < KW; ++k) legal[k] = ~(cm[k] | ctb[k]) & P.boardmask[k];
            if (!r_have) {
                r_sum = nsum[idx];
#pragma unroll
                for (int k = 0; k < KW; ++k) {
                    const size_t off = (size_t)idx * CP + lane + 64 * k;
                    r_raw[k] = en[off];
                    r_c[k] = ec[off];
                    if constexpr (W64) r_w[k] = ew64[off]; else r_w[k] = ew[off];
                    r_p[k] = ep[off];
                }
            }
            r_have = false;
            const int sum_n = rfli(r_sum) + 1;                              // :237
            if (lane == 0) nsum[idx] = sum_n;
            const uint32_t sel_id = sel++;
            int nn[KW];
            float pp[KW];
            wrow_t ww[KW];
            bool ff[KW];
            int32_t cc[KW];
#pragma unroll
            for (int k = 0; k < KW; ++k) {
                nn[k] = r_raw[k] & 0x7fffffff;
                ff[k] = r_raw[k] < 0;
                cc[k] = r_c[k];
                ww[k] = r_w[k];
                pp[k] = r_p[k];
            }
            double dd[KW];
            if (P.training) {
                // Dirichlet(alpha) noise, tier-B generator (af_noise.h); summation order = lane partials
                // then xor butterfly, identical to oracle afo_noise_philox_dirichlet
                // a lane draws its KW cells in ONE rejection loop (the wave leaves it when the slowest lane has
                // all its variates; the rounds of a cell are indexed by the counter, so the values are those
                // of af_gamma_lt1 cell by cell)
#ifdef AF_TICK_TIMING
                int tk_myit = 0;
#endif
                {
                    const float a_ = (float)P.alpha, inv_a = 1.0f / a_, one_m_a = 1.0f - a_;
                    uint32_t todo = 0;
#pragma unroll
                    for (int k = 0; k < KW; ++k) { dd[k] = 0.0; todo |= ((legal[k] >> lane) & 1ull) ? (1u << k) : 0u; }
                    // Round 0 of EVERY cell of the lane first, side by side (two independent dependency chains: the kernel's slow
                    // waves run alone on their SIMD and wait on instruction latency, not on issue slots), then the rejected cells
                    // (10 % of them) one at a time as before: the wave needs 1 double + ~2 single rounds instead of ~4 singles, at the
                    // same instruction count.  (Both cells side by side in every round costs twice the instructions in the clean-up
                    // rounds, where most lanes idle: +7 % per launch, profiles/r4_07.)  The rounds of a cell are indexed by its own
                    // counter, so every variate is the one af_gamma_lt1 returns.
                    af_u32x4 rk[KW];
                    uint32_t pend = 0;
#pragma unroll
                    for (int k = 0; k < KW; ++k) {
                        rk[k] = af_philox4x32(sel_id, episode, (AF_STREAM_GAMMA << 24) | (uint32_t)(lane + 64 * k), 0u, k0, k1);
                        float X;
                        const int ok = af_gamma_round(a_, inv_a, one_m_a, rk[k].v[0], rk[k].v[1], &X);
                        const bool lg = (todo >> k) & 1u;
                        dd[k] = (lg && ok) ? (double)X : 0.0;
                        pend |= (lg && !ok) ? (1u << k) : 0u;
                        if (KW > 2 && (k & 1)) __builtin_amdgcn_sched_barrier(0);      // (15x15: two chains at a time, not four)
                    }
#ifdef AF_TICK_TIMING
                    ++tk_myit;
#endif
                    todo = pend;
                    {
                    uint32_t it = 1;
                    af_u32x4 r = rk[0];
#pragma unroll
                    for (int q = 1; q < KW; ++q) if (!(todo & ((1u << q) - 1u))) r = rk[q];       // the first pending cell's block
                    while (todo) {
#ifdef AF_TICK_TIMING
                        ++tk_myit;
#endif
                        const int k = __builtin_ctz(todo);
                        if ((it & 1u) == 0u) r = af_philox4x32(sel_id, episode, (AF_STREAM_GAMMA << 24) | (uint32_t)(lane + 64 * k), it >> 1, k0, k1);
                        const uint32_t wu = (it & 1u) ? r.v[2] : r.v[0], wv = (it & 1u) ? r.v[3] : r.v[1];
                        float X;
                        const int ok = af_gamma_round(a_, inv_a, one_m_a, wu, wv, &X);
                        if (ok || it == 0xFFFFu) {
                            const double Xd = ok ? (double)X : 0.0;
#pragma unroll
                            for (int q = 0; q < KW; ++q) dd[q] = q == k ? Xd : dd[q];
                            todo &= todo - 1u;
                            it = 1;
#pragma unroll
                            for (int q = 1; q < KW; ++q) if (todo && !(todo & ((1u << q) - 1u))) r = rk[q];
                        } else {
                            ++it;
                        }
                    }
                    }
                }
#ifdef AF_TICK_TIMING
                tk[12] += (unsigned long long)wave_max_i32(tk_myit);       // the wave's iterations = its slowest lane's
#endif
                TK_T(tk_n1);
                TK_ACC(8, tk_b, tk_n1);
                double acc = 0.0;
#pragma unroll
                for (int k = 0; k < KW; ++k) acc = k == 0 ? dd[0] : acc + dd[k];
                acc = wave_sum_f64_tree(acc);
                if (acc == 0.0) {
                    // every variate underflowed (alpha small, few legal cells; ~2^-24 per select at alpha = 0.3 and L = 1):
                    // 1/acc would make the priors NaN and the candidate set empty.  Spec (include/af_noise.h, the oracle does
                    // the same): the draw degenerates to the uniform distribution over the legal cells.
                    int cnt = 0;
#pragma unroll
                    for (int k = 0; k < KW; ++k) cnt += __popcll(legal[k]);
                    const double u = 1.0 / (double)cnt;
#pragma unroll
                    for (int k = 0; k < KW; ++k) dd[k] = ((legal[k] >> lane) & 1ull) ? u : 0.0;
                } else {
                    const double inv = 1.0 / acc;
#pragma unroll
                    for (int k = 0; k < KW; ++k) dd[k] = dd[k] * inv;
                }
                TK_T(tk_n1e);
                TK_ACC(9, tk_n1, tk_n1e);
            }
            TK_T(tk_n2);
            const double sq = sqrt((double)(sum_n + 1));
            float sc[KW];
            float mx = -3.0e38f;
#pragma unroll
            for (int k = 0; k < KW; ++k) {
                const bool lg = (legal[k] >> lane) & 1ull;
                double q64;                                                 // SURVEY §8a rule 2
                if (nn[k] == 0) q64 = 0.0;
                else if (!W64 && ff[k]) q64 = (double)((float)ww[k] / (float)nn[k]);
                else q64 = (double)ww[k] / (double)nn[k];
                double t;
                if (P.training) {                                           // rule 3
                    double p_;
                    if (is_root) p_ = (double)(0.75f * pp[k]) + 0.25 * dd[k];
                    else p_ = (double)(0.9f * pp[k]) + 0.1 * dd[k];
                    t = ((P.c_puct * p_) * sq) / (double)(1 + nn[k]);
                } else {
                    t = ((double)(P.c_puct32 * pp[k]) * sq) / (double)(1 + nn[k]);
                }
                sc[k] = lg ? (float)(q64 + t) : -3.0e38f;
                mx = sc[k] > mx ? sc[k] : mx;
            }
            TK_T(tk_n3);
            TK_ACC(10, tk_n2, tk_n3);
            int cell = -1;
            if (is_root && P.training) {                                    // :264-276 forced root visits
                u64 cand[KW];
#pragma unroll
                for (int k = 0; k < KW; ++k) cand[k] = __ballot(((legal[k] >> lane) & 1ull) && nn[k] == 0);
                int m = bb_count<KW>(cand);
                if (m == 0) {
#pragma unroll
                    for (int k = 0; k < KW; ++k) cand[k] = __ballot(((legal[k] >> lane) & 1ull) && nn[k] == 1);
                    m = bb_count<KW>(cand);
                }
                if (m > 0) cell = bb_select_wave<KW>(cand, (int)af_pick((uint32_t)m, sel_id, episode, AF_STREAM_PICK, k0, k1), lane);
            }
            if (cell < 0) {                                                 // :277-279 argmax with uniform tie-break
                mx = wave_max_f32(mx);
                u64 cand[KW];
#pragma unroll
                for (int k = 0; k < KW; ++k) cand[k] = __ballot(((legal[k] >> lane) & 1ull) && sc[k] == mx);
                const int m = bb_count<KW>(cand);
                cell = bb_select_wave<KW>(cand, (int)af_pick((uint32_t)m, sel_id, episode, AF_STREAM_PICK, k0, k1), lane);
            }
            cell = rfli(cell);
            hint = 0;
#pragma unroll
            for (int k = 0; k < KW; ++k) { const int t_ = __builtin_amdgcn_readlane(cc[k], cell & 63); if ((cell >> 6) == k) hint = t_; }   // (cell is wave-uniform)
            hint = rfli(hint);
            pidx = idx; pcl = cell;
            if (hint) {
                const int ci = hint - 1;
                r_sum = nsum[ci];
#pragma unroll
                for (int k = 0; k < KW; ++k) {
                    const size_t off = (size_t)ci * CP + lane + 64 * k;
                    r_raw[k] = en[off];
                    r_c[k] = ec[off];
                    if constexpr (W64) r_w[k] = ew64[off]; else r_w[k] = ew[off];
                    r_p[k] = ep[off];
                }
                r_have = true;
            }
            ++work;
            ct[CT_SELECTS]++;
            ct[CT_LSUM] += (uint32_t)bb_count<KW>(legal);
            if (lane == 0) { pnode[depth] = idx; pcell[depth] = cell; }
            ++depth;
            // utils.py:275 step
            {
                u64 nm[KW], nt[KW];
#pragma unroll
                for (int k = 0; k < KW; ++k) { nm[k] = ctb[k]; nt[k] = cm[k]; }
                bb_set<KW>(nt, cell);
#pragma unroll
                for (int k = 0; k < KW; ++k) { cm[k] = nm[k]; ctb[k] = nt[k]; }
            }
            last = cell;
            __syncthreads();
            TK_T(tk_c);
            TK_ACC(4, tk_b, tk_c);
            TK_ACC(11, tk_n3, tk_c);
#ifdef AF_TICK_TIMING
            tk[7] += 1;
#endif
        }
    }
    if (!MEMO || !have_eval) break;
  }
    TK_T(tk_parked);
    if constexpr (MEMO) {
        if (lane == 0 && memo_probes) {
            atomicAdd(reinterpret_cast<unsigned long long*>(&P.memo_stats[MEMO_PROBES]), (unsigned long long)memo_probes);
            if (memo_hits) atomicAdd(reinterpret_cast<unsigned long long*>(&P.memo_stats[MEMO_HITS]), (unsigned long long)memo_hits);
        }
    }

    // ---- 3. store state ----
    if (err) { phase = PH_ERROR; status = err; }
    if (lane < 2 * KW) P.root[(size_t)g * 2 * KW + lane] = key_word<KW>(root_m, root_t, lane);
    if (lane == 0) {
        P.phase[g] = phase;
        P.pending[g] = status == AF_STATUS_NEED_EVAL ? 1 : 0;
        P.root_last[g] = root_last;
        P.sims_left[g] = sims_left;
        P.ply[g] = ply;
        P.nodes[g] = nodes;
        P.nfree[g] = nfree;
        P.tau[g] = tau;
        P.episode[g] = episode;
        P.sel[g] = sel;
        P.plyctr[g] = plyctr;
        P.status[g] = status;
        // (fire-and-forget atomics: a read-modify-write would cost this wave one more memory round trip at its very end)
        for (int i = 0; i < CT_N; ++i)
            if (ct[i]) atomicAdd(reinterpret_cast<unsigned long long*>(&P.counters[(size_t)g * CT_N + i]), (unsigned long long)ct[i]);
        // launch-shape evidence: selects per launch and the wave's lifetime (100 MHz constant clock)
        const u64 dt = wall_clock64() - t_start;
        atomicAdd(&P.hist[HIST_WORK + (work < 63 ? work : 63)], 1ull);
        const u64 tb = dt / 800ull;
        atomicAdd(&P.hist[HIST_TIME + (tb < 31ull ? tb : 31ull)], 1ull);
        atomicMax(&P.hist[HIST_MAXTIME], dt);
    }
#ifdef AF_TICK_TIMING
    {
        TK_T(tk_end);
        tk[5] = tk_end - tk_parked;
        tk[6] = tk_end - tk_entry;
        if (lane == 0 && g < 8192) for (int q = 0; q < 13; ++q) g_tick_cycles[g][q] = tk[q];
    }
#endif
}

// ----------------------------------------------------------------------------------------------
// finished-episode hand-off (main.py:94 q.put): two small kernels compact every finished, not yet popped
// episode into ONE int32 buffer (include/af_engine.h: af_engine_pack_episodes), in (game, sequence) order
// ----------------------------------------------------------------------------------------------
// One 1024-thread workgroup walks the games in chunks of 1024 with an inclusive scan of (episodes, plies); an
// episode is taken iff every earlier one was taken and it still fits max_eps / max_plies (the taken set is a prefix
// of the (game, sequence) order).  Writes the header, meta[], final[] and advances ep_popped.
__global__ __launch_bounds__(1024) void af_pack_scan(EngineParams P, int max_eps, int max_plies, int32_t* __restrict__ out) {
    __shared__ int s_e[1024], s_p[1024];
    __shared__ int s_cut;                                       // first game index that did not fit completely
    const int t = threadIdx.x;
    int base_e = 0, base_p = 0;
    int32_t* meta = out + 4;
    int32_t* fin = out + 4 + (size_t)max_eps * 4;
    if (t == 0) s_cut = 0x7fffffff;
    __syncthreads();
    for (int g0 = 0; g0 < P.G; g0 += 1024) {
        const int g = g0 + t;
        int ne = 0, T0 = 0, T1 = 0;
        uint32_t pop = 0;
        if (g < P.G) {
            pop = P.ep_popped[g];
            ne = (int)(P.ep_seq[g] - pop);                       // 0, 1 or 2 finished episodes wait
            if (ne >= 1) T0 = P.ep_len[(size_t)g * 2 + (pop & 1u)];
            if (ne >= 2) T1 = P.ep_len[(size_t)g * 2 + ((pop + 1u) & 1u)];
        }
        s_e[t] = ne; s_p[t] = T0 + T1;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {              // Hillis-Steele inclusive scan
            const int ve = t >= off ? s_e[t - off] : 0, vp = t >= off ? s_p[t - off] : 0;
            __syncthreads();
            s_e[t] += ve; s_p[t] += vp;
            __syncthreads();
        }
        const int e_excl = base_e + s_e[t] - ne, p_excl = base_p + s_p[t] - (T0 + T1);
        int fit = 0;                                             // episodes of this game that fit the caps
        if (ne >= 1 && e_excl + 1 <= max_eps && p_excl + T0 <= max_plies) {
            fit = 1;
            if (ne >= 2 && e_excl + 2 <= max_eps && p_excl + T0 + T1 <= max_plies) fit = 2;
        }
        if (g < P.G && fit < ne) atomicMin(&s_cut, g);
        __syncthreads();
        const int cut = s_cut;
        if (g < P.G && g <= cut && fit > 0) {                    // g == cut: its first episode may still fit
            for (int k = 0; k < fit; ++k) {
                const uint32_t seq = pop + (uint32_t)k;
                const int idx = e_excl + k;
                meta[idx * 4 + 0] = g; meta[idx * 4 + 1] = (int32_t)seq;
                meta[idx * 4 + 2] = k == 0 ? T0 : T1; meta[idx * 4 + 3] = p_excl + (k == 0 ? 0 : T0);
                fin[idx] = __float_as_int(P.ep_final[(size_t)g * 2 + (seq & 1u)]);
            }
            P.ep_popped[g] = pop + (uint32_t)fit;
        }
        __syncthreads();
        if (cut != 0x7fffffff) {
            // totals = everything before the cut game + what the cut game itself contributed
            if (g == cut) {
                out[0] = e_excl + fit;
                out[1] = p_excl + (fit >= 1 ? T0 : 0) + (fit >= 2 ? T1 : 0);
            }
            break;
        }
        base_e += s_e[1023]; base_p += s_p[1023];
        __syncthreads();
    }
    __syncthreads();
    if (t == 0) {
        if (s_cut == 0x7fffffff) { out[0] = base_e; out[1] = base_p; }
        out[2] = 2 * (P.C <= 128 ? 2 : 4); out[3] = P.C;
    }
}

// One workgroup per packed episode: copies its T ply records (contiguous in the game's record buffer) into the
// ply region.  Record r = [key 2*KW2 ints | policy C | visits C | last | action].
__global__ __launch_bounds__(256) void af_pack_copy(EngineParams P, int max_eps, int32_t* __restrict__ out) {
    const int idx = blockIdx.x;
    if (idx >= out[0]) return;
    const int32_t* meta = out + 4 + (size_t)idx * 4;
    const int g = meta[0], T = meta[2], poff = meta[3];
    const uint32_t seq = (uint32_t)meta[1];
    const int C = P.C, KW2 = 2 * (C <= 128 ? 2 : 4), R = 2 * KW2 + 2 * C + 2;
    const size_t r0 = ((size_t)g * 2 + (seq & 1u)) * P.max_ply;
    int32_t* dst = out + 4 + (size_t)max_eps * 5 + (size_t)poff * R;
    const int32_t* keys = reinterpret_cast<const int32_t*>(P.rec_key + r0 * KW2);
    const int32_t* pol = reinterpret_cast<const int32_t*>(P.rec_policy + r0 * C);
    const int32_t* vis = P.rec_visits + r0 * C;
    for (int i = threadIdx.x; i < T * R; i += blockDim.x) {
        const int ply = i / R, f = i - ply * R;
        int32_t v;
        if (f < 2 * KW2) v = keys[(size_t)ply * 2 * KW2 + f];
        else if (f < 2 * KW2 + C) v = pol[(size_t)ply * C + (f - 2 * KW2)];
        else if (f < 2 * KW2 + 2 * C) v = vis[(size_t)ply * C + (f - 2 * KW2 - C)];
        else if (f == 2 * KW2 + 2 * C) v = P.rec_last[r0 + ply];
        else v = P.rec_action[r0 + ply];
        dst[i] = v;
    }
}

// ----------------------------------------------------------------------------------------------
// EXTERNAL mode, many games at once (arena: choose_best_player.py:38-60): set the roots / fetch the move results of n
// games with one upload, one launch and one download instead of a device synchronisation and ~10 tiny copies per game
// ----------------------------------------------------------------------------------------------
// req[i] = { game, last_cell, random_a, reset_tree }, keys[i][2KW]
__global__ __launch_bounds__(64) void af_set_roots_kernel(EngineParams P, int n, const int32_t* __restrict__ req, const u64* __restrict__ keys, int KW2) {
    const int i = blockIdx.x, lane = threadIdx.x;
    if (i >= n) return;
    const int g = req[4 * i], last = req[4 * i + 1], ra = req[4 * i + 2], reset = req[4 * i + 3];
    if (reset) {                                                 // Player.reset(): player.py:48-51
        uint32_t* slots = P.hash + (size_t)g * (P.hash_mask + 1);
        for (uint32_t s_ = lane; s_ <= P.hash_mask; s_ += 64) slots[s_] = 0;
        if (lane == 0) { P.nodes[g] = 0; P.nfree[g] = 0; P.tau[g] = P.init_temp; P.episode[g] += 1; P.sel[g] = 0; P.plyctr[g] = 0; }
    }
    if (lane < KW2) P.root[(size_t)g * KW2 + lane] = keys[(size_t)i * KW2 + lane];
    if (lane == 0) { P.root_last[g] = last; P.random_a[g] = ra ? 1 : 0; P.phase[g] = PH_MOVE_START; P.pending[g] = 0; P.status[g] = 0; }
}

// out[i] = { action, has_policy, status/phase error (0 ok), tau bits lo, tau bits hi }, then policy[C] and visits[C]
__global__ __launch_bounds__(64) void af_move_results_kernel(EngineParams P, int n, const int32_t* __restrict__ games, int32_t* __restrict__ out, int CP) {
    const int i = blockIdx.x, lane = threadIdx.x, C = P.C, R = 8 + 2 * C;
    if (i >= n) return;
    const int g = games[i];
    int32_t* o = out + (size_t)i * R;
    if (lane == 0) {
        const int ph = P.phase[g];
        o[0] = P.action[g]; o[1] = P.has_policy[g];
        o[2] = ph == PH_MOVE_DONE ? 0 : (ph == PH_ERROR ? P.status[g] : AF_ERR_STATE);
        const u64 tb = (u64)__double_as_longlong(P.tau[g]);
        o[3] = (int32_t)(uint32_t)tb; o[4] = (int32_t)(uint32_t)(tb >> 32);
    }
    for (int c = lane; c < C; c += 64) {
        o[8 + c] = __float_as_int(P.policy[(size_t)g * CP + c]);
        o[8 + C + c] = P.visits[(size_t)g * CP + c];
    }
}

// ----------------------------------------------------------------------------------------------
// host side
// ----------------------------------------------------------------------------------------------
// Evaluation memo, the writing side: one wave per game, launched after the forward that evaluated this tick's parked leaves.  Reads
// nothing the tick kernel writes concurrently (same stream, a launch of its own) and the tick kernel never writes the memo, so
// neither side can see a torn entry.  Inside this launch an entry has one writer: the first wave whose atomicMax of the launch
// serial on the entry's lock word returns an older serial; the others drop their insert (the position will come by again).
template <int KW>
__global__ __launch_bounds__(64) void af_memo_insert_kernel(EngineParams P, const float* __restrict__ policy, const float* __restrict__ value) {
    constexpr int CP = 64 * KW, MK = 4 * KW;
    const int g = blockIdx.x, lane = threadIdx.x;
    if (!rfli(P.pending[g])) return;
    u64 lm[KW], lt[KW];
#pragma unroll
    for (int k = 0; k < KW; ++k) {
        lm[k] = rfl64(P.leaf[(size_t)g * 2 * KW + k]);
        lt[k] = rfl64(P.leaf[(size_t)g * 2 * KW + KW + k]);
    }
    if (bb_count<KW>(lm) + bb_count<KW>(lt) > P.memo_max_stones) return;
    const int last = rfli(P.leaf_last[g]);
    const uint32_t bucket = memo_bucket<KW>(P, lm, lt, last);
    uint32_t empty_ways;
    if (memo_probe<KW>(P, bucket, lm, lt, last, lane, &empty_ways) >= 0) return;            // another game brought it earlier
    const u64 serial = rfl64(P.memo_stats[MEMO_SERIAL]);
    // way: an entry never written, else one chosen by hash and launch (every entry of a full bucket is replaced in turn)
    const int way = empty_ways ? __builtin_ctz(empty_ways) : (int)((key_hash<KW>(lm, lt) >> 24) + (uint32_t)serial) & 3;
    const size_t e = (size_t)bucket * 4 + way;
    int own = 0;
    if (lane == 0) own = atomicMax(reinterpret_cast<unsigned long long*>(&P.memo_lock[e]), (unsigned long long)serial) < serial ? 1 : 0;
    if (!rfli(own)) return;
    if (lane < MK) P.memo_key[e * MK + lane] = memo_word<KW>(P, lm, lt, last, lane);
#pragma unroll
    for (int k = 0; k < KW; ++k) {
        const int c = lane + 64 * k;
        P.memo_pv[e * CP + c] = c < P.C ? policy[(size_t)g * P.C + c] : 0.0f;
    }
    if (lane == 0) {
        P.memo_val[e] = value[g];
        atomicAdd(reinterpret_cast<unsigned long long*>(&P.memo_stats[MEMO_INSERTS]), 1ull);
        if (!empty_ways) atomicAdd(reinterpret_cast<unsigned long long*>(&P.memo_stats[MEMO_REPLACED]), 1ull);
    }
}

struct af_engine {
    EngineParams P;
    int device;
    int KW;
    std::vector<void*> allocs;
    std::vector<int32_t> h_i32, h_i32b;
    int32_t* pack_dev = nullptr;      // staging buffer of af_engine_pop_episodes
    char* stage_dev = nullptr;        // staging buffer of af_engine_set_roots / af_engine_move_results
    size_t stage_cap = 0;
    int64_t pack_cap = 0;
    std::vector<int32_t> pack_host;
    std::vector<u64> h_ct;
    unsigned long long* stamps = nullptr;   // af_engine_stamp (allocated on first use; freed with the other allocations)
    bool memo = false;                // af_engine_memo_enable
    bool memo_budget_fixed = false;   // AF_MEMO_BUDGET given: af_engine_set_tick_budget leaves memo_budget alone
    size_t memo_entries = 0;
};

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "[af_engine] %s failed: %s\n", #x, hipGetErrorString(e_)); return AF_ERR_HIP; } } while (0)

template <typename T>
static int dalloc(af_engine* e, T** p, size_t n) {
    void* q = nullptr;
    if (hipMalloc(&q, n * sizeof(T)) != hipSuccess) {
        // an allocation that does not fit is the caller's sizing error, not a broken device: report it and take the failure out of
        // the runtime's sticky last-error slot (the next HIP user of the process — torch — would otherwise trip over it)
        fprintf(stderr, "[af_engine] hipMalloc of %zu bytes failed (node_cap / num_games too large for the device?)\n", n * sizeof(T));
        (void)hipGetLastError();
        return AF_ERR_HIP;
    }
    HIP_OK(hipMemset(q, 0, n * sizeof(T)));
    e->allocs.push_back(q);
    *p = (T*)q;
    return AF_OK;
}

extern "C" {

int af_abi_version(void) { return AF_ABI_VERSION; }

const char* af_strerror(int code) {
    switch (code) {
        case AF_OK: return "ok";
        case AF_ERR_ARG: return "bad argument";
        case AF_ERR_HIP: return "HIP runtime error";
        case AF_ERR_NODE_CAP: return "transposition store full (raise node_cap)";
        case AF_ERR_NO_ROOT: return "get_action on a finished position";
        case AF_ERR_EP_OVERRUN: return "finished episodes not popped in time";
        case AF_ERR_STATE: return "engine state error";
        default: return "unknown error";
    }
}

int af_engine_create(const af_config* cfg, int32_t num_games, int32_t device, int32_t mode, int32_t training,
                     uint64_t seed, uint32_t first_game_id, int32_t node_cap, af_engine** out) {
    if (!cfg || !out || num_games < 1) return AF_ERR_ARG;
    const int S = cfg->board_size, C = S * S;
    if (S < 2 || C > 256 || cfg->goal < 2 || cfg->goal > S || cfg->simulation_per_step < 1) return AF_ERR_ARG;
    const int value_f64 = (mode & AF_MODE_VALUE_F64) ? 1 : 0;
    mode &= ~AF_MODE_VALUE_F64;
    if (mode != AF_MODE_SELFPLAY && mode != AF_MODE_EXTERNAL) return AF_ERR_ARG;
    // the noise sampler is numpy's legacy gamma for shape < 1 only (include/af_noise.h); the reference uses 0.3
    if (!(cfg->dirichlet_alpha > 0.0 && cfg->dirichlet_alpha < 1.0)) return AF_ERR_ARG;
    if (cfg->upper_simulation_per_step < 1 || !(cfg->c_puct >= 0.0)) return AF_ERR_ARG;
    HIP_OK(hipSetDevice(device));
    af_engine* e = new af_engine();
    e->device = device;
    e->KW = C <= 128 ? 2 : 4;
    const int KW = e->KW, CP = 64 * KW;
    EngineParams& P = e->P;
    memset(&P, 0, sizeof(P));
    P.G = num_games; P.S = S; P.C = C; P.goal = cfg->goal;
    P.sims = cfg->simulation_per_step; P.upper = cfg->upper_simulation_per_step;
    P.training = training ? 1 : 0; P.mode = mode; P.max_ply = C; P.w64 = value_f64;
    P.budget = AF_DEFAULT_TICK_BUDGET; P.budget_hard = AF_DEFAULT_TICK_BUDGET_HARD;
    if (const char* b = getenv("AF_TICK_BUDGET")) { const int v = atoi(b); if (v > 0) P.budget = v; }
    if (const char* b = getenv("AF_TICK_BUDGET_HARD")) { const int v = atoi(b); if (v > 0) P.budget_hard = v; }
    if (P.budget_hard < P.budget) P.budget_hard = P.budget;
    if (node_cap <= 0) node_cap = mode == AF_MODE_SELFPLAY ? 4 * P.sims + 64 : 32768;
    if (node_cap < P.sims + 8) node_cap = P.sims + 8;
    P.node_cap = node_cap;
    uint32_t hcap = 64;
    while (hcap < 2u * (uint32_t)node_cap) hcap <<= 1;
    P.hash_mask = hcap - 1;
    P.init_temp = cfg->init_temp; P.tau_decay = cfg->tau_decay_rate; P.tau_decay_r = cfg->tau_decay_rate_r;
    P.alpha = cfg->dirichlet_alpha; P.c_puct = cfg->c_puct; P.c_puct32 = (float)cfg->c_puct;
    P.k0 = af_key0(seed); P.seed_hi = (uint32_t)(seed >> 32); P.first_game_id = first_game_id;
    for (int c = 0; c < C; ++c) {
        P.boardmask[c >> 6] |= 1ull << (c & 63);
        if (c % S <= S - cfg->goal) P.colmask[c >> 6] |= 1ull << (c & 63);
    }
    const size_t G = num_games;
    int rc = AF_OK;
#define A(p, n) if (rc == AF_OK) rc = dalloc(e, &P.p, (n))
    A(phase, G); A(pending, G); A(sims_left, G); A(ply, G); A(root_last, G); A(nodes, G); A(nfree, G); A(depth, G);
    A(leaf_last, G); A(leaf_slot, G); A(status, G); A(random_a, G); A(action, G); A(has_policy, G);
    A(visits, G * CP); A(policy, G * CP); A(root, G * 2 * KW); A(leaf, G * 2 * KW); A(tau, G);
    A(episode, G); A(sel, G); A(plyctr, G); A(path_node, G * CP); A(path_cell, G * CP);
    A(hash, G * hcap); A(node_key, G * node_cap * 2 * KW); A(node_sum, G * node_cap); A(free_idx, G * node_cap);
    A(edge_n, G * node_cap * CP); A(edge_p, G * node_cap * CP); A(edge_c, G * node_cap * CP);
    if (value_f64) { A(edge_w64, G * node_cap * CP); A(edge_w, 64); } else { A(edge_w, G * node_cap * CP); A(edge_w64, 8); }
    A(ep_seq, G); A(ep_popped, G); A(counters, G * CT_N); A(progress, 2); A(hist, HIST_N);
    if (mode == AF_MODE_SELFPLAY) {
        const size_t R = G * 2 * P.max_ply;
        A(ep_len, G * 2); A(ep_final, G * 2); A(rec_key, R * 2 * KW); A(rec_policy, R * C); A(rec_visits, R * C);
        A(rec_last, R); A(rec_action, R);
    }
#undef A
    if (rc != AF_OK) { af_engine_destroy(e); return rc; }
    // initial state
    std::vector<int32_t> ph(G, mode == AF_MODE_SELFPLAY ? PH_MOVE_START : PH_IDLE), rl(G, -1);
    std::vector<double> tau(G, cfg->init_temp);
    HIP_OK(hipMemcpy(P.phase, ph.data(), G * 4, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(P.root_last, rl.data(), G * 4, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(P.tau, tau.data(), G * 8, hipMemcpyHostToDevice));
    e->h_i32.resize(G); e->h_i32b.resize(G); e->h_ct.resize(G * CT_N);
    *out = e;
    return AF_OK;
}

void af_engine_destroy(af_engine* e) {
    if (!e) return;
    (void)hipSetDevice(e->device);
    for (void* p : e->allocs) (void)hipFree(p);
    if (e->pack_dev) (void)hipFree(e->pack_dev);
    if (e->stage_dev) (void)hipFree(e->stage_dev);
    delete e;
}

int32_t af_engine_num_games(const af_engine* e) { return e->P.G; }
int32_t af_engine_cells(const af_engine* e) { return e->P.C; }
int32_t af_engine_key_words(const af_engine* e) { return 2 * e->KW; }
int32_t af_engine_max_plies(const af_engine* e) { return e->P.max_ply; }

int af_engine_tick(af_engine* e, void* stream, const float* policy_dev, const float* value_dev, float* planes_dev) {
    if (!e || !planes_dev) return AF_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
#define AF_TICK(KW_, W_, M_) hipLaunchKernelGGL((af_tick_kernel<KW_, W_, M_>), dim3(e->P.G), dim3(64), 0, st, e->P, policy_dev, value_dev, planes_dev)
    if (e->memo) {
        if (e->KW == 2) { if (e->P.w64) AF_TICK(2, true, true); else AF_TICK(2, false, true); }
        else { if (e->P.w64) AF_TICK(4, true, true); else AF_TICK(4, false, true); }
    } else {
        if (e->KW == 2) { if (e->P.w64) AF_TICK(2, true, false); else AF_TICK(2, false, false); }
        else { if (e->P.w64) AF_TICK(4, true, false); else AF_TICK(4, false, false); }
    }
#undef AF_TICK
    HIP_OK(hipGetLastError());
    return AF_OK;
}

int af_engine_memo_enable(af_engine* e, int32_t log2_buckets, int32_t max_stones) {
    if (!e || e->memo || log2_buckets < 4 || log2_buckets > 28 || max_stones < 0) return AF_ERR_ARG;
    HIP_OK(hipSetDevice(e->device));
    EngineParams& P = e->P;
    const size_t NE = (size_t)4 << log2_buckets, CP = 64 * (size_t)e->KW, MK = 4 * (size_t)e->KW;
    int rc = AF_OK;
    if (rc == AF_OK) rc = dalloc(e, &P.memo_key, NE * MK);
    if (rc == AF_OK) rc = dalloc(e, &P.memo_pv, NE * CP);
    if (rc == AF_OK) rc = dalloc(e, &P.memo_val, NE);
    if (rc == AF_OK) rc = dalloc(e, &P.memo_lock, NE);
    if (rc == AF_OK) rc = dalloc(e, &P.memo_stats, (size_t)MEMO_N);
    if (rc != AF_OK) return rc;
    P.memo_bucket_mask = (uint32_t)((1ull << log2_buckets) - 1ull);
    P.memo_max_stones = max_stones;
    P.memo_epoch = 1;
    P.memo_budget = P.budget < 6 ? P.budget : 6;      // profiles/r5_31: 5 and 8 give the same moves/s (fewer yields vs a shorter launch)
    if (const char* b = getenv("AF_MEMO_BUDGET")) { const int v = atoi(b); if (v > 0) { P.memo_budget = v; e->memo_budget_fixed = true; } }
    e->memo_entries = NE;
    e->memo = true;
    return AF_OK;
}

int af_engine_memo_insert(af_engine* e, void* stream, const float* policy_dev, const float* value_dev) {
    if (!e || !e->memo || !policy_dev || !value_dev) return AF_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (e->KW == 2) hipLaunchKernelGGL((af_memo_insert_kernel<2>), dim3(e->P.G), dim3(64), 0, st, e->P, policy_dev, value_dev);
    else hipLaunchKernelGGL((af_memo_insert_kernel<4>), dim3(e->P.G), dim3(64), 0, st, e->P, policy_dev, value_dev);
    HIP_OK(hipGetLastError());
    return AF_OK;
}

int af_engine_memo_clear(af_engine* e, void* stream) {
    if (!e || !e->memo) return AF_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    // O(1): launches enqueued from now on carry the next epoch, under which no stored entry matches and every way counts as empty
    // (EngineParams travel by value, so this is stream-ordered like the memset it replaces; a captured graph has the epoch baked
    // in — af_engine_memo_epoch is part of what its owner keys it on).  Lock words and the serial keep counting.  Only when the
    // 32-bit epoch would wrap are the keys really zeroed.
    if (e->P.memo_epoch == 0xffffffffu) {
        HIP_OK(hipMemsetAsync(e->P.memo_key, 0, e->memo_entries * 4 * (size_t)e->KW * 8, st));
        e->P.memo_epoch = 1;
    } else {
        ++e->P.memo_epoch;
    }
    return AF_OK;
}

int64_t af_engine_memo_epoch(af_engine* e) { return (e && e->memo) ? (int64_t)e->P.memo_epoch : 0; }

int af_engine_memo_stats(af_engine* e, void* stream, uint64_t* out) {
    if (!e || !e->memo || !out) return AF_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    HIP_OK(hipMemcpyAsync(out, e->P.memo_stats, (size_t)MEMO_N * 8, hipMemcpyDeviceToHost, st));
    HIP_OK(hipStreamSynchronize(st));
    out[MEMO_N] = (uint64_t)e->memo_entries;
    return AF_OK;
}

int af_engine_status(af_engine* e, void* stream, int32_t* status_host) {
    if (!e || !status_host) return AF_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    HIP_OK(hipMemcpyAsync(status_host, e->P.status, (size_t)e->P.G * 4, hipMemcpyDeviceToHost, st));
    HIP_OK(hipStreamSynchronize(st));
    for (int g = 0; g < e->P.G; ++g) if (status_host[g] < 0) return status_host[g];
    return AF_OK;
}

int af_engine_set_training(af_engine* e, int32_t training) {
    if (!e) return AF_ERR_ARG;
    e->P.training = training ? 1 : 0;
    return AF_OK;
}

int af_engine_set_simulations(af_engine* e, int32_t sims, int32_t upper) {
    if (!e || sims < 1 || upper < 1 || sims + 8 > e->P.node_cap) return AF_ERR_ARG;
    e->P.sims = sims;
    e->P.upper = upper;
    return AF_OK;
}

int af_engine_set_root(af_engine* e, int32_t game, const uint64_t* key, int32_t last_cell, int32_t random_a,
                       int32_t reset_tree) {
    if (!e || !key || game < 0 || game >= e->P.G || e->P.mode != AF_MODE_EXTERNAL) return AF_ERR_ARG;
    EngineParams& P = e->P;
    const int KW = e->KW;
    for (int k = 0; k < KW; ++k) {                 // a position: stones on the board only, no cell owned by both colours
        if ((key[k] | key[KW + k]) & ~P.boardmask[k]) return AF_ERR_ARG;
        if (key[k] & key[KW + k]) return AF_ERR_ARG;
    }
    if (last_cell < -1 || last_cell >= P.C) return AF_ERR_ARG;
    HIP_OK(hipDeviceSynchronize());
    if (reset_tree) {   // Player.reset(): player.py:48-51
        const size_t hcap = (size_t)P.hash_mask + 1;
        HIP_OK(hipMemset(P.hash + (size_t)game * hcap, 0, hcap * 4));
        int32_t zero = 0;
        HIP_OK(hipMemcpy(P.nodes + game, &zero, 4, hipMemcpyHostToDevice));
        HIP_OK(hipMemcpy(P.nfree + game, &zero, 4, hipMemcpyHostToDevice));
        double tau = P.init_temp;
        HIP_OK(hipMemcpy(P.tau + game, &tau, 8, hipMemcpyHostToDevice));
        uint32_t ep;
        HIP_OK(hipMemcpy(&ep, P.episode + game, 4, hipMemcpyDeviceToHost));
        ep += 1;
        HIP_OK(hipMemcpy(P.episode + game, &ep, 4, hipMemcpyHostToDevice));
        uint32_t z = 0;
        HIP_OK(hipMemcpy(P.sel + game, &z, 4, hipMemcpyHostToDevice));
        HIP_OK(hipMemcpy(P.plyctr + game, &z, 4, hipMemcpyHostToDevice));
    }
    HIP_OK(hipMemcpy(P.root + (size_t)game * 2 * KW, key, (size_t)2 * KW * 8, hipMemcpyHostToDevice));
    int32_t v = last_cell;
    HIP_OK(hipMemcpy(P.root_last + game, &v, 4, hipMemcpyHostToDevice));
    v = random_a ? 1 : 0;
    HIP_OK(hipMemcpy(P.random_a + game, &v, 4, hipMemcpyHostToDevice));
    v = PH_MOVE_START;
    HIP_OK(hipMemcpy(P.phase + game, &v, 4, hipMemcpyHostToDevice));
    v = 0;
    HIP_OK(hipMemcpy(P.pending + game, &v, 4, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(P.status + game, &v, 4, hipMemcpyHostToDevice));
    return AF_OK;
}

int af_engine_set_roots(af_engine* e, void* stream, int32_t n, const int32_t* games, const uint64_t* keys, const int32_t* last_cells,
                        const int32_t* random_a, const int32_t* reset_tree) {
    if (!e || e->P.mode != AF_MODE_EXTERNAL || n < 0 || (n > 0 && (!games || !keys || !last_cells))) return AF_ERR_ARG;
    if (n == 0) return AF_OK;
    EngineParams& P = e->P;
    const int KW = e->KW, KW2 = 2 * KW;
    std::vector<int32_t> req((size_t)4 * n);
    for (int i = 0; i < n; ++i) {
        if (games[i] < 0 || games[i] >= P.G || last_cells[i] < -1 || last_cells[i] >= P.C) return AF_ERR_ARG;
        const uint64_t* k = keys + (size_t)i * KW2;
        for (int q = 0; q < KW; ++q)
            if (((k[q] | k[KW + q]) & ~P.boardmask[q]) || (k[q] & k[KW + q])) return AF_ERR_ARG;
        req[4 * i] = games[i]; req[4 * i + 1] = last_cells[i]; req[4 * i + 2] = random_a ? random_a[i] : 0; req[4 * i + 3] = reset_tree ? reset_tree[i] : 0;
    }
    hipStream_t st = (hipStream_t)stream;
    const size_t rb = (size_t)4 * n * 4, kb = (size_t)n * KW2 * 8, need = rb + kb;
    if (e->stage_cap < need) {
        if (e->stage_dev) (void)hipFree(e->stage_dev);
        e->stage_dev = nullptr; e->stage_cap = 0;
        HIP_OK(hipMalloc((void**)&e->stage_dev, need * 2));
        e->stage_cap = need * 2;
    }
    HIP_OK(hipMemcpyAsync(e->stage_dev, req.data(), rb, hipMemcpyHostToDevice, st));
    HIP_OK(hipMemcpyAsync(e->stage_dev + rb, keys, kb, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(af_set_roots_kernel, dim3(n), dim3(64), 0, st, P, n, (const int32_t*)e->stage_dev, (const u64*)(e->stage_dev + rb), KW2);
    HIP_OK(hipGetLastError());
    HIP_OK(hipStreamSynchronize(st));             // req / keys are caller memory: the upload has to be over before we return
    return AF_OK;
}

int af_engine_move_results(af_engine* e, void* stream, int32_t n, const int32_t* games, int32_t* action_cells, int32_t* has_policy,
                           float* policies, int32_t* visits, double* taus) {
    if (!e || n < 0 || (n > 0 && (!games || !action_cells))) return AF_ERR_ARG;
    if (n == 0) return AF_OK;
    EngineParams& P = e->P;
    for (int i = 0; i < n; ++i) if (games[i] < 0 || games[i] >= P.G) return AF_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    const int C = P.C, R = 8 + 2 * C, CP = 64 * e->KW;
    const size_t gb = (size_t)n * 4, ob = (size_t)n * R * 4, need = gb + ob;
    if (e->stage_cap < need) {
        if (e->stage_dev) (void)hipFree(e->stage_dev);
        e->stage_dev = nullptr; e->stage_cap = 0;
        HIP_OK(hipMalloc((void**)&e->stage_dev, need * 2));
        e->stage_cap = need * 2;
    }
    HIP_OK(hipMemcpyAsync(e->stage_dev, games, gb, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(af_move_results_kernel, dim3(n), dim3(64), 0, st, P, n, (const int32_t*)e->stage_dev, (int32_t*)(e->stage_dev + gb), CP);
    HIP_OK(hipGetLastError());
    std::vector<int32_t> out((size_t)n * R);
    HIP_OK(hipMemcpyAsync(out.data(), e->stage_dev + gb, ob, hipMemcpyDeviceToHost, st));
    HIP_OK(hipStreamSynchronize(st));
    for (int i = 0; i < n; ++i) {
        const int32_t* o = out.data() + (size_t)i * R;
        if (o[2] != 0) return o[2];
        action_cells[i] = o[0];
        if (has_policy) has_policy[i] = o[1];
        if (policies) memcpy(policies + (size_t)i * C, o + 8, (size_t)C * 4);
        if (visits) memcpy(visits + (size_t)i * C, o + 8 + C, (size_t)C * 4);
        if (taus) { const uint64_t tb = (uint64_t)(uint32_t)o[3] | ((uint64_t)(uint32_t)o[4] << 32); memcpy(taus + i, &tb, 8); }
    }
    return AF_OK;
}

int af_engine_move_result(af_engine* e, int32_t game, int32_t* action_cell, int32_t* has_policy, float* policy,
                          int32_t* visits, double* tau) {
    if (!e || game < 0 || game >= e->P.G) return AF_ERR_ARG;
    EngineParams& P = e->P;
    const int CP = 64 * e->KW;
    HIP_OK(hipDeviceSynchronize());
    int32_t ph;
    HIP_OK(hipMemcpy(&ph, P.phase + game, 4, hipMemcpyDeviceToHost));
    if (ph == PH_ERROR) { int32_t st; HIP_OK(hipMemcpy(&st, P.status + game, 4, hipMemcpyDeviceToHost)); return st; }
    if (ph != PH_MOVE_DONE) return AF_ERR_STATE;
    if (action_cell) HIP_OK(hipMemcpy(action_cell, P.action + game, 4, hipMemcpyDeviceToHost));
    if (has_policy) HIP_OK(hipMemcpy(has_policy, P.has_policy + game, 4, hipMemcpyDeviceToHost));
    if (policy) HIP_OK(hipMemcpy(policy, P.policy + (size_t)game * CP, (size_t)P.C * 4, hipMemcpyDeviceToHost));
    if (visits) HIP_OK(hipMemcpy(visits, P.visits + (size_t)game * CP, (size_t)P.C * 4, hipMemcpyDeviceToHost));
    if (tau) HIP_OK(hipMemcpy(tau, P.tau + game, 8, hipMemcpyDeviceToHost));
    return AF_OK;
}

int64_t af_engine_pack_ints(const af_engine* e, int32_t max_episodes, int32_t max_plies) {
    if (!e || max_episodes < 0 || max_plies < 0) return AF_ERR_ARG;
    const int64_t R = 4 * e->KW + 2 * e->P.C + 2;
    return 4 + (int64_t)max_episodes * 5 + (int64_t)max_plies * R;
}

int af_engine_pack_episodes(af_engine* e, void* stream, int32_t max_episodes, int32_t max_plies, int32_t* out_dev) {
    if (!e || e->P.mode != AF_MODE_SELFPLAY || max_episodes < 1 || max_plies < 1 || !out_dev) return AF_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(af_pack_scan, dim3(1), dim3(1024), 0, st, e->P, max_episodes, max_plies, out_dev);
    hipLaunchKernelGGL(af_pack_copy, dim3(max_episodes), dim3(256), 0, st, e->P, max_episodes, out_dev);
    HIP_OK(hipGetLastError());
    return AF_OK;
}

// Host-array form of the hand-off (kept for callers that want padded arrays): one pack on the device, ONE copy of
// the used prefix to the host, then a scatter on the host.
int af_engine_pop_episodes(af_engine* e, void* stream, int32_t cap, int32_t* meta, float* final_value,
                           uint64_t* keys, float* policies, int32_t* visits, int32_t* lasts, int32_t* actions) {
    if (!e || e->P.mode != AF_MODE_SELFPLAY || cap < 0) return AF_ERR_ARG;
    if (cap == 0) return 0;
    EngineParams& P = e->P;
    hipStream_t st = (hipStream_t)stream;
    const int C = P.C, KW2 = 2 * e->KW, MP = P.max_ply, R = 2 * KW2 + 2 * C + 2;
    const int max_plies = cap * MP > (1 << 18) ? (1 << 18) : cap * MP;           // bounds the staging buffer (<= 264 MB at 11x11)
    const int64_t ints = af_engine_pack_ints(e, cap, max_plies);
    if (e->pack_cap < ints) {
        if (e->pack_dev) (void)hipFree(e->pack_dev);
        e->pack_dev = nullptr; e->pack_cap = 0;
        HIP_OK(hipMalloc((void**)&e->pack_dev, (size_t)ints * 4));
        e->pack_cap = ints;
    }
    int rc = af_engine_pack_episodes(e, stream, cap, max_plies, e->pack_dev);
    if (rc) return rc;
    const size_t head = 4 + (size_t)cap * 5;
    e->pack_host.resize(head);
    HIP_OK(hipMemcpyAsync(e->pack_host.data(), e->pack_dev, head * 4, hipMemcpyDeviceToHost, st));
    HIP_OK(hipStreamSynchronize(st));
    const int n = e->pack_host[0], np = e->pack_host[1];
    if (n <= 0) return 0;
    e->pack_host.resize(head + (size_t)np * R);
    HIP_OK(hipMemcpyAsync(e->pack_host.data() + head, e->pack_dev + head, (size_t)np * R * 4, hipMemcpyDeviceToHost, st));
    HIP_OK(hipStreamSynchronize(st));
    const int32_t* H = e->pack_host.data();
    for (int i = 0; i < n; ++i) {
        const int32_t* m = H + 4 + (size_t)i * 4;
        const int T = m[2], poff = m[3];
        meta[4 * i] = m[0]; meta[4 * i + 1] = m[1]; meta[4 * i + 2] = T; meta[4 * i + 3] = 0;
        memcpy(final_value + i, H + 4 + (size_t)cap * 4 + i, 4);
        for (int t = 0; t < T; ++t) {
            const int32_t* r = H + head + (size_t)(poff + t) * R;
            memcpy(keys + ((size_t)i * MP + t) * KW2, r, (size_t)KW2 * 8);
            memcpy(policies + ((size_t)i * MP + t) * C, r + 2 * KW2, (size_t)C * 4);
            memcpy(visits + ((size_t)i * MP + t) * C, r + 2 * KW2 + C, (size_t)C * 4);
            lasts[(size_t)i * MP + t] = r[2 * KW2 + 2 * C];
            actions[(size_t)i * MP + t] = r[2 * KW2 + 2 * C + 1];
        }
    }
    return n;
}

int af_engine_counters(af_engine* e, void* stream, uint64_t* out) {
    if (!e || !out) return AF_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    const int G = e->P.G;
    HIP_OK(hipMemcpyAsync(e->h_ct.data(), e->P.counters, (size_t)G * CT_N * 8, hipMemcpyDeviceToHost, st));
    HIP_OK(hipMemcpyAsync(e->h_i32.data(), e->P.nodes, (size_t)G * 4, hipMemcpyDeviceToHost, st));
    HIP_OK(hipMemcpyAsync(e->h_i32b.data(), e->P.nfree, (size_t)G * 4, hipMemcpyDeviceToHost, st));
    HIP_OK(hipStreamSynchronize(st));
    for (int i = 0; i < AF_NUM_COUNTERS; ++i) out[i] = 0;
    // out[0..7] as ABI v1, out[8] = live nodes, out[9..12] = collector runs, slots scanned, yields, stalls
    for (int g = 0; g < G; ++g) {
        for (int i = 0; i < 8; ++i) out[i] += e->h_ct[(size_t)g * CT_N + i];
        for (int i = 8; i < CT_N; ++i) out[i + 1] += e->h_ct[(size_t)g * CT_N + i];
        out[8] += (uint64_t)(e->h_i32[g] - e->h_i32b[g]);
    }
    return AF_OK;
}

int af_engine_tick_histogram(af_engine* e, void* stream, uint64_t* out, int32_t reset) {
    if (!e || !out) return AF_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    HIP_OK(hipMemcpyAsync(out, e->P.hist, (size_t)HIST_N * 8, hipMemcpyDeviceToHost, st));
    if (reset) HIP_OK(hipMemsetAsync(e->P.hist, 0, (size_t)HIST_N * 8, st));
    HIP_OK(hipStreamSynchronize(st));
    return AF_OK;
}

int af_engine_set_tick_budget(af_engine* e, int32_t selects_per_launch) {
    if (!e || selects_per_launch < 1) return AF_ERR_ARG;
    e->P.budget = selects_per_launch;
    if (e->P.budget_hard < selects_per_launch) e->P.budget_hard = selects_per_launch;
    if (e->memo && !e->memo_budget_fixed) e->P.memo_budget = selects_per_launch < 6 ? selects_per_launch : 6;   // stays min(budget, 6)
    return AF_OK;
}

int af_engine_progress(af_engine* e, void* stream, uint64_t* out) {
    if (!e || !out) return AF_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    HIP_OK(hipMemcpyAsync(out, e->P.progress, 16, hipMemcpyDeviceToHost, st));
    HIP_OK(hipStreamSynchronize(st));
    return AF_OK;
}

int af_engine_progress_async(af_engine* e, void* stream, uint64_t* out_pinned) {
    if (!e || !out_pinned) return AF_ERR_ARG;
    HIP_OK(hipMemcpyAsync(out_pinned, e->P.progress, 16, hipMemcpyDeviceToHost, (hipStream_t)stream));
    return AF_OK;
}

// ABI v6: device-clock stamps, so that a kernel's duration can be measured INSIDE a captured graph (ROCm refuses timing events there)
__global__ void af_stamp_kernel(unsigned long long* stamps, int slot) {
    if (threadIdx.x == 0) stamps[slot] = wall_clock64();       // s_memrealtime: the device-wide 100-MHz clock
}

int af_engine_stamp(af_engine* e, void* stream, int32_t slot) {
    if (!e || slot < 0 || slot >= AF_STAMP_SLOTS) return AF_ERR_ARG;
    if (!e->stamps) {
        HIP_OK(hipSetDevice(e->device));
        int rc = dalloc(e, &e->stamps, (size_t)AF_STAMP_SLOTS);
        if (rc != AF_OK) return rc;
    }
    hipLaunchKernelGGL(af_stamp_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, e->stamps, (int)slot);
    HIP_OK(hipGetLastError());
    return AF_OK;
}

int af_engine_stamps_async(af_engine* e, void* stream, uint64_t* out_pinned) {
    if (!e || !out_pinned || !e->stamps) return AF_ERR_ARG;
    HIP_OK(hipMemcpyAsync(out_pinned, e->stamps, AF_STAMP_SLOTS * 8, hipMemcpyDeviceToHost, (hipStream_t)stream));
    return AF_OK;
}

int af_engine_tree_dump(af_engine* e, int32_t game, int32_t cap, uint64_t* keys, int32_t* sum_n, int32_t* n,
                        float* w, float* p, uint8_t* f32) {
    if (!e || game < 0 || game >= e->P.G) return AF_ERR_ARG;
    EngineParams& P = e->P;
    const int KW2 = 2 * e->KW, CP = 64 * e->KW, C = P.C;
    HIP_OK(hipDeviceSynchronize());
    int32_t hw;
    HIP_OK(hipMemcpy(&hw, P.nodes + game, 4, hipMemcpyDeviceToHost));
    const size_t nb = (size_t)game * P.node_cap;
    std::vector<int32_t> bs((size_t)(hw > 0 ? hw : 1));
    if (hw > 0) HIP_OK(hipMemcpy(bs.data(), P.node_sum + nb, (size_t)hw * 4, hipMemcpyDeviceToHost));
    int live = 0;
    for (int i = 0; i < hw; ++i) live += bs[i] >= 0;
    if (cap < live || live == 0) return live;        // size query (or nothing to copy)
    std::vector<int32_t> bn((size_t)hw * CP);
    std::vector<float> bw((size_t)hw * CP), bp((size_t)hw * CP);
    std::vector<uint64_t> bk((size_t)hw * KW2);
    HIP_OK(hipMemcpy(bk.data(), P.node_key + nb * KW2, (size_t)hw * KW2 * 8, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(bn.data(), P.edge_n + nb * CP, (size_t)hw * CP * 4, hipMemcpyDeviceToHost));
    if (P.w64) {                                     // fp64 store: w is rounded here, af_engine_tree_w64 has the exact rows
        std::vector<double> bd((size_t)hw * CP);
        HIP_OK(hipMemcpy(bd.data(), P.edge_w64 + nb * CP, (size_t)hw * CP * 8, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < bd.size(); ++i) bw[i] = (float)bd[i];
    } else
    HIP_OK(hipMemcpy(bw.data(), P.edge_w + nb * CP, (size_t)hw * CP * 4, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(bp.data(), P.edge_p + nb * CP, (size_t)hw * CP * 4, hipMemcpyDeviceToHost));
    int o = 0;
    for (int i = 0; i < hw; ++i) {
        if (bs[i] < 0) continue;                     // slot freed by the collector
        for (int k = 0; k < KW2; ++k) keys[(size_t)o * KW2 + k] = bk[(size_t)i * KW2 + k];
        sum_n[o] = bs[i];
        for (int c = 0; c < C; ++c) {
            const int32_t raw = bn[(size_t)i * CP + c];
            n[(size_t)o * C + c] = raw & 0x7fffffff;
            f32[(size_t)o * C + c] = raw < 0 ? 1 : 0;
            w[(size_t)o * C + c] = bw[(size_t)i * CP + c];
            p[(size_t)o * C + c] = bp[(size_t)i * CP + c];
        }
        ++o;
    }
    return live;
}

// Player.reset(search_tree) (player.py:48-51): adopt a tree handed over by the caller — replaces game `game`'s store by the
// `count` nodes given in tree_dump's format (keys [count][2KW]; sum_n [count]; n, w, p [count][C] by cell; f32 [count][C] =
// "W is fp32-typed", SURVEY rule 2).  EXTERNAL mode, between moves.
int af_engine_load_tree(af_engine* e, int32_t game, int32_t count, const uint64_t* keys, const int32_t* sum_n, const int32_t* n,
                        const float* w, const float* p, const uint8_t* f32) {
    if (!e || game < 0 || game >= e->P.G || e->P.mode != AF_MODE_EXTERNAL || count < 0 || count > e->P.node_cap) return AF_ERR_ARG;
    if (count > 0 && (!keys || !sum_n || !n || !w || !p || !f32)) return AF_ERR_ARG;
    EngineParams& P = e->P;
    const int KW = e->KW, KW2 = 2 * KW, CP = 64 * KW, C = P.C;
    const size_t hcap = (size_t)P.hash_mask + 1, nb = (size_t)game * P.node_cap;
    std::vector<uint32_t> slots(hcap, 0u);
    std::vector<int32_t> bn((size_t)count * CP, 0);
    std::vector<float> bw((size_t)count * CP, 0.0f), bp((size_t)count * CP, 0.0f);
    for (int i = 0; i < count; ++i) {
        const uint64_t* k = keys + (size_t)i * KW2;
        for (int q = 0; q < KW; ++q)
            if (((k[q] | k[KW + q]) & ~P.boardmask[q]) || (k[q] & k[KW + q])) return AF_ERR_ARG;
        const u64* kk = reinterpret_cast<const u64*>(k);
        uint32_t h = KW == 2 ? key_hash<2>(kk, kk + KW) : key_hash<4>(kk, kk + KW);
        for (;;) {
            h &= P.hash_mask;
            if (slots[h] == 0u) { slots[h] = (uint32_t)i + 1u; break; }
            const uint64_t* o = keys + (size_t)(slots[h] - 1u) * KW2;
            if (memcmp(o, k, (size_t)KW2 * 8) == 0) return AF_ERR_ARG;          // the same position twice
            ++h;
        }
        for (int c = 0; c < C; ++c) {
            const int32_t nn = n[(size_t)i * C + c];
            if (nn < 0) return AF_ERR_ARG;
            bn[(size_t)i * CP + c] = nn | (f32[(size_t)i * C + c] ? (int32_t)0x80000000 : 0);
            bw[(size_t)i * CP + c] = w[(size_t)i * C + c];
            bp[(size_t)i * CP + c] = p[(size_t)i * C + c];
        }
    }
    HIP_OK(hipDeviceSynchronize());
    HIP_OK(hipMemcpy(P.hash + (size_t)game * hcap, slots.data(), hcap * 4, hipMemcpyHostToDevice));
    if (count > 0) {
        HIP_OK(hipMemcpy(P.node_key + nb * KW2, keys, (size_t)count * KW2 * 8, hipMemcpyHostToDevice));
        HIP_OK(hipMemcpy(P.node_sum + nb, sum_n, (size_t)count * 4, hipMemcpyHostToDevice));
        HIP_OK(hipMemcpy(P.edge_n + nb * CP, bn.data(), bn.size() * 4, hipMemcpyHostToDevice));
        if (P.w64) {
            std::vector<double> bd(bw.begin(), bw.end());
            HIP_OK(hipMemcpy(P.edge_w64 + nb * CP, bd.data(), bd.size() * 8, hipMemcpyHostToDevice));
        } else
        HIP_OK(hipMemcpy(P.edge_w + nb * CP, bw.data(), bw.size() * 4, hipMemcpyHostToDevice));
        HIP_OK(hipMemcpy(P.edge_p + nb * CP, bp.data(), bp.size() * 4, hipMemcpyHostToDevice));
        HIP_OK(hipMemset(P.edge_c + nb * CP, 0, (size_t)count * CP * 4));
    }
    int32_t zero = 0;
    HIP_OK(hipMemcpy(P.nodes + game, &count, 4, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(P.nfree + game, &zero, 4, hipMemcpyHostToDevice));
    return AF_OK;
}

// utils.py:178-196 state_to_board, straight into a key
int af_state_to_key(const char* state, int32_t S, uint64_t* key) {
    if (!state || !key || S < 1 || S * S > 256) return AF_ERR_ARG;
    const int KW = S * S <= 128 ? 2 : 4;
    for (int k = 0; k < 2 * KW; ++k) key[k] = 0;
    int i = 0, j = 0;
    for (const char* p = state; *p; ++p) {
        const char ch = *p;
        if (ch == '/') { ++i; j = 0; }
        else if (ch >= 'a' && ch <= 'z') j += ch - 'a';
        else if (ch == '3' || ch == '1') {
            if (i >= S || j >= S) return AF_ERR_ARG;
            const int c = i * S + j;
            key[(ch == '3' ? 0 : KW) + (c >> 6)] |= 1ull << (c & 63);
            ++j;
        } else return AF_ERR_ARG;
    }
    return AF_OK;
}

// AF_MODE_VALUE_F64: the exact W rows of the live nodes, in af_engine_tree_dump's order ([count][C] doubles)
int af_engine_tree_w64(af_engine* e, int32_t game, int32_t cap, double* w) {
    if (!e || game < 0 || game >= e->P.G || !e->P.w64) return AF_ERR_ARG;
    EngineParams& P = e->P;
    const int CP = 64 * e->KW, C = P.C;
    HIP_OK(hipDeviceSynchronize());
    int32_t hw;
    HIP_OK(hipMemcpy(&hw, P.nodes + game, 4, hipMemcpyDeviceToHost));
    const size_t nb = (size_t)game * P.node_cap;
    std::vector<int32_t> bs((size_t)(hw > 0 ? hw : 1));
    if (hw > 0) HIP_OK(hipMemcpy(bs.data(), P.node_sum + nb, (size_t)hw * 4, hipMemcpyDeviceToHost));
    int live = 0;
    for (int i = 0; i < hw; ++i) live += bs[i] >= 0;
    if (cap < live || live == 0 || !w) return live;
    std::vector<double> bd((size_t)hw * CP);
    HIP_OK(hipMemcpy(bd.data(), P.edge_w64 + nb * CP, (size_t)hw * CP * 8, hipMemcpyDeviceToHost));
    int o = 0;
    for (int i = 0; i < hw; ++i) {
        if (bs[i] < 0) continue;
        for (int c = 0; c < C; ++c) w[(size_t)o * C + c] = bd[(size_t)i * CP + c];
        ++o;
    }
    return live;
}

// ... and the way in: overwrite the W rows of the `count` nodes af_engine_load_tree has just stored
int af_engine_set_tree_w64(af_engine* e, int32_t game, int32_t count, const double* w) {
    if (!e || game < 0 || game >= e->P.G || !e->P.w64 || e->P.mode != AF_MODE_EXTERNAL || count < 0 || count > e->P.node_cap || (count > 0 && !w))
        return AF_ERR_ARG;
    EngineParams& P = e->P;
    const int CP = 64 * e->KW, C = P.C;
    HIP_OK(hipDeviceSynchronize());
    int32_t hw;
    HIP_OK(hipMemcpy(&hw, P.nodes + game, 4, hipMemcpyDeviceToHost));
    if (hw != count) return AF_ERR_STATE;
    if (count == 0) return AF_OK;
    std::vector<double> bd((size_t)count * CP, 0.0);
    for (int i = 0; i < count; ++i)
        for (int c = 0; c < C; ++c) bd[(size_t)i * CP + c] = w[(size_t)i * C + c];
    HIP_OK(hipMemcpy(P.edge_w64 + (size_t)game * P.node_cap * CP, bd.data(), bd.size() * 8, hipMemcpyHostToDevice));
    return AF_OK;
}

#ifdef AF_TICK_TIMING
int af_engine_debug_tick_cycles(unsigned long long* host) {
    HIP_OK(hipDeviceSynchronize());
    HIP_OK(hipMemcpyFromSymbol(host, HIP_SYMBOL(g_tick_cycles), sizeof(unsigned long long) * 8192 * 13));
    return AF_OK;
}
#endif

// utils.py:156-175 board_to_state from a key
int af_key_to_state(const uint64_t* key, int32_t S, char* out, int32_t cap) {
    if (!key || !out || S < 1 || S * S > 256) return AF_ERR_ARG;
    const int KW = S * S <= 128 ? 2 : 4;
    int n = 0;
    for (int i = 0; i < S; ++i) {
        int run = 0;
        for (int j = 0; j < S; ++j) {
            const int c = i * S + j;
            const bool m = (key[c >> 6] >> (c & 63)) & 1ull, t = (key[KW + (c >> 6)] >> (c & 63)) & 1ull;
            if (!m && !t) { ++run; continue; }
            if (run) { if (n + 1 >= cap) return AF_ERR_ARG; out[n++] = (char)('a' + run); run = 0; }
            if (n + 1 >= cap) return AF_ERR_ARG;
            out[n++] = m ? '3' : '1';
        }
        if (run) { if (n + 1 >= cap) return AF_ERR_ARG; out[n++] = (char)('a' + run); }
        if (n + 1 >= cap) return AF_ERR_ARG;
        out[n++] = '/';
    }
    out[n] = 0;
    return n;
}

}  // extern "C"
