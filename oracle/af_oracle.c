/*
 * af_oracle.c — CPU restatement of the alphaFive self-play hot path.
 * TEST INFRASTRUCTURE ONLY (see af_oracle.h).  Not linked into the product.
 *
 * Every function cites the reference lines (under /root/reference) it restates.
 * Arithmetic follows SURVEY.md §8a "dtype rules" (NumPy >= 2 / NEP 50 semantics).
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math -shared -fPIC (oracle/Makefile).
 */
#include "af_oracle.h"
#include "../include/af_noise.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------- */
/* MT19937 (Matsumoto & Nishimura) + the numpy-legacy / CPython draws on it  */
/* ------------------------------------------------------------------------- */
typedef struct { uint32_t mt[624]; int idx; } mt_t;

static void mt_init_genrand(mt_t* m, uint32_t s) {          /* np.random.seed(int) */
    m->mt[0] = s;
    for (int i = 1; i < 624; ++i)
        m->mt[i] = 1812433253u * (m->mt[i - 1] ^ (m->mt[i - 1] >> 30)) + (uint32_t)i;
    m->idx = 624;
}
static void mt_init_by_array(mt_t* m, const uint32_t* key, int len) {  /* random.seed(int) */
    mt_init_genrand(m, 19650218u);
    int i = 1, j = 0;
    int k = 624 > len ? 624 : len;
    for (; k; --k) {
        m->mt[i] = (m->mt[i] ^ ((m->mt[i - 1] ^ (m->mt[i - 1] >> 30)) * 1664525u)) + key[j] + (uint32_t)j;
        ++i; ++j;
        if (i >= 624) { m->mt[0] = m->mt[623]; i = 1; }
        if (j >= len) j = 0;
    }
    for (k = 623; k; --k) {
        m->mt[i] = (m->mt[i] ^ ((m->mt[i - 1] ^ (m->mt[i - 1] >> 30)) * 1566083941u)) - (uint32_t)i;
        ++i;
        if (i >= 624) { m->mt[0] = m->mt[623]; i = 1; }
    }
    m->mt[0] = 0x80000000u;
    m->idx = 624;
}
static uint32_t mt_u32(mt_t* m) {
    if (m->idx >= 624) {
        uint32_t* mt = m->mt;
        int kk;
        for (kk = 0; kk < 624 - 397; ++kk) {
            uint32_t y = (mt[kk] & 0x80000000u) | (mt[kk + 1] & 0x7fffffffu);
            mt[kk] = mt[kk + 397] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        for (; kk < 623; ++kk) {
            uint32_t y = (mt[kk] & 0x80000000u) | (mt[kk + 1] & 0x7fffffffu);
            mt[kk] = mt[kk + (397 - 624)] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        uint32_t y = (mt[623] & 0x80000000u) | (mt[0] & 0x7fffffffu);
        mt[623] = mt[396] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        m->idx = 0;
    }
    uint32_t y = m->mt[m->idx++];
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}
static double mt_double(mt_t* m) {                          /* legacy_double */
    uint32_t a = mt_u32(m) >> 5, b = mt_u32(m) >> 6;
    return ((double)a * 67108864.0 + (double)b) / 9007199254740992.0;
}
/* numpy legacy_standard_gamma, shape < 1 branch — libm log/pow */
static double mt_gamma_lt1(mt_t* m, double shape) {
    for (;;) {
        double U = mt_double(m);
        double V = -log(1.0 - mt_double(m));
        if (U <= 1.0 - shape) {
            double X = pow(U, 1.0 / shape);
            if (X <= V) return X;
        } else {
            double Y = -log((1.0 - U) / shape);
            double X = pow(1.0 - shape + shape * Y, 1.0 / shape);
            if (X <= V + Y) return X;
        }
    }
}
/* np.random.dirichlet(alpha*ones(L)) — player.py:240 */
static void mt_dirichlet(mt_t* m, double alpha, int L, double* out) {
    double acc = 0.0;
    for (int j = 0; j < L; ++j) { out[j] = mt_gamma_lt1(m, alpha); acc = acc + out[j]; }
    double inv = 1.0 / acc;
    for (int j = 0; j < L; ++j) out[j] = out[j] * inv;
}
/* np.random.choice(seq of m) index draw — masked rejection, none when m == 1 */
static int mt_np_randbelow(mt_t* m, int n) {
    if (n <= 1) return 0;
    uint32_t rng = (uint32_t)(n - 1), mask = rng;
    mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
    uint32_t v;
    do { v = mt_u32(m) & mask; } while (v > rng);
    return (int)v;
}
/* CPython random.choice(seq of n): _randbelow_with_getrandbits — draws even for n == 1 */
static int mt_py_randbelow(mt_t* m, int n) {
    int k = 0;
    for (int t = n; t; t >>= 1) ++k;
    uint32_t r;
    do { r = mt_u32(m) >> (32 - k); } while ((int)r >= n);
    return (int)r;
}

/* ------------------------------------------------------------------------- */
/* rules & codecs — utils.py                                                 */
/* ------------------------------------------------------------------------- */

/* utils.py:156-175 board_to_state */
int afo_board_to_state(const int8_t* board, int S, char* out, int cap) {
    int n = 0;
    for (int i = 0; i < S; ++i) {
        int c = 0;
        for (int j = 0; j < S; ++j) {
            int8_t b = board[i * S + j];
            if (b == 0) { ++c; continue; }
            if (c > 0) { if (n + 1 >= cap) return -1; out[n++] = (char)('a' + c); }
            if (n + 1 >= cap) return -1;
            out[n++] = (char)('0' + (b + 2));
            c = 0;
        }
        if (c > 0) { if (n + 1 >= cap) return -1; out[n++] = (char)('a' + c); }
        if (n + 1 >= cap) return -1;
        out[n++] = '/';
    }
    out[n] = 0;
    return n;
}

/* utils.py:178-196 state_to_board */
int afo_state_to_board(const char* state, int S, int8_t* board) {
    memset(board, 0, (size_t)S * S);
    int i = 0, j = 0;
    for (const char* p = state; *p; ++p) {
        char ch = *p;
        if (ch == '/') { ++i; j = 0; }
        else if ((ch >= 'a' && ch <= 'z') || (ch >= 'A' && ch <= 'Z')) j += ch - 'a';
        else {
            if (i >= S || j >= S) return -1;
            board[i * S + j] = (int8_t)((ch - '0') - 2);
            ++j;
        }
    }
    return 0;
}

/* utils.py:199-235 is_game_over — same scan order, truncated edge windows */
int afo_is_game_over(const int8_t* board, int S, int goal, double* value) {
    const int h = S, w = S;
    for (int i = 0; i < h; ++i) {
        for (int j = 0; j < w; ++j) {
            int hang = 0;
            for (int k = i; k < (i + goal < w ? i + goal : w); ++k) hang += board[k * S + j];
            if (hang == goal) { *value = 1.0; return 1; }
            if (hang == -goal) { *value = -1.0; return 1; }
            int lie = 0;
            for (int k = j; k < (j + goal < h ? j + goal : h); ++k) lie += board[i * S + k];
            if (lie == goal) { *value = 1.0; return 1; }
            if (lie == -goal) { *value = -1.0; return 1; }
            if (i <= h - goal && j <= w - goal) {
                int xie = 0;
                for (int k = 0; k < goal; ++k) xie += board[(i + k) * S + (j + k)];
                if (xie == goal) { *value = 1.0; return 1; }
                if (xie == -goal) { *value = -1.0; return 1; }
            }
            if (i >= goal - 1 && j <= w - goal) {
                int xie = 0;
                for (int k = 0; k < goal; ++k) xie += board[(i - k) * S + (j + k)];
                if (xie == goal) { *value = 1.0; return 1; }
                if (xie == -goal) { *value = -1.0; return 1; }
            }
        }
    }
    int empties = 0;
    for (int c = 0; c < S * S; ++c) empties += (board[c] == 0);
    *value = 0.0;
    return empties == 0 ? 1 : 0;
}

/* utils.py:238-245 get_legal_actions — row-major empties */
int afo_legal_actions(const int8_t* board, int S, int* cells) {
    int L = 0;
    for (int c = 0; c < S * S; ++c) if (board[c] == 0) cells[L++] = c;
    return L;
}

/* utils.py:256-272 board_to_inputs */
void afo_board_to_inputs(const int8_t* board, int S, int last_cell, float* planes) {
    const int C = S * S;
    for (int c = 0; c < C; ++c) {
        planes[c] = board[c] == 1 ? 1.0f : 0.0f;
        planes[C + c] = board[c] == -1 ? 1.0f : 0.0f;
        planes[2 * C + c] = 0.0f;
    }
    if (last_cell >= 0) planes[2 * C + last_cell] = 1.0f;
}

/* utils.py:275-283 step */
void afo_step(int8_t* board, int S, int cell) {
    board[cell] = 1;
    for (int c = 0; c < S * S; ++c) board[c] = (int8_t)(-board[c]);
}

/* numpy float32 add.reduce: pairwise summation (8 accumulators, block 128) */
float afo_pairwise_sum_f32(const float* a, int n) {
    if (n < 8) {
        float res = 0.0f;
        for (int i = 0; i < n; ++i) res = res + a[i];
        return res;
    } else if (n <= 128) {
        float r[8];
        for (int k = 0; k < 8; ++k) r[k] = a[k];
        int i;
        for (i = 8; i < n - (n % 8); i += 8)
            for (int k = 0; k < 8; ++k) r[k] = r[k] + a[i + k];
        float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i) res = res + a[i];
        return res;
    } else {
        int n2 = n / 2;
        n2 -= n2 % 8;
        return afo_pairwise_sum_f32(a, n2) + afo_pairwise_sum_f32(a + n2, n - n2);
    }
}

/* utils.py:286-296 construct_weights (fp32, gamma weak-cast to fp32) */
void afo_construct_weights(int T, double gamma, float* w) {
    const float g = (float)gamma;
    w[T - 1] = 1.0f;
    for (int i = T - 2; i >= 0; --i) w[i] = w[i + 1] * g;
    float s = afo_pairwise_sum_f32(w, T);
    for (int i = 0; i < T; ++i) w[i] = ((float)T * w[i]) / s;
}

/* ------------------------------------------------------------------------- */
/* pseudo-net: integer hash of the input planes -> exactly representable fp32 */
/* ------------------------------------------------------------------------- */
static uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x45d9f3bu; x ^= x >> 16; x *= 0x45d9f3bu; x ^= x >> 16;
    return x;
}
void afo_pseudonet(const float* planes, int C, uint32_t salt, uint32_t peak, float* policy, float* value) {
    uint32_t h = salt;
    for (int c = 0; c < C; ++c) {
        if (planes[c] != 0.0f) h += mix32((uint32_t)c + 1u);
        if (planes[C + c] != 0.0f) h += mix32((uint32_t)c + 1001u);
        if (planes[2 * C + c] != 0.0f) h += mix32((uint32_t)c + 2001u);
    }
    for (int c = 0; c < C; ++c) {
        uint32_t m = mix32(h ^ mix32((uint32_t)c + 3001u));
        uint32_t k = 1u + (m & 0x3ffu) + ((((m >> 10) & 7u) == 0u) ? peak : 0u);
        policy[c] = (float)k * (1.0f / 131072.0f);
    }
    uint32_t mv = mix32(h ^ 0x9e3779b9u);
    *value = (float)((int)(mv & 0xffffu) - 32768) * (1.0f / 65536.0f);
}

/* ------------------------------------------------------------------------- */
/* transposition store — player.py:9-20,29 (exact-key map)                    */
/* ------------------------------------------------------------------------- */
typedef struct { uint64_t w[4]; } bb_t;

typedef struct afo_node {
    bb_t mine, theirs;
    int sum_n;
    int32_t* n;      /* [C] by cell */
    float* w;        /* running sum, fp32 (rule 2) */
    double* w64;     /* pipe path (networkAPI.py:72 float(v)): running sum of python floats */
    float* p;        /* prior */
    uint8_t* f32;    /* 1 once an fp32 (net) value has been added (rule 2 exception) */
    struct afo_node* next;
} afo_node;

#define AFO_BUCKETS (1u << 16)

struct afo_player {
    afo_config cfg;
    int S, C, training, rng_mode;
    int value_f64;   /* 1: leaf values are python floats (pipe path): w, q fp64 throughout (SURVEY 8a rule 2) */
    uint64_t seed; uint32_t game_id, k0, k1;
    mt_t np_rng, py_rng;
    uint32_t episode, sel_ctr, ply_ctr;
    afo_node** buckets;
    afo_node** order; int n_nodes, cap_nodes;
    double tau;
    int pv_kind; afo_pv_fn fn; void* user; uint32_t salt, peak;
    uint64_t st_sims, st_selects, st_expands, st_terminals, st_plies;
    uint64_t st_tie_select, st_tie_forced, st_tie_best;    /* picks among m > 1 candidates: the only places where the RNG decides */
};

static void board_to_bb(const int8_t* board, int C, bb_t* mine, bb_t* theirs) {
    memset(mine, 0, sizeof(*mine)); memset(theirs, 0, sizeof(*theirs));
    for (int c = 0; c < C; ++c) {
        if (board[c] == 1) mine->w[c >> 6] |= 1ull << (c & 63);
        else if (board[c] == -1) theirs->w[c >> 6] |= 1ull << (c & 63);
    }
}
static uint32_t bb_hash(const bb_t* a, const bb_t* b) {
    uint64_t h = 0x9E3779B97F4A7C15ull;
    for (int i = 0; i < 4; ++i) {
        h ^= a->w[i] + 0x9E3779B97F4A7C15ull + (h << 6) + (h >> 2);
        h ^= b->w[i] * 0xC2B2AE3D27D4EB4Full + (h << 6) + (h >> 2);
    }
    h ^= h >> 33; h *= 0xff51afd7ed558ccdull; h ^= h >> 33;
    return (uint32_t)h & (AFO_BUCKETS - 1);
}
static afo_node* tree_find(const afo_player* P, const bb_t* m, const bb_t* t) {
    for (afo_node* nd = P->buckets[bb_hash(m, t)]; nd; nd = nd->next)
        if (!memcmp(&nd->mine, m, sizeof(bb_t)) && !memcmp(&nd->theirs, t, sizeof(bb_t))) return nd;
    return NULL;
}
static afo_node* tree_insert(afo_player* P, const bb_t* m, const bb_t* t) {
    afo_node* nd = (afo_node*)calloc(1, sizeof(afo_node));
    nd->mine = *m; nd->theirs = *t;
    nd->n = (int32_t*)calloc((size_t)P->C, sizeof(int32_t));
    nd->w = (float*)calloc((size_t)P->C, sizeof(float));
    nd->w64 = (double*)calloc((size_t)P->C, sizeof(double));
    nd->p = (float*)calloc((size_t)P->C, sizeof(float));
    nd->f32 = (uint8_t*)calloc((size_t)P->C, 1);
    uint32_t h = bb_hash(m, t);
    nd->next = P->buckets[h]; P->buckets[h] = nd;
    if (P->n_nodes == P->cap_nodes) {
        P->cap_nodes = P->cap_nodes ? P->cap_nodes * 2 : 1024;
        P->order = (afo_node**)realloc(P->order, sizeof(afo_node*) * (size_t)P->cap_nodes);
    }
    P->order[P->n_nodes++] = nd;
    return nd;
}
static void tree_clear(afo_player* P) {
    for (int i = 0; i < P->n_nodes; ++i) {
        afo_node* nd = P->order[i];
        free(nd->n); free(nd->w); free(nd->w64); free(nd->p); free(nd->f32); free(nd);
    }
    P->n_nodes = 0;
    memset(P->buckets, 0, sizeof(afo_node*) * AFO_BUCKETS);
}

/* ------------------------------------------------------------------------- */
/* player                                                                     */
/* ------------------------------------------------------------------------- */
afo_player* afo_create(const afo_config* cfg, int training, int rng_mode, uint64_t seed,
                       uint32_t game_id, int pv_kind, afo_pv_fn fn, void* user,
                       uint32_t pseudo_salt, uint32_t pseudo_peak) {
    if (cfg->board_size < 1 || cfg->board_size * cfg->board_size > AFO_MAXC) return NULL;
    afo_player* P = (afo_player*)calloc(1, sizeof(afo_player));
    P->cfg = *cfg; P->S = cfg->board_size; P->C = P->S * P->S;
    P->training = training; P->rng_mode = rng_mode; P->seed = seed; P->game_id = game_id;
    P->k0 = af_key0(seed); P->k1 = af_key1(seed, game_id);
    mt_init_genrand(&P->np_rng, (uint32_t)seed);                /* np.random.seed(seed) */
    {                                                             /* random.seed(seed)    */
        uint32_t key[2] = { (uint32_t)seed, (uint32_t)(seed >> 32) };
        mt_init_by_array(&P->py_rng, key, key[1] ? 2 : 1);
    }
    P->buckets = (afo_node**)calloc(AFO_BUCKETS, sizeof(afo_node*));
    P->tau = cfg->init_temp;                                     /* player.py:32 */
    P->pv_kind = pv_kind; P->fn = fn; P->user = user; P->salt = pseudo_salt; P->peak = pseudo_peak;
    return P;
}
void afo_destroy(afo_player* P) {
    if (!P) return;
    tree_clear(P); free(P->order); free(P->buckets); free(P);
}
/* player.py:48-51 reset */
void afo_reset(afo_player* P) {
    tree_clear(P);
    P->tau = P->cfg.init_temp;
    P->episode += 1; P->sel_ctr = 0; P->ply_ctr = 0;
}
void afo_set_training(afo_player* P, int training) { P->training = training; }
/* player.py:140-143 reads config.simulation_per_step / upper_simulation_per_step from the live config at every get_action */
void afo_set_simulations(afo_player* P, int sims, int upper) { P->cfg.sims = sims; P->cfg.upper_sims = upper; }
void afo_set_value_f64(afo_player* P, int on) { P->value_f64 = on ? 1 : 0; }
double afo_tau(const afo_player* P) { return P->tau; }

static void run_pv(afo_player* P, const float* planes, float* policy, float* value) {
    if (P->pv_kind == AFO_PV_PSEUDO) afo_pseudonet(planes, P->C, P->salt, P->peak, policy, value);
    else P->fn(planes, policy, value, P->user);
}

/* tier-B Dirichlet: per-cell gamma variates, fixed wave-shaped summation order
 * (lane l = cell & 63 accumulates cells l, l+64, ...; then a 6-step xor butterfly).
 * Build-defined; the HIP select kernel evaluates exactly this expression tree. */
void afo_noise_philox_dirichlet(double alpha, const uint64_t* legal_bb, int C, uint32_t sel, uint32_t episode,
                                uint32_t k0, uint32_t k1, double* d) {
    double g[AFO_MAXC];
    for (int c = 0; c < AFO_MAXC; ++c) {
        int legal = c < C && ((legal_bb[c >> 6] >> (c & 63)) & 1ull);
        g[c] = legal ? af_gamma_lt1(alpha, sel, episode, (uint32_t)c, k0, k1) : 0.0;
    }
    double s[64], t[64];
    const int K = (C + 63) / 64;
    for (int l = 0; l < 64; ++l) {
        s[l] = g[l];
        for (int k = 1; k < K; ++k) s[l] = s[l] + g[l + 64 * k];
    }
    for (int off = 32; off >= 1; off >>= 1) {
        for (int l = 0; l < 64; ++l) t[l] = s[l] + s[l ^ off];
        memcpy(s, t, sizeof(s));
    }
    if (s[0] == 0.0) {       /* every variate underflowed: uniform over the legal cells (include/af_noise.h; same in the kernel) */
        int cnt = 0;
        for (int c = 0; c < C; ++c) cnt += (int)((legal_bb[c >> 6] >> (c & 63)) & 1ull);
        const double u = 1.0 / (double)cnt;
        for (int c = 0; c < C; ++c) d[c] = ((legal_bb[c >> 6] >> (c & 63)) & 1ull) ? u : 0.0;
        return;
    }
    const double inv = 1.0 / s[0];
    for (int c = 0; c < C; ++c) d[c] = g[c] * inv;
}

/* player.py:230-279 select_action_q_and_u; returns the chosen cell */
static int select_edge(afo_player* P, afo_node* nd, const int8_t* board, int is_root) {
    const int C = P->C;
    nd->sum_n += 1;                                               /* :237 */
    int cells[AFO_MAXC];
    const int L = afo_legal_actions(board, P->S, cells);          /* == list(node.a.keys()) order */
    double dcell[AFO_MAXC];
    const uint32_t sel = P->sel_ctr++;
    if (P->rng_mode == AFO_RNG_MT) {
        double d[AFO_MAXC];
        mt_dirichlet(&P->np_rng, P->cfg.alpha, L, d);             /* :240, always drawn */
        for (int e = 0; e < L; ++e) dcell[cells[e]] = d[e];
    } else if (P->training) {
        uint64_t legal[4] = {0, 0, 0, 0};
        for (int e = 0; e < L; ++e) legal[cells[e] >> 6] |= 1ull << (cells[e] & 63);
        afo_noise_philox_dirichlet(P->cfg.alpha, legal, C, sel, P->episode, P->k0, P->k1, dcell);
    }
    float scores[AFO_MAXC];
    const double sq = sqrt((double)(nd->sum_n + 1));              /* np.sqrt(int) -> fp64 */
    const float c32 = (float)P->cfg.c_puct;
    for (int e = 0; e < L; ++e) {
        const int c = cells[e];
        const float p = nd->p[c];
        const int n = nd->n[c];
        double q64;                                               /* rule 2 */
        if (n == 0) q64 = 0.0;
        else if (P->value_f64) q64 = nd->w64[c] / (double)n;
        else if (nd->f32[c]) q64 = (double)(nd->w[c] / (float)n);
        else q64 = (double)nd->w[c] / (double)n;
        double t;
        if (P->training) {                                        /* :247-253, rule 3 */
            double p_;
            if (is_root) p_ = (double)(0.75f * p) + 0.25 * dcell[c];
            else p_ = (double)(0.9f * p) + 0.1 * dcell[c];
            t = ((P->cfg.c_puct * p_) * sq) / (double)(1 + n);
        } else {
            t = ((double)(c32 * p) * sq) / (double)(1 + n);
        }
        scores[e] = (float)(q64 + t);                             /* stored to float32 array :241,261 */
    }
    if (is_root && P->training) {                                 /* :264-276 forced root visits */
        int cand[AFO_MAXC], m = 0;
        for (int e = 0; e < L; ++e) if (nd->n[cells[e]] == 0) cand[m++] = e;
        if (m == 0) for (int e = 0; e < L; ++e) if (nd->n[cells[e]] == 1) cand[m++] = e;
        if (m > 0) {
            if (m > 1) P->st_tie_forced++;
            int r = P->rng_mode == AFO_RNG_MT ? mt_np_randbelow(&P->np_rng, m)
                                              : (int)af_pick((uint32_t)m, sel, P->episode, AF_STREAM_PICK, P->k0, P->k1);
            return cells[cand[r]];
        }
    }
    float mx = scores[0];
    for (int e = 1; e < L; ++e) if (scores[e] > mx) mx = scores[e];
    int cand[AFO_MAXC], m = 0;
    for (int e = 0; e < L; ++e) if (scores[e] == mx) cand[m++] = e;   /* :277-278 */
    if (m > 1) P->st_tie_select++;
    int r = P->rng_mode == AFO_RNG_MT ? mt_np_randbelow(&P->np_rng, m)
                                      : (int)af_pick((uint32_t)m, sel, P->episode, AF_STREAM_PICK, P->k0, P->k1);
    return cells[cand[r]];
}

/* player.py:186-202 evaluate_and_expand */
static float expand(afo_player* P, const int8_t* board, const bb_t* m, const bb_t* t, int last_cell) {
    float planes[3 * AFO_MAXC], policy[AFO_MAXC], value;
    afo_board_to_inputs(board, P->S, last_cell, planes);
    run_pv(P, planes, policy, &value);
    int cells[AFO_MAXC];
    const int L = afo_legal_actions(board, P->S, cells);
    float all_p = 0.0f;                                           /* rule 1: sequential fp32 */
    for (int e = 0; e < L; ++e) all_p = all_p + policy[cells[e]];
    if (!(all_p >= 1e-5)) all_p = (float)1e-5;                    /* max(all_p, 1e-5) in fp64 compare */
    afo_node* nd = tree_insert(P, m, t);
    for (int e = 0; e < L; ++e) nd->p[cells[e]] = policy[cells[e]] / all_p;
    P->st_expands++;
    return value;
}

/* player.py:166-184 update_tree */
static void backup(const afo_player* P, afo_node** path_nodes, const int* path_cells, int depth, float v, int is_f32) {
    for (int d = depth - 1; d >= 0; --d) {
        v = -v;
        afo_node* nd = path_nodes[d];
        const int c = path_cells[d];
        nd->n[c] += 1;
        if (P->value_f64) { nd->w64[c] = nd->w64[c] + (double)v; continue; }   /* w += float(v) */
        nd->w[c] = nd->w[c] + v;
        if (is_f32) nd->f32[c] = 1;
    }
}

/* player.py:204-228 MCTS_search — one simulation */
static void simulate(afo_player* P, const int8_t* root_board, int last_cell) {
    int8_t board[AFO_MAXC];
    memcpy(board, root_board, (size_t)P->C);
    afo_node* path_nodes[AFO_MAXC + 1]; int path_cells[AFO_MAXC + 1]; int depth = 0;
    P->st_sims++;
    for (;;) {
        double tv;
        if (afo_is_game_over(board, P->S, P->cfg.goal, &tv)) {    /* :213-217 terminal first */
            backup(P, path_nodes, path_cells, depth, (float)tv, 0);
            P->st_terminals++;
            return;
        }
        bb_t m, t; board_to_bb(board, P->C, &m, &t);
        afo_node* nd = tree_find(P, &m, &t);
        if (!nd) {                                                /* :218-222 */
            float v = expand(P, board, &m, &t, last_cell);
            backup(P, path_nodes, path_cells, depth, v, 1);
            return;
        }
        const int cell = select_edge(P, nd, board, depth == 0);   /* :223 */
        P->st_selects++;
        path_nodes[depth] = nd; path_cells[depth] = cell; ++depth;
        afo_step(board, P->S, cell);                              /* :225 */
        last_cell = cell;                                         /* :228 */
    }
}

/* player.py:84-126 calc_policy */
static int calc_policy(afo_player* P, afo_node* nd, const int8_t* board, int random_a,
                       float* policy_out, int* action_cell, int32_t* visits_out) {
    const int C = P->C;
    int cells[AFO_MAXC];
    const int L = afo_legal_actions(board, P->S, cells);
    if (L == 0) return -2;
    if (visits_out) { memset(visits_out, 0, sizeof(int32_t) * (size_t)C); for (int e = 0; e < L; ++e) visits_out[cells[e]] = nd->n[cells[e]]; }
    float pv[AFO_MAXC];
    int most = -1;
    for (int e = 0; e < L; ++e) { pv[e] = (float)nd->n[cells[e]]; if (nd->n[cells[e]] > most) most = nd->n[cells[e]]; }
    int best[AFO_MAXC], nb = 0;
    for (int e = 0; e < L; ++e) if (nd->n[cells[e]] == most) best[nb++] = e;
    if (nb > 1) P->st_tie_best++;
    const uint32_t ply = P->ply_ctr++;
    int bi = P->rng_mode == AFO_RNG_MT ? mt_py_randbelow(&P->py_rng, nb)            /* :102 always drawn */
                                       : (int)af_pick((uint32_t)nb, ply, P->episode, AF_STREAM_BEST, P->k0, P->k1);
    const int best_cell = cells[best[bi]];
    P->st_plies++;
    if (!P->training && !random_a) { *action_cell = best_cell; return 1; }          /* :106-107 */
    if (policy_out) memset(policy_out, 0, sizeof(float) * (size_t)C);
    P->tau *= random_a ? P->cfg.tau_decay_r : P->cfg.tau_decay;                     /* :108-111 */
    if (P->tau <= 0.01) {                                                           /* :112-115 */
        if (policy_out) for (int b = 0; b < nb; ++b) policy_out[cells[best[b]]] = (float)(1.0 / (double)nb);
        *action_cell = best_cell;
        return 0;
    }
    float mx = pv[0];
    for (int e = 1; e < L; ++e) if (pv[e] > mx) mx = pv[e];
    const float inv_tau = (float)(1.0 / P->tau);                                    /* weak python float -> fp32 */
    for (int e = 0; e < L; ++e) {
        pv[e] = pv[e] / mx;
        pv[e] = P->rng_mode == AFO_RNG_MT ? powf(pv[e], inv_tau) : af_powf(pv[e], inv_tau);
    }
    const float s = afo_pairwise_sum_f32(pv, L);
    for (int e = 0; e < L; ++e) pv[e] = pv[e] / s;
    if (policy_out) for (int e = 0; e < L; ++e) policy_out[cells[e]] = pv[e];
    /* :125 np.random.choice(L, p=pv): cdf in fp64, normalised, searchsorted(side='right') */
    double cdf[AFO_MAXC], acc = 0.0;
    for (int e = 0; e < L; ++e) { acc = acc + (double)pv[e]; cdf[e] = acc; }
    const double last = cdf[L - 1];
    for (int e = 0; e < L; ++e) cdf[e] = cdf[e] / last;
    const double u = P->rng_mode == AFO_RNG_MT ? mt_double(&P->np_rng)
                                               : af_uniform(ply, P->episode, AF_STREAM_MOVE, P->k0, P->k1);
    int idx = 0;
    while (idx < L && cdf[idx] <= u) ++idx;
    if (idx >= L) idx = L - 1;
    *action_cell = cells[idx];
    return 0;
}

/* player.py:128-147 get_action */
int afo_get_action(afo_player* P, const char* state, int last_cell, int random_a,
                   float* policy_out, int* action_cell, int32_t* visits_out) {
    int8_t board[AFO_MAXC];
    if (afo_state_to_board(state, P->S, board)) return -1;
    bb_t m, t; board_to_bb(board, P->C, &m, &t);
    afo_node* nd = tree_find(P, &m, &t);
    int num = nd ? (P->cfg.sims < P->cfg.upper_sims - nd->sum_n ? P->cfg.sims : P->cfg.upper_sims - nd->sum_n)
                 : P->cfg.sims;                                   /* :140-143 */
    for (int i = 0; i < num; ++i) simulate(P, board, last_cell);
    nd = tree_find(P, &m, &t);
    if (!nd) return -3;                                           /* terminal root: reference raises */
    return calc_policy(P, nd, board, random_a, policy_out, action_cell, visits_out);
}

/* player.py:53-82 run */
int afo_run(afo_player* P, int max_T, char* states, float* policies, int* last_cells,
            int* action_cells, int32_t* visits, double* values, float* weights, double* final_value) {
    const int C = P->C;
    int8_t board[AFO_MAXC];
    memset(board, 0, sizeof(board));
    char state[272];
    afo_board_to_state(board, P->S, state, sizeof(state));       /* == get_init_state :37-46 */
    int T = 0, last = -1, over = 0;
    double value = 0.0;
    while (!over) {
        if (T >= max_T) return -4;
        int action;
        int rc = afo_get_action(P, state, last, 0, policies + (size_t)T * C, &action, visits + (size_t)T * C);
        if (rc < 0) return rc;
        if (rc == 1) memset(policies + (size_t)T * C, 0, sizeof(float) * (size_t)C);
        memcpy(states + (size_t)T * 272, state, 272);
        last_cells[T] = last; action_cells[T] = action;
        ++T;
        afo_state_to_board(state, P->S, board);
        afo_step(board, P->S, action);
        afo_board_to_state(board, P->S, state, sizeof(state));
        over = afo_is_game_over(board, P->S, P->cfg.goal, &value);
        last = action;
    }
    afo_reset(P);                                                 /* :73 */
    *final_value = value;
    if (T % 2 == 1) value = -value;                               /* :75-76 */
    afo_construct_weights(T, P->cfg.gamma, weights);
    for (int i = 0; i < T; ++i) { values[i] = value; value = -value; }
    return T;
}

/* ---- inspection ---- */
int afo_tree_size(const afo_player* P) { return P->n_nodes; }

int afo_node_get(const afo_player* P, const char* state, int* sum_n, int32_t* n, float* w, float* pr, uint8_t* f32) {
    int8_t board[AFO_MAXC];
    if (afo_state_to_board(state, P->S, board)) return -1;
    bb_t m, t; board_to_bb(board, P->C, &m, &t);
    afo_node* nd = tree_find(P, &m, &t);
    if (!nd) return 0;
    *sum_n = nd->sum_n;
    memcpy(n, nd->n, sizeof(int32_t) * (size_t)P->C); memcpy(w, nd->w, sizeof(float) * (size_t)P->C);
    memcpy(pr, nd->p, sizeof(float) * (size_t)P->C); memcpy(f32, nd->f32, (size_t)P->C);
    return 1;
}

int afo_node_get_w64(const afo_player* P, const char* state, double* w) {
    int8_t board[AFO_MAXC];
    if (afo_state_to_board(state, P->S, board)) return -1;
    bb_t m, t; board_to_bb(board, P->C, &m, &t);
    afo_node* nd = tree_find(P, &m, &t);
    if (!nd) return 0;
    memcpy(w, nd->w64, sizeof(double) * (size_t)P->C);
    return 1;
}

int afo_tree_dump_w64(const afo_player* P, int cap, double* w) {
    int cnt = P->n_nodes < cap ? P->n_nodes : cap;
    for (int i = 0; i < cnt; ++i) memcpy(w + (size_t)i * P->C, P->order[i]->w64, sizeof(double) * (size_t)P->C);
    return P->n_nodes;
}

int afo_tree_dump(const afo_player* P, int cap, uint64_t* keys, int32_t* sum_n, int32_t* n, float* w, float* pr, uint8_t* f32) {
    const int C = P->C;
    int cnt = P->n_nodes < cap ? P->n_nodes : cap;
    for (int i = 0; i < cnt; ++i) {
        const afo_node* nd = P->order[i];
        memcpy(keys + (size_t)i * 8, nd->mine.w, 32); memcpy(keys + (size_t)i * 8 + 4, nd->theirs.w, 32);
        sum_n[i] = nd->sum_n;
        memcpy(n + (size_t)i * C, nd->n, sizeof(int32_t) * (size_t)C);
        memcpy(w + (size_t)i * C, nd->w, sizeof(float) * (size_t)C);
        memcpy(pr + (size_t)i * C, nd->p, sizeof(float) * (size_t)C);
        memcpy(f32 + (size_t)i * C, nd->f32, (size_t)C);
    }
    return P->n_nodes;
}

void afo_stats(const afo_player* P, uint64_t* out) {
    out[0] = P->st_sims; out[1] = P->st_selects; out[2] = P->st_expands; out[3] = P->st_terminals; out[4] = P->st_plies;
}

/* how often a uniform pick had more than one candidate so far: score ties in select (:277-279), forced root visits (:264-276),
 * max-visit ties in calc_policy (:101-102).  With all three at 0 a search is a function of the net alone — no RNG decided
 * anything — which is what lets an eval-mode trace of the reference be compared with an engine that uses another generator. */
void afo_tie_stats(const afo_player* P, uint64_t* out) {
    out[0] = P->st_tie_select; out[1] = P->st_tie_forced; out[2] = P->st_tie_best;
}

uint32_t afo_np_u32(afo_player* P) { return mt_u32(&P->np_rng); }
uint32_t afo_py_u32(afo_player* P) { return mt_u32(&P->py_rng); }
void afo_np_dirichlet(afo_player* P, double alpha, int L, double* out) { mt_dirichlet(&P->np_rng, alpha, L, out); }
double afo_log(double x) { return af_log(x); }
double afo_exp(double x) { return af_exp(x); }
float afo_powf(float x, float y) { return af_powf(x, y); }
float afo_logf(float x) { return af_logf(x); }
float afo_expf(float x) { return af_expf(x); }
void afo_philox(const uint32_t* ctr, const uint32_t* key, uint32_t* out) {
    af_u32x4 r = af_philox4x32(ctr[0], ctr[1], ctr[2], ctr[3], key[0], key[1]);
    memcpy(out, r.v, 16);
}
