/*
 * af_oracle.h — CPU restatement of the alphaFive self-play hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in alphafive_amd/ (the product) may import,
 * link or execute this library; it is the checker used by tests/, by
 * __graft_entry__.smoke() and by bench.py's cpu_baseline leg.
 *
 * Pinned ("tier A") against the unmodified reference imported in the build
 * container: tests/golden/make_golden.py runs /root/reference's
 * genData.player.Player / utils under np.random.seed(s); random.seed(s) and
 * commits per-ply visit-count vectors, actions, episode records and tree dumps;
 * tests/test_oracle_golden.py replays them bit-for-bit through this library in
 * AFO_RNG_MT mode (MT19937 + numpy-legacy samplers restated here).
 * In AFO_RNG_PHILOX mode ("tier B") the arithmetic is identical but noise comes
 * from include/af_noise.h, which is what the HIP engine uses.
 */
#ifndef AF_ORACLE_H
#define AF_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AFO_MAXC 256          /* up to 16x16 cells */
#define AFO_RNG_MT 0          /* tier A: numpy/python MT19937 streams          */
#define AFO_RNG_PHILOX 1      /* tier B: include/af_noise.h counter generator  */

#define AFO_PV_PSEUDO 0       /* built-in integer-hash pseudo-net              */
#define AFO_PV_CALLBACK 1     /* user callback (e.g. torch-CPU real net)       */

typedef struct {
    int board_size;           /* config.py:2  */
    int goal;                 /* config.py:6  */
    int sims;                 /* config.py:4  simulation_per_step       */
    int upper_sims;           /* config.py:5  upper_simulation_per_step */
    double init_temp;         /* config.py:19 */
    double gamma;             /* config.py:18 */
    double tau_decay;         /* config.py:12 */
    double tau_decay_r;       /* config.py:15 */
    double alpha;             /* config.py:17 dirichlet_alpha */
    double c_puct;            /* config.py:16 */
} afo_config;

/* planes float32[3*C] -> policy float32[C] (softmaxed), value float32 */
typedef void (*afo_pv_fn)(const float* planes, float* policy, float* value, void* user);

typedef struct afo_player afo_player;

/* ---- rules & codecs (utils.py:149-296) ---- */
int  afo_board_to_state(const int8_t* board, int S, char* out, int cap);
int  afo_state_to_board(const char* state, int S, int8_t* board);
int  afo_is_game_over(const int8_t* board, int S, int goal, double* value);
int  afo_legal_actions(const int8_t* board, int S, int* cells);
void afo_board_to_inputs(const int8_t* board, int S, int last_cell, float* planes);
void afo_step(int8_t* board, int S, int cell);
void afo_construct_weights(int T, double gamma, float* w);
float afo_pairwise_sum_f32(const float* a, int n);

/* ---- pseudo-net (integer-exact; same spec as tests/pseudonet.py) ---- */
void afo_pseudonet(const float* planes, int C, uint32_t salt, uint32_t peak, float* policy, float* value);

/* ---- player (genData/player.py) ---- */
afo_player* afo_create(const afo_config* cfg, int training, int rng_mode, uint64_t seed,
                       uint32_t game_id, int pv_kind, afo_pv_fn fn, void* user,
                       uint32_t pseudo_salt, uint32_t pseudo_peak);
void afo_destroy(afo_player* p);
void afo_reset(afo_player* p);                         /* player.py:48 */
void afo_set_training(afo_player* p, int training);
void afo_set_simulations(afo_player* p, int sims, int upper);
/* 1 = the reference's pipe path: leaf values arrive as python floats (networkAPI.py:72), so w and q are fp64 (SURVEY 8a rule 2) */
void afo_set_value_f64(afo_player* p, int on);
int  afo_node_get_w64(const afo_player* p, const char* state, double* w);
int  afo_tree_dump_w64(const afo_player* p, int cap, double* w);

/* player.py:128 get_action.  policy_out: float32[C] or NULL; returns 0 ok,
 * 1 = policy is None (eval mode), <0 error.  visits_out int32[C] (0 at occupied). */
int afo_get_action(afo_player* p, const char* state, int last_cell, int random_a,
                   float* policy_out, int* action_cell, int32_t* visits_out);

/* player.py:53 run.  Arrays sized for max_T plies.  states: max_T x 272 chars.
 * Returns T (>0) or <0 on error. final_value is is_game_over's value at the end. */
int afo_run(afo_player* p, int max_T, char* states, float* policies, int* last_cells,
            int* action_cells, int32_t* visits, double* values, float* weights, double* final_value);

/* tree inspection */
int  afo_tree_size(const afo_player* p);
int  afo_node_get(const afo_player* p, const char* state, int* sum_n, int32_t* n, float* w,
                  float* pr, uint8_t* f32);
/* dump all nodes in insertion order: keys[n][8] (mine[4],theirs[4]) etc. */
int  afo_tree_dump(const afo_player* p, int cap, uint64_t* keys, int32_t* sum_n, int32_t* n,
                   float* w, float* pr, uint8_t* f32);
void afo_stats(const afo_player* p, uint64_t* out /* sims, selects, expands, terminals, plies */);
void afo_tie_stats(const afo_player* p, uint64_t* out /* picks among > 1 candidates: score ties, forced root visits, max-visit ties */);
double afo_tau(const afo_player* p);

/* raw RNG access for unit tests of the MT restatement */
uint32_t afo_np_u32(afo_player* p);
uint32_t afo_py_u32(afo_player* p);
void afo_np_dirichlet(afo_player* p, double alpha, int L, double* out);
void afo_noise_philox_dirichlet(double alpha, const uint64_t* legal_bb, int C, uint32_t sel, uint32_t episode,
                                uint32_t k0, uint32_t k1, double* d_by_cell);
double afo_log(double x);
double afo_exp(double x);
float  afo_powf(float x, float y);
float  afo_logf(float x);
float  afo_expf(float x);
void   afo_philox(const uint32_t* ctr, const uint32_t* key, uint32_t* out);

#ifdef __cplusplus
}
#endif
#endif
