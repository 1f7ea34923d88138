/*
 * af_engine.h — C ABI of the MI355X-native alphaFive self-play engine (libaf_hip.so).
 *
 * Drop-in boundary for ONE path of GuoYi0/alphaFive: MCTS self-play move
 * generation.  The reference implements it in Python (no FFI of its own); a
 * maintainer binds these entry points with ctypes from
 *   genData/player.py:23-284   Player.get_action / run / MCTS_search / select /
 *                              evaluate_and_expand / update_tree / calc_policy
 *   utils.py:149-296           state codec, is_game_over, legal moves, inputs
 *   genData/networkAPI.py:43   leaf batching (replaced by the device leaf batch)
 *   main.py:82-94              gen_data episode hand-off
 *
 * Conventions: plain pointers and sizes, int return codes (0 = ok, <0 = error,
 * see af_strerror), caller-allocated outputs, no exceptions across the ABI, one
 * engine handle per GPU, a handle is not thread-safe.  `*_dev` pointers are HIP
 * device pointers on the engine's device (e.g. torch tensors' data_ptr()).
 *
 * Board convention everywhere (utils.py:185,194,275-283): cell c = i*S + j,
 * +1 = stone of the player to move ("mine"), -1 = opponent ("theirs").
 * A position key is 2*KW uint64 words: mine bitboard then theirs bitboard,
 * bit c of word c>>6; KW = 2 for S*S <= 128, 4 for S*S <= 256.
 */
#ifndef AF_ENGINE_H
#define AF_ENGINE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AF_ABI_VERSION 6

/* engine modes */
#define AF_MODE_SELFPLAY 0   /* Player.run loop on device: games restart forever (main.py:82 gen_data) */
#define AF_MODE_EXTERNAL 1   /* Player.get_action: host sets roots, engine parks when a move is decided */
/* OR into `mode`: leaf values arrive as python floats — the reference's pipe path (networkAPI.py:72 `float(v)`, what
 * main.py's workers run): every edge's W is an fp64 running sum and Q = W/N in fp64 (SURVEY 8a rule 2, pipe variant).
 * Without it W follows the pv_fn path (self_play.py): fp32 running sum + the "only terminal values so far" exception. */
#define AF_MODE_VALUE_F64 0x100

/* per-game status after af_engine_tick */
#define AF_STATUS_IDLE      0
#define AF_STATUS_NEED_EVAL 1   /* a leaf's input planes were written; feed policy/value to the next tick */
#define AF_STATUS_MOVE_DONE 2   /* EXTERNAL mode: read af_engine_move_result, then set the next root */
#define AF_STATUS_YIELD     3   /* the game used up its per-launch work budget (or ran its store collector, or waits
                                 * for its finished episodes to be popped): nothing to evaluate, tick again */

/* selects a game may do in one launch before it yields at the next simulation boundary (a launch lasts as long
 * as its slowest game; simulations that end in terminal positions never park).  Results do not depend on it. */
#define AF_DEFAULT_TICK_BUDGET 8
/* hard cap: past this many selects in one launch a game yields even inside a simulation (AF_TICK_BUDGET_HARD) */
#define AF_DEFAULT_TICK_BUDGET_HARD 12

/* error codes */
#define AF_OK 0
#define AF_ERR_ARG        (-1)
#define AF_ERR_HIP        (-2)
#define AF_ERR_NODE_CAP   (-3)  /* a game's transposition store is full */
#define AF_ERR_NO_ROOT    (-4)  /* get_action on a finished position (reference: IndexError at player.py:102) */
#define AF_ERR_EP_OVERRUN (-5)  /* internal assertion: a game never overwrites an un-popped episode (it waits instead) */
#define AF_ERR_STATE      (-6)

/* config.py:2-20 — the attributes Player reads (player.py:31,32,44,77,109,111,141,143,240,261) */
typedef struct {
    int32_t board_size;            /* config.py:2  */
    int32_t goal;                  /* config.py:6  */
    int32_t simulation_per_step;   /* config.py:4  */
    int32_t upper_simulation_per_step; /* config.py:5 */
    double init_temp;              /* config.py:19 */
    double gamma;                  /* config.py:18 (host side: utils.construct_weights) */
    double tau_decay_rate;         /* config.py:12 */
    double tau_decay_rate_r;       /* config.py:15 */
    double dirichlet_alpha;        /* config.py:17 */
    double c_puct;                 /* config.py:16 */
} af_config;

typedef struct af_engine af_engine;

/* Create an engine holding `num_games` concurrent games on HIP device `device`.
 * training: Player(training=...) (player.py:23).  seed + (first_game_id + g) key the
 * per-game noise streams (include/af_noise.h).  node_cap: transposition-store
 * capacity per game in nodes (0 = default for the mode). */
int af_engine_create(const af_config* cfg, int32_t num_games, int32_t device, int32_t mode, int32_t training,
                     uint64_t seed, uint32_t first_game_id, int32_t node_cap, af_engine** out);
void af_engine_destroy(af_engine* e);

/* geometry */
int32_t af_engine_num_games(const af_engine* e);
int32_t af_engine_cells(const af_engine* e);        /* C = S*S            */
int32_t af_engine_key_words(const af_engine* e);    /* 2*KW               */
int32_t af_engine_max_plies(const af_engine* e);    /* C                  */

/* One tick (player.py:204 MCTS_search, batched): every game first consumes the
 * evaluation of the leaf it parked on (policy_dev[g][C] softmax probs, value_dev[g];
 * player.py:186 evaluate_and_expand + :166 update_tree), then descends
 * (terminal test utils.py:199, select player.py:230, step utils.py:275), finishing
 * simulations that end in terminal positions and whole moves (player.py:84
 * calc_policy) on the way, until it parks on the next unseen leaf, whose input
 * planes (utils.py:256 board_to_inputs) it writes to planes_dev[g][3][S][S].
 * Asynchronous on `stream` (a hipStream_t, NULL = default stream). */
int af_engine_tick(af_engine* e, void* stream, const float* policy_dev, const float* value_dev, float* planes_dev);

/* Copy per-game status words to the host (synchronises `stream`). Returns the
 * first negative per-game error if any game failed, else 0. */
int af_engine_status(af_engine* e, void* stream, int32_t* status_host);

/* EXTERNAL mode: Player.get_action(state, last_action, random_a) for game g.
 * key: 2*KW words.  last_cell: -1 for None.  reset_tree != 0: Player.reset() first. */
int af_engine_set_root(af_engine* e, int32_t game, const uint64_t* key, int32_t last_cell, int32_t random_a,
                       int32_t reset_tree);
/* result of the finished move: action cell, temperature policy float32[C] (has_policy == 0
 * when the reference returns None, player.py:106), root visit counts int32[C], tau after. */
int af_engine_move_result(af_engine* e, int32_t game, int32_t* action_cell, int32_t* has_policy, float* policy,
                          int32_t* visits, double* tau);
int af_engine_set_training(af_engine* e, int32_t training);
/* Simulation budget of the moves that START after this call (player.py:140-143 reads config.simulation_per_step /
 * upper_simulation_per_step from the live config object at every get_action; the engine snapshots them at create).
 * A move in progress keeps the budget it started with.  AF_ERR_ARG unless 1 <= sims, 1 <= upper and sims + 8 <= the
 * node capacity chosen at create (the store collector's head-room). */
int af_engine_set_simulations(af_engine* e, int32_t simulation_per_step, int32_t upper_simulation_per_step);

/* The same for n games at once (batched arena, choose_best_player.py:38-60): one upload + one launch / one launch + one
 * download on `stream` instead of a device synchronisation and ~10 small copies per game.  keys [n][2KW]; random_a and
 * reset_tree may be NULL (= 0); policies / visits [n][C], has_policy, taus may be NULL. */
int af_engine_set_roots(af_engine* e, void* stream, int32_t n, const int32_t* games, const uint64_t* keys, const int32_t* last_cells,
                        const int32_t* random_a, const int32_t* reset_tree);
int af_engine_move_results(af_engine* e, void* stream, int32_t n, const int32_t* games, int32_t* action_cells, int32_t* has_policy,
                           float* policies, int32_t* visits, double* taus);

/* SELFPLAY mode, device-resident hand-off of finished episodes (main.py:94 `q.put`): two small kernels on `stream`
 * compact every finished, not yet popped episode — in (game, sequence) order, at most max_episodes episodes and
 * max_plies plies; the rest waits for the next call — into ONE int32 buffer and mark them popped.  No host
 * synchronisation, no copy: `out` is device memory (hand it to RCCL) or pinned host memory the device can write
 * (read it after an event).  Layout, R = 2*KW2 + 2*C + 2 ints per ply:
 *   out[0] = episodes n, out[1] = plies, out[2] = KW2 (key words), out[3] = C
 *   out[4 + 4*i ..]                      meta of episode i: game, episode_seq, T, first ply index
 *   out[4 + 4*max_episodes + i]          final value (float bits): is_game_over value of the last position
 *   out[4 + 5*max_episodes + R*p ..]     ply p: key (KW2 uint64) | policy float[C] | visits int[C] | last cell | action cell
 * af_engine_pack_ints = the buffer size in ints for the given caps. */
int64_t af_engine_pack_ints(const af_engine* e, int32_t max_episodes, int32_t max_plies);
int af_engine_pack_episodes(af_engine* e, void* stream, int32_t max_episodes, int32_t max_plies, int32_t* out);

/* The same hand-off into caller-allocated padded host arrays (one pack, one copy of the used prefix, synchronous):
 * pop finished episodes (Player.run's return value before value/weight
 * assembly, which is host-side arithmetic on T and final_value).
 * meta[i] = {game, episode_seq, T, 0}; final_value[i] = is_game_over value of the last
 * position; per ply: keys [cap][max_plies][2KW], policies/visits [cap][max_plies][C],
 * lasts/actions [cap][max_plies].  Returns the number popped (<= cap) or <0. */
int af_engine_pop_episodes(af_engine* e, void* stream, int32_t cap, int32_t* meta, float* final_value,
                           uint64_t* keys, float* policies, int32_t* visits, int32_t* lasts, int32_t* actions);

/* counters summed over games: out[0..8] = sims, selects, expands, terminal hits, plies,
 * episodes, sum of L over selects, sum of L over expands, nodes currently stored;
 * out[9..12] = store-collector runs, node slots it scanned, yields, episode-buffer stalls */
#define AF_NUM_COUNTERS 13
int af_engine_counters(af_engine* e, void* stream, uint64_t* out);

/* ABI v5 — evaluation memo (optional, off by default).  The net is a pure function of a leaf's (stones, last move) planes
 * (utils.py:256-272) and this package's forward is independent of batch slot and batch size to the bit, so a position that ANY
 * game of the engine has had evaluated before need not wait a tick for the same bits again: with the memo enabled a game that
 * reaches an unseen position holding at most `max_stones` stones looks it up (4-way buckets, exact key = both bitboards + last
 * move) and on a hit expands it at once (player.py:186-202) and goes on with its next simulation inside the same launch.  Every
 * game still owns its tree and every tree is bit-identical to the run without the memo; only the number of ticks differs.
 *   af_engine_memo_enable  allocates 4 << log2_buckets entries of (32 KW + 256 KW + 12) bytes; once per engine.
 *   af_engine_memo_insert  enqueue after the forward of a tick (same stream): stores the evaluations of the leaves parked by the
 *                          last af_engine_tick, read from the same policy/value buffers the next tick will consume.
 *   af_engine_memo_clear   enqueue whenever the evaluator's weights change (stream-ordered): the stored bits are the old net's.
 *                          O(1) since ABI v6: every entry carries the epoch it was written in, a clear bumps the epoch the
 *                          following launches are issued with, older entries stop matching and their ways count as empty.
 *   af_engine_memo_epoch   (ABI v6) the current epoch (>= 1; 0 = memo off).  Launch parameters travel by value: whoever replays a
 *                          captured graph of af_engine_tick / af_engine_memo_insert keys it on this as well.
 *   af_engine_memo_stats   out[0..5] = launches, probes, hits, inserts, replacements, entries allocated (synchronises). */
#define AF_MEMO_STATS 6
int af_engine_memo_enable(af_engine* e, int32_t log2_buckets, int32_t max_stones);
int af_engine_memo_insert(af_engine* e, void* stream, const float* policy_dev, const float* value_dev);
int af_engine_memo_clear(af_engine* e, void* stream);
int af_engine_memo_stats(af_engine* e, void* stream, uint64_t* out);
int64_t af_engine_memo_epoch(af_engine* e);

/* launch-shape evidence, accumulated over all games and launches since the last reset:
 * out[0..63] = histogram of selects per game and launch (63 = 63 or more), out[64..95] = histogram of
 * a game's wave lifetime in 8-us bins (31 = 248 us or more), out[96] = longest wave lifetime in 10-ns units */
#define AF_HIST_WORDS 97
int af_engine_tick_histogram(af_engine* e, void* stream, uint64_t* out, int32_t reset);

/* override AF_DEFAULT_TICK_BUDGET (also settable through the environment: AF_TICK_BUDGET) */
int af_engine_set_tick_budget(af_engine* e, int32_t selects_per_launch);

/* cheap progress poll (16-byte copy): out[0] = plies committed, out[1] = episodes finished */
int af_engine_progress(af_engine* e, void* stream, uint64_t* out);
/* ABI v4: the same 16-byte copy WITHOUT the synchronisation — `out_pinned` must be page-locked host memory (or device memory)
 * that stays valid until the copy has run.  Stream-ordered and capturable: a driver that replays n x (tick + leaf evaluation)
 * as one HIP graph ends the graph with this copy and reads the words after the replay's event, one replay later, so that the
 * poll never drains the device (main.py:57-76's loop blocks on its Queue instead: the same role, without the wait). */
int af_engine_progress_async(af_engine* e, void* stream, uint64_t* out_pinned);

/* ABI v6: device-clock stamps — what a HIP timing event would be if ROCm allowed one inside a captured graph.  af_engine_stamp enqueues
 * a one-wave kernel that writes the device-wide 100-MHz clock (s_memrealtime, 10-ns ticks) into slot `slot` (0 .. AF_STAMP_SLOTS-1) of the
 * engine's stamp array; af_engine_stamps_async copies the AF_STAMP_SLOTS slots into pinned host memory, stream-ordered (both capturable).
 * A driver that replays n x (stamp, af_engine_tick, stamp, forward, stamp) reads each kernel's duration as it runs inside the graph —
 * including the two launch boundaries the stamp kernels add around it (~1.5 us each): an upper bound of the kernel's own time.
 * (bench.py: roofline.ms_per_launch; reference: there is nothing to time in genData/player.py — the measurement contract is this task's.) */
#define AF_STAMP_SLOTS 64
int af_engine_stamp(af_engine* e, void* stream, int32_t slot);
int af_engine_stamps_async(af_engine* e, void* stream, uint64_t* out_pinned);

/* tree inspection (tests / Player.tree): nodes of game g in storage order.
 * n has the "w is fp32-typed" flag stripped into f32[]. Returns node count or <0. */
int af_engine_tree_dump(af_engine* e, int32_t game, int32_t cap, uint64_t* keys, int32_t* sum_n, int32_t* n,
                        float* w, float* p, uint8_t* f32);

/* Player.reset(search_tree) (player.py:48-51), EXTERNAL mode: replace game `game`'s store by `count` nodes in
 * af_engine_tree_dump's format (keys [count][2KW], sum_n [count], n / w / p [count][C] by cell, f32 [count][C]).
 * Follow with af_engine_set_root(..., reset_tree = 0). */
int af_engine_load_tree(af_engine* e, int32_t game, int32_t count, const uint64_t* keys, const int32_t* sum_n,
                        const int32_t* n, const float* w, const float* p, const uint8_t* f32);

/* AF_MODE_VALUE_F64 engines: the exact fp64 W rows ([count][C], af_engine_tree_dump's node order; its float `w` is
 * the rounded value there), and the way in after af_engine_load_tree. */
int af_engine_tree_w64(af_engine* e, int32_t game, int32_t cap, double* w);
int af_engine_set_tree_w64(af_engine* e, int32_t game, int32_t count, const double* w);

/* state-string codec (utils.py:156-196) <-> position key */
int af_state_to_key(const char* state, int32_t board_size, uint64_t* key);
int af_key_to_state(const uint64_t* key, int32_t board_size, char* out, int32_t cap);

const char* af_strerror(int code);
int af_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* AF_ENGINE_H */
