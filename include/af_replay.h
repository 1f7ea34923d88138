/*
 * af_replay.h — C ABI of the device-resident replay buffer (libaf_replay.so): the storage and the
 * sampling/augmentation half of the reference's utils.RandomStack (utils.py:14-146) on the GPU, so that
 * self-play positions stay in HBM for the trainer.
 *
 * Division of labour (drop-in: alphafive_amd/replay.py:DeviceRandomStack keeps RandomStack's call surface):
 *   host  — the scalar bookkeeping of push() (utils.py:65-116: short-game rejection, colour re-balancing
 *           duplicates, FIFO eviction with partial-episode accounting) and every random draw of get_data()
 *           (utils.py:122,129,136: which positions, how many quarter turns, flip or not), taken from the same
 *           global streams in the same order as the reference, so seeded runs stay reproducible against it;
 *   device — the positions (a ring of boards int8[C], policies float32[C], last-move cell, value, weight) and
 *           get_data()'s per-sample work (utils.py:127-145): rot90^k + vertical flip of board and policy,
 *           the last_action remap, board_to_inputs' three planes, gathered into the batch tensors.
 * Results are bit-identical to the host class (pure gathers of fp32 / small-integer data).
 *
 * Plain pointers, int return codes (0 ok / count, <0 error), no exceptions; one handle per GPU.
 */
#ifndef AF_REPLAY_H
#define AF_REPLAY_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct af_replay af_replay;

#define AF_REPLAY_OK 0
#define AF_REPLAY_ERR_ARG   (-1)
#define AF_REPLAY_ERR_HIP   (-2)
#define AF_REPLAY_ERR_FULL  (-3)   /* append beyond capacity: drop from the front first */
#define AF_REPLAY_ERR_RANGE (-4)   /* sample index / drop count outside the stored range */

/* capacity = most positions ever resident at once (RandomStack.length + the longest episode pushed twice). */
int af_replay_create(int32_t board_size, int32_t capacity, int32_t device, af_replay** out);
void af_replay_destroy(af_replay* r);

/* Append n positions at the tail (host pointers; copied asynchronously on `stream`, the call returns after the
 * staging copy so the host arrays may be reused): boards int8[n][C] (+1 mine / -1 theirs / 0, utils.py:185),
 * policies float32[n][C], last_cell int32[n] (i*S+j, or -1 for None), values float32[n], weights float32[n]. */
int af_replay_append(af_replay* r, void* stream, int32_t n, const int8_t* boards, const float* policies,
                     const int32_t* last_cell, const float* values, const float* weights);
/* Device-to-device append ("keep episodes on-GPU for the trainer"): episode `episode` of a packed hand-off buffer
 * that af_engine_pack_episodes (include/af_engine.h) wrote into DEVICE memory with the same max_episodes.  One launch on `stream`
 * decodes every ply of it on the device — position key -> board int8 (+1 mine / -1 theirs: utils.py:185 state_to_board of the
 * recorded root), temperature policy, last move, value = the final value with the signs of player.py:74-82, weight = row T of the
 * table below — straight into the ring.  `T` (the episode's length, which the host reads from the buffer's small header for
 * RandomStack.push's bookkeeping anyway) sizes the launch and must equal the length recorded in the buffer (checked on the
 * device: a mismatch appends nothing and raises the flag af_replay_check() reports).  No host copy of the
 * episode, no per-ply host work. */
int af_replay_append_packed(af_replay* r, void* stream, const int32_t* packed_dev, int32_t max_episodes, int32_t episode, int32_t T);
/* The per-ply training weights utils.construct_weights(T, gamma) (utils.py:286-296; numpy float32 arithmetic) for every episode
 * length: table[max_T + 1][max_T] float32, row T holds w[0..T-1].  Computed by the host mirror (the numpy restatement is the
 * spec: pairwise float32 sum), uploaded once per gamma. */
int af_replay_set_weights(af_replay* r, const float* table_host, int32_t max_T);
/* AF_REPLAY_ERR_RANGE if a packed append since the last call found the buffer not to hold the episode the host described (it
 * appended nothing then); synchronises `stream`.  AF_REPLAY_OK otherwise. */
int af_replay_check(af_replay* r, void* stream);

/* Forget the n oldest positions (utils.py:103 `del self.data[:beyond]`). */
int af_replay_drop_front(af_replay* r, int32_t n);
int32_t af_replay_size(const af_replay* r);

/* get_data (utils.py:118-146) for `num` samples: idx int32[num] = positions counted from the oldest,
 * quarter_turns int32[num] in 0..3 (np.rot90 k, counter-clockwise), flip int32[num] (1 = np.flip axis 0 applied
 * after the rotation).  Device outputs: boards float32[num][3][S][S] (board_to_inputs planes of the transformed
 * board with the remapped last action), weights float32[num], values float32[num], policies float32[num][C]. */
int af_replay_sample(af_replay* r, void* stream, int32_t num, const int32_t* idx, const int32_t* quarter_turns,
                     const int32_t* flip, float* boards_dev, float* weights_dev, float* values_dev, float* policies_dev);

const char* af_replay_strerror(int code);

#ifdef __cplusplus
}
#endif
#endif
