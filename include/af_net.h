/*
 * af_net.h — C ABI of the hand-written gfx950 forward pass of the reference's policy/value
 * network (libaf_net.so).  Replaces, for the self-play path, what genData/network.py:90-97
 * (ResNet.eval -> sess.run([prob, value])) executes: the graph of network.py:52-88,163-165
 *   conv5x5(3->32)+ELU, residual(64), residual(128), value head (residual(32), 1x1->4, fc 64, fc 1,
 *   tanh(x/2)), policy head (residual(64), residual(32), 1x1->16, fc S*S, softmax)
 * in fp32 on the matrix cores (v_mfma_f32_32x32x2_f32: exact fp32 products, fp32 accumulate).
 *
 * Variables are handed over under their checkpoint names in TF layout (conv kernels HWIO
 * [kh][kw][cin][cout], dense [in][out], flatten order NCHW), exactly as
 * alphafive_amd.tensorbundle reads them from ckpt/alphaFive-*.  Plain pointers, int return
 * codes (0 ok, <0 error), no exceptions; one handle per GPU; not thread-safe per handle.
 */
#ifndef AF_NET_H
#define AF_NET_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct af_net af_net;

#define AF_NET_OK 0
#define AF_NET_ERR_ARG   (-1)
#define AF_NET_ERR_HIP   (-2)
#define AF_NET_ERR_NAME  (-3)   /* unknown variable name or wrong element count */
#define AF_NET_ERR_STATE (-4)   /* forward before finalize / missing variables */

/* board_size S (3..16); max_batch = largest batch af_net_forward will be called with. */
int af_net_create(int32_t board_size, int32_t max_batch, int32_t device, af_net** out);
void af_net_destroy(af_net* n);

/* Provide one variable (host pointer, fp32, TF layout) — e.g. "bone/block1_conv1/kernel". */
int af_net_set_variable(af_net* n, const char* tf_name, const float* host_data, int64_t count);
/* Repack (k-pair-major streams for the MFMA kernels) and upload; call after all 42 variables are set. */
int af_net_finalize(af_net* n);

/* planes_dev float32[batch][3][S][S] (utils.py:256 board_to_inputs layout) ->
 * policy_dev float32[batch][S*S] (softmax probabilities), value_dev float32[batch].
 * Asynchronous on `stream` (hipStream_t; NULL = default stream). */
int af_net_forward(af_net* n, void* stream, const float* planes_dev, int32_t batch, float* policy_dev, float* value_dev);

/* Benchmark / A-B knobs (process-global; an unknown key or value returns AF_NET_ERR_ARG):
 *   key 0: conv path — 5 (default) fp16 split-operand implicit GEMM (af_conv_f16s.hip; 11x11 and 15x15 boards: three fp16 MFMA products
 *          per MAC, fp32 accumulate, 22 operand mantissa bits), 1 fp32 MFMA Winograd F(2x2,3x3) (af_net.hip: 24 bits — the path of every
 *          other board size, and bench.py's config2_fp32mfma leg)
 *   key 4: value branch on a side stream (default 1: on the fp32 path only; 2 forces it on path 5 too, for A/B)
 *   key 5: MFMA policy head of the fp32 path (default 1)
 *   key 7: bits of path 5 — 1 / 2 / 4 / 8 / 4096 profiling ablations (results wrong by design: no slab loads after the first position /
 *          no stores / loads from a hot address / stores, loads addressed modulo 128 positions), 16 VALU stem, 32 one workgroup per CU in
 *          the 32-channel-input layers; launch structures, each bit selecting the launch it replaced (same results bit for bit):
 *          64 two-halves launch on 15x15 (instead of the 4-tile / 3-tile classes + corner kernel), 128 one workgroup per position at
 *          batches <= 8 (instead of the pixel-tile split), 256 two launches per 32-wide block (instead of af_block_f16s),
 *          512 / 1024 the two-branch launch order at batches <= 8 / above (instead of the value branch as a workgroup class),
 *          2048 (r6) nine dependent launches at batches <= 8 (instead of the single launch of dataflow roles)
 *   key 9: path 5 computes the heads itself — 1x1 head convolutions fused into the last conv of each branch, dense layers on
 *          the same split-operand MFMA (default 1); 0 = the fp32 head kernels of path 1 on fp32 planes
 * (removed in r6 with the kernels they selected: conv paths 0 / 2 / 3 / 4, keys 1 / 2 / 3 / 6 / 8) */
int af_net_tune(int32_t key, int32_t value);

/* Tests / debugging of the fp16 split-operand path (conv path 5, 11x11 boards): intermediate activation `which`
 * (0 stem, 1/2 block1 conv1/output, 3/4 block2, 5 block3 conv1, 6/7 block4, 8 block5 conv1) of the first `batch`
 * positions of the last forward, as fp32 [batch][C][121] on the host.  Returns the channel count C or <0. */
int af_net_debug_activation(af_net* n, int32_t which, int32_t batch, float* host_out);

/* Batches of at most 8 positions on 11x11 (what genData/player.py:186-202 asks for: one leaf per simulation) run as ONE launch of
 * dataflow roles + the policy head's dense layer (csrc/af_conv_f16s.hip: af_small_forward_f16s; af_net_tune(7, 2048) = the nine
 * dependent launches, for A/B).  A role's wait for its producers is bounded; this returns 1 if any wait ever gave up (the outputs
 * of that forward are then undefined), 0 if none did, <0 without the split-operand path.  Synchronises the device. */
int af_net_small_forward_error(af_net* n);

/* FLOPs (2*MAC) of one position's forward pass, as executed (direct convolution). */
int64_t af_net_flops_per_position(const af_net* n);
const char* af_net_strerror(int code);

#ifdef __cplusplus
}
#endif
#endif
