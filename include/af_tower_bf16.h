/*
 * af_tower_bf16.h — C ABI of the hand-written gfx950 bf16 residual tower (libaf_tower.so) used by
 * BASELINE configs[4] (11x11, 8 residual blocks of the reference's block type at
 * constant width 128, bf16, performance-only).  One block is genData/network.py:52-56:
 *     r = conv1x1(h) + b_res ;  g = ELU(conv3x3(h) + b1) ;  h' = ELU(r + conv3x3(g) + b2)
 * evaluated as two launches of one kernel (the projection is 8 extra k-steps of the second one) on
 * v_mfma_f32_32x32x16_bf16, fp32 accumulation, activations stored in bf16 like the PyTorch-ROCm bf16
 * evaluator it replaces (alphafive_amd/network_deep.py).
 *
 * Activation layout ("C8"): bf16 [batch][width/8][PIX][8], PIX = (S+2)*S rounded up to 16 (144 at S = 11): rows
 * are stored back to back with one zero row above and below the board, pixel (y, x) at unit S*(y+1) + x (a unit =
 * the 8 channels of one pixel = 16 bytes).  The kernels never write the two zero rows; the missing left / right
 * neighbours of a row's first / last pixel are handled inside the kernel.  The caller owns the two buffers.
 *
 * Plain pointers, int return codes (0 ok, <0 error), no exceptions; one handle per GPU.
 */
#ifndef AF_TOWER_BF16_H
#define AF_TOWER_BF16_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct af_tower af_tower;

#define AF_TOWER_OK 0
#define AF_TOWER_ERR_ARG   (-1)
#define AF_TOWER_ERR_HIP   (-2)
#define AF_TOWER_ERR_STATE (-3)   /* forward before every block was set */

/* board_size must be 11 and width 128 in this build (4 waves x 32 output channels, 4 pixel tiles). */
int af_tower_create(int32_t board_size, int32_t width, int32_t blocks, int32_t device, af_tower** out);
void af_tower_destroy(af_tower* t);

/* Weights of block `b` (host pointers, fp32, PyTorch OIHW): c1_w/c2_w [W][W][3][3], res_w [W][W][1][1], biases [W].
 * Values are rounded to bf16 (round-to-nearest-even) when packed. */
int af_tower_set_block(af_tower* t, int32_t b, const float* c1_w, const float* c1_b, const float* c2_w, const float* c2_b,
                       const float* res_w, const float* res_b);

/* Stem (network.py:63 shape at width 128): 5x5 conv 3 -> W, SAME, + bias + ELU.  w [W][3][5][5] OIHW, b [W]. */
int af_tower_set_stem(af_tower* t, const float* w, const float* b);
/* The heads' 1x1 convolutions (network.py:70,82): value [4][W] + [4], policy [16][W] + [16] (OIHW with 1x1 dropped). */
int af_tower_set_heads(af_tower* t, const float* vconv_w, const float* vconv_b, const float* pconv_w, const float* pconv_b);

int32_t af_tower_pix(const af_tower* t);           /* PIX of the C8 layout */
int64_t af_tower_plane_elems(const af_tower* t);   /* bf16 elements per position = width * PIX */

/* Runs all blocks in place on x_dev (C8 bf16, zero borders); g_dev is scratch of the same size whose
 * borders must be zero as well.  Asynchronous on `stream` (hipStream_t; NULL = default stream). */
int af_tower_forward(af_tower* t, void* stream, void* x_dev, void* g_dev, int32_t batch);

/* planes_dev float32[batch][3][S][S] (utils.py:256 layout) -> x_dev (C8 bf16; its zero rows are left untouched). */
int af_tower_stem(af_tower* t, void* stream, const float* planes_dev, void* x_dev, int32_t batch);
/* x_dev (C8 bf16) -> ELU(1x1 conv + bias), flattened NCHW as the dense layers take it (network.py:71,83):
 * vin_dev bf16 [batch][4*S*S], pin_dev bf16 [batch][16*S*S]. */
int af_tower_heads(af_tower* t, void* stream, const void* x_dev, void* vin_dev, void* pin_dev, int32_t batch);

/* The three dense layers + softmax behind the heads' 1x1 convolutions (network.py:73-76,85-88) on the same MFMA:
 * vfc1 [4*S*S][64] + [64] (ELU), vfc2 [64][1] + [1] (tanh(x/2)), pfc [16*S*S][S*S] + [S*S] (softmax); host fp32 [in][out],
 * rounded to bf16 when packed.  af_tower_dense: vin / pin (af_tower_heads' outputs) -> policy float32[batch][S*S],
 * value float32[batch] — what the engine's tick consumes. */
int af_tower_set_dense(af_tower* t, const float* vfc1_w, const float* vfc1_b, const float* vfc2_w, const float* vfc2_b,
                       const float* pfc_w, const float* pfc_b);
int af_tower_dense(af_tower* t, void* stream, const void* vin_dev, const void* pin_dev, float* policy_dev, float* value_dev,
                   int32_t batch);

/* A/B knobs (process-global): key 0 = B-fragment ring depth (0 = per-kernel default, 8, 12, 16), key 1 = persistent
 * workgroups (0 = one per CU), key 2 = profiling ablation bits (results wrong by design: 1 no re-staging, 2 no stores;
 * "no LDS reads" is the build-time macro AF_TOWER_ABL_NOLDS since r3: a run-time test in the MFMA loop was not free), key 3 = convolution kernels
 * (3 = default: af_tower_conv3 — epilogue under the other tile pair's MFMAs — for a block's first convolution and af_tower_conv for its
 * second; 0 = af_tower_conv for both, bit-identical to 3; 2 = af_tower_conv3 for both), key 4 = the heads' 1x1 convolutions (1 MFMA kernel,
 * 0 VALU kernel). */
int af_tower_tune(int32_t key, int32_t value);

int64_t af_tower_flops_per_position(const af_tower* t);   /* 2*MAC of the tower, direct convolution */
const char* af_tower_strerror(int code);

#ifdef __cplusplus
}
#endif
#endif
