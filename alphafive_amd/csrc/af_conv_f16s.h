// af_conv_f16s.h — internal interface (inside libaf_net.so) between af_net.hip and af_conv_f16s.hip: the fp16
// split-operand convolution path of the network (11x11 and 15x15 boards).  Not part of the C ABI (include/af_net.h is).
#ifndef AF_CONV_F16S_H
#define AF_CONV_F16S_H

#include <hip/hip_runtime.h>

#include <map>
#include <string>
#include <vector>

struct f16s_net;

// variables under their checkpoint names in TF layout (as af_net_set_variable received them); board sizes 11 (one pseudo-position
// per board) and 15 (two half-board pseudo-positions per board); on both the heads are fused (the 1x1 head convolutions ride the last conv
// of each branch, the dense layers run on af_value_fc_f16s / af_policy_fc_f16s<Geo<S>>)
int f16s_supported(int board_size);
int f16s_create(f16s_net** out, int board_size, int max_batch, int device, const std::map<std::string, std::vector<float>>& vars);
void f16s_destroy(f16s_net* n);
// stem + bone/block1 + bone/block2 on stream st
int f16s_trunk(f16s_net* n, hipStream_t st, const float* planes_dev, int batch);
// value/block3 -> o3, policy/block4+5 -> o5: fp32 planes [batch][32][PP], pixel (y,x) at (y+1)*WP + x+1 (the head kernels' input)
// value / policy != nullptr: the branch's last conv also applies the head's 1x1 convolution and the head's dense layers run on the
// same split-operand MFMA (af_value_fc_f16s / af_policy_fc_f16s) -> value [batch] / policy [batch][121]; o3 / o5 are then not written
int f16s_value_branch(f16s_net* n, hipStream_t st, int batch, float* o3_dev, int WP, int PP, float* value);
int f16s_policy_branch(f16s_net* n, hipStream_t st, int batch, float* o5_dev, int WP, int PP, float* policy);
// batches <= 8 on 11x11, heads fused: both branches on one stream ({policy conv1 || value block} in one launch): no side stream
int f16s_small_branches_ok(const f16s_net* n, int batch);          // (abl bit 9 = the two-stream form, for A/B)
int f16s_small_branches(f16s_net* n, hipStream_t st, int batch, float* value, float* policy);
// batches <= 8 on 11x11 (r6): the whole forward up to the policy head's input as ONE launch of dataflow roles + the policy dense layer
int f16s_small_forward_ok(const f16s_net* n, int batch);           // (abl bit 11 = the multi-launch form, for A/B)
int f16s_small_forward(f16s_net* n, hipStream_t st, const float* planes_dev, int batch, float* value, float* policy);
int f16s_small_forward_error(f16s_net* n);                         // 1 if a role ever gave up waiting (synchronises)
void f16s_set_ablation(f16s_net* n, int bits);
int f16s_read_activation(f16s_net* n, int which, int batch, float* host);

#endif
