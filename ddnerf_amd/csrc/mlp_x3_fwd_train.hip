// K2 "x3" training forward: the inference kernel of mlp_x3_fwd.hip (same body, same weight image) that also RECORDS what the
// backward pass and the weight gradients need, so that the training forward runs on the 16x16x32 formulation too:
//   * every layer's output, and the tile's encoded features, as blocked hi/lo words (mlp_x3_common.h) -- the operand format of
//     mlp_x3_wgrad_packed.hip; the words are one v_perm_b32 away from the hi / lo pairs the re-pack computes anyway;
//   * the ReLU sign words of mlp_x3_train.hip's backward-data kernel (its 32x32 tile / lane-half filing: the two nibbles of a
//     byte sit in lanes g and g + 2 here and meet through one v_permlane32_swap).
// The stores are four more re-pack steps per tile and column block, dealt over the MFMA gaps like the rest; they are counted
// into the vmcnt that certifies a weight stage (vector memory operations retire in order).
#include "common.h"
#define M16_PLANES 2
#define M16_TRAIN
#define M16_NO_PACK
#define M16_SYM(x) ddnerf_mlp_x3_##x
#define M16_KERNEL mlp_x3_fwd16_train_kernel
#define M16_FEAT_T float
#include "mlp_x3_stages.h"

#include "mlp_mfma16.inc"
