// K2 "x3" training forward with EXACT records (DDNERF_X3_WGRAD=exact): the x3 forward (bf16 MFMA on exact hi/lo operand splits, three
// MFMAs per product) that records every layer's output as blocked hi/lo words -- the fp32 value's exact split -- instead of the default
// tier's bf16 row pairs, so that the weight gradients contract fp32-class operands (three MFMAs per product there too).  Round 2's
// kernel, frozen (mlp_mfma16_hilo.inc), with its own weight image.  Reference: train_model.py:154-177 trains in fp32.
#include "common.h"
#define M16_PLANES 2
#define M16_TRAIN
#define M16_SYM(x) ddnerf_mlp_x3e_##x
#define M16_KERNEL mlp_x3e_fwd16_train_kernel
#define M16_PACK_KERNEL mlp_x3e_pack_kernel
#define M16_FEAT_T float
#include "mlp_x3_stages.h"

#include "mlp_mfma16_hilo.inc"
