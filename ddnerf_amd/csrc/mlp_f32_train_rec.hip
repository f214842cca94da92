// The fp32 training kernels of mlp_f32_train.hip built a second time with `acts` / `deltas` as records of blocked hi/lo words
// (ddnerf_mlp_f32_forward_train_rec, ddnerf_mlp_f32_backward_data_rec): what the fp32 tier uses with the default (bf16x3) weight
// gradients, which then run on the packed-operand kernel (mlp_x3_wgrad_packed.hip).  The forward / backward arithmetic is the
// exact-fp32 arithmetic of the first build; only what is recorded differs (a value's exact hi/lo split instead of the value).
#define F32_REC 1
#include "mlp_f32_train.hip"
