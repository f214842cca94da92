// The fp32 training kernels of mlp_f32_train.hip built a third time with `acts` / `deltas` as records of bf16 ROW PAIRS
// (ddnerf_mlp_f32_forward_train_recp, ddnerf_mlp_f32_backward_data_recp): the opt-in speed mode of the fp32 tier (DDNERF_WGRAD=pairs) --
// exact-fp32 forward / backward-data arithmetic as in the other builds, half the record bytes of the hi/lo-word build, and weight
// gradients on the x3 tier's one-MFMA kernel (bf16-rounded operands: not fp32-class).  Reference: train_model.py:154-177.
#define F32_REC 2
#include "mlp_f32_train.hip"
