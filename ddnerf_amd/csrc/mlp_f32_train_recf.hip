// The fp32 training kernels of mlp_f32_train.hip built with `acts` / `deltas` as blocked records of the fp32 VALUES
// (ddnerf_mlp_f32_forward_train_recf, ddnerf_mlp_f32_backward_data_recf; round 5): the layout of the hi/lo-word records
// (mlp_f32_train_rec.hip), the values unsplit.  Their weight gradients run on ddnerf_mlp_x3_wgrad_blocked (mlp_x3_wgrad_packed.hip),
// which splits each value into bf16 hi / lo where it builds its MFMA fragments -- the same split and products, so the same
// gradients bit for bit -- and leaves these kernels' fp32 MFMA chains without the 3.5 vector-ALU instructions per recorded element.
#define F32_REC 3
#include "mlp_f32_train.hip"
