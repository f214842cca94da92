// The fp16 tier's fused encoder + MLP kernel: mlp_bf16_g2e.hip compiled a second time on the f16 forms of the instructions (the body generated
// by gen_bf16_g2.py, mlp_f16_g2e_body_d*.gen.inc; entry point ddnerf_encode_mlp_f16_forward; bit-identical to ddnerf_encode(feat_dtype = 2) +
// ddnerf_mlp_f16_forward).
#define G2E_HALF 1
#include "mlp_bf16_g2e.hip"
