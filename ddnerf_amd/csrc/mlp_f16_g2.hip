// K2 (fp16), "two groups per weight pass": mlp_bf16_g2.hip built a second time on the fp16 forms of the matrix instruction and of the
// re-pack conversion (v_mfma_f32_16x16x32_f16, v_cvt_pk_f16_f32: same rate, same registers, same LDS image size, same schedule).  fp16
// keeps 11 significant bits against bf16's 8: operand rounding 8x smaller; its range (65504) holds the network's hidden activations
// and weights with orders of magnitude to spare (tests/test_hip_f16.py).  Reference stage: models/base_architectures.py:40-61, 103-126.
#define G2_HALF 1
#include "mlp_bf16_g2.hip"
