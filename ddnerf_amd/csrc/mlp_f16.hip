// K2 (fp16), one group per wave: mlp_bf16.hip built a second time on the fp16 forms of the matrix instruction and of the re-pack
// conversion (see mlp_f16_g2.hip, the two-group build the entry point ddnerf_mlp_f16_forward runs at large launches; the two produce
// the same bits for the same sample).  Reference stage: models/base_architectures.py:40-61, 103-126.
#define M16_HALF 1
#include "mlp_bf16.hip"
