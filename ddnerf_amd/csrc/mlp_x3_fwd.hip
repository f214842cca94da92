// K2 "x3" (inference): the 8x256 MLP at fp32-class accuracy on the bf16 matrix cores, as ONE persistent kernel on
// v_mfma_f32_16x16x32_bf16 -- the structure of mlp_bf16.hip (see there: transposed formulation, activations in registers,
// four-buffer LDS stage ring, buffer-load weight stream, asm MFMAs with in-place accumulators) with every operand as two bf16
// planes.  An fp32 value splits exactly into hi = bf16(x), lo = bf16(x - hi) and a residual below 2^-17 |x|, and
//     w * a ~ hi_w hi_a + hi_w lo_a + lo_w hi_a      (three MFMAs per product, fp32 accumulation, small terms first)
// which is ~1e-6 from an fp64 evaluation (exact fp32 kernel 6e-8, plain bf16 6e-4) and is held to the fp32 parity bar (1e-4).
// Per wave: two 16-sample column blocks x (hi, lo) = the same four B-operand columns and the same register budget as the
// bf16 kernel; per k-step two A fragments (hi, lo rows of the LDS slice) feed six MFMAs.  A tile is 128 samples.  Feature
// rows are the fp32 [M,128] matrix in natural column order; they are split into planes while they move to the accumulator
// half.  The re-pack of a finished tile is ten small steps (ReLU, hi, hi as float, x - hi, lo, home hi, home lo) dealt over
// the MFMA gaps of the next block.
// Replaces the 32x32x16 forward of round 1; mlp_x3_fwd_train.hip is this kernel plus the training records, mlp_x3_bwd.hip the
// backward-data pass on the same body (transposed steps).
#include "common.h"
#define M16_PLANES 2
#define M16_SYM(x) ddnerf_mlp_x3_##x
#define M16_KERNEL mlp_x3_fwd16_kernel
#define M16_FEAT_T float
#define M16_PACK_KERNEL mlp_x3_pack16_kernel
#include "mlp_x3_stages.h"
#include "mlp_mfma16.inc"
