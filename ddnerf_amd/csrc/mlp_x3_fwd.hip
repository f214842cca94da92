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
// Replaces the 32x32x16 forward of round 1 (mlp_x3_train.hip keeps that formulation for the training passes, which need the
// transposed activation stores).
#include "common.h"
#define M16_PLANES 2
#define M16_SYM(x) ddnerf_mlp_x3_##x
#define M16_KERNEL mlp_x3_fwd16_kernel
#define M16_FEAT_T float
#define M16_PACK_KERNEL mlp_x3_pack16_kernel
#define NSTAGE 88
// stage -> (layer, first block, blocks); the last stage also carries the heads block.  A slice holds two planes: two K = 256
// slices, one K = 352 / 288 slice, up to three K = 96 slices fit a 36-KiB buffer.
static constexpr int kStage[NSTAGE][3] = {
    {0, 0, 3}, {0, 3, 3}, {0, 6, 2}, {0, 8, 2}, {0, 10, 2}, {0, 12, 2}, {0, 14, 2},
    {1, 0, 2}, {1, 2, 2}, {1, 4, 2}, {1, 6, 2}, {1, 8, 2}, {1, 10, 2}, {1, 12, 2}, {1, 14, 2},
    {2, 0, 2}, {2, 2, 2}, {2, 4, 2}, {2, 6, 2}, {2, 8, 2}, {2, 10, 2}, {2, 12, 2}, {2, 14, 2},
    {3, 0, 2}, {3, 2, 2}, {3, 4, 2}, {3, 6, 2}, {3, 8, 2}, {3, 10, 2}, {3, 12, 2}, {3, 14, 2},
    {4, 0, 2}, {4, 2, 2}, {4, 4, 2}, {4, 6, 2}, {4, 8, 2}, {4, 10, 2}, {4, 12, 2}, {4, 14, 2},
    {5, 0, 1}, {5, 1, 1}, {5, 2, 1}, {5, 3, 1}, {5, 4, 1}, {5, 5, 1}, {5, 6, 1}, {5, 7, 1},
    {5, 8, 1}, {5, 9, 1}, {5, 10, 1}, {5, 11, 1}, {5, 12, 1}, {5, 13, 1}, {5, 14, 1}, {5, 15, 1},
    {6, 0, 2}, {6, 2, 2}, {6, 4, 2}, {6, 6, 2}, {6, 8, 2}, {6, 10, 2}, {6, 12, 2}, {6, 14, 2},
    {7, 0, 2}, {7, 2, 2}, {7, 4, 2}, {7, 6, 2}, {7, 8, 2}, {7, 10, 2}, {7, 12, 2}, {7, 14, 2},
    {8, 0, 2}, {8, 2, 2}, {8, 4, 2}, {8, 6, 2}, {8, 8, 2}, {8, 10, 2}, {8, 12, 2}, {8, 14, 2},
    {9, 0, 1}, {9, 1, 1}, {9, 2, 1}, {9, 3, 1}, {9, 4, 1}, {9, 5, 1}, {9, 6, 1}, {9, 7, 1}, {9, 8, 1}};

#include "mlp_mfma16.inc"
