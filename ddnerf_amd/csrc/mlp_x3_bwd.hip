// K2 "x3" backward-data on v_mfma_f32_16x16x32_bf16: ONE persistent kernel chains d(raw) back through all layers with exact
// hi/lo operand splits -- the body of mlp_x3_fwd.hip (mlp_mfma16.inc: transposed formulation, tiles chained through registers,
// four-buffer LDS stage ring fed by LDS-DMA, asm MFMAs with in-place accumulators) run over the transposed weights:
//     delta_l^T = (W_{l+1}^T delta_{l+1}^T) * relu'(h_l)
// Ten steps (K = rows of the incoming delta; 16-row blocks of the outgoing one):
//   d0: heads^T          K = 32  (the d(raw) tile)               -> d(dir hidden) 128 rows, masked
//   d1: [dir | alpha]^T  K = 160 (d(dir hidden) + the d(raw) tile) -> d(fc_feat out) 256 rows (fc_feat has no activation)
//   d2: fc_feat^T        K = 256                                 -> d(h7), masked
//   d3..d9: layers_xyz.{7..1}^T (layer 5: its hidden columns only) -> d(h6) .. d(h0), masked
// relu' comes from the sign words the training forward recorded (one 16-bit word per 32-row tile, lane half and sample; fetched
// two LDS stages ahead of the re-pack that applies them: x & -(bit)).  Every outgoing tile is also written to `deltas` as
// blocked hi/lo words (mlp_x3_common.h), the operand of mlp_x3_wgrad_packed.hip, from the same re-pack steps as in
// mlp_x3_fwd_train.hip; the d(raw) tile goes to rows 2432.. .  Tiles start from the inline constant 0 (no bias).
// Replaces the 32x32x16 kernel of round 1 (mlp_x3_train.hip): 2.1 -> see DESIGN.md.
#include "common.h"
#define M16_PLANES 2
#define M16_BWD
#define M16_SYM(x) ddnerf_mlp_x3_##x
#define M16_KERNEL mlp_x3_bwd16_kernel
#define M16_FEAT_T float
#define M16_PACK_KERNEL mlp_x3_pack_t16_kernel
#define M16_PACK_SYM ddnerf_mlp_x3_pack_t
#define M16_PACKED_BYTES_SYM ddnerf_mlp_x3_packed_t_bytes
#define NSTAGE 72
// stage -> (step, first block, blocks): K = 32 slices four to a stage, K = 160 slices three (34 KiB), K = 256 slices two
static constexpr int kStage[NSTAGE][3] = {
    {0, 0, 4}, {0, 4, 4},
    {1, 0, 3}, {1, 3, 3}, {1, 6, 3}, {1, 9, 3}, {1, 12, 2}, {1, 14, 2},
    {2, 0, 2}, {2, 2, 2}, {2, 4, 2}, {2, 6, 2}, {2, 8, 2}, {2, 10, 2}, {2, 12, 2}, {2, 14, 2},
    {3, 0, 2}, {3, 2, 2}, {3, 4, 2}, {3, 6, 2}, {3, 8, 2}, {3, 10, 2}, {3, 12, 2}, {3, 14, 2},
    {4, 0, 2}, {4, 2, 2}, {4, 4, 2}, {4, 6, 2}, {4, 8, 2}, {4, 10, 2}, {4, 12, 2}, {4, 14, 2},
    {5, 0, 2}, {5, 2, 2}, {5, 4, 2}, {5, 6, 2}, {5, 8, 2}, {5, 10, 2}, {5, 12, 2}, {5, 14, 2},
    {6, 0, 2}, {6, 2, 2}, {6, 4, 2}, {6, 6, 2}, {6, 8, 2}, {6, 10, 2}, {6, 12, 2}, {6, 14, 2},
    {7, 0, 2}, {7, 2, 2}, {7, 4, 2}, {7, 6, 2}, {7, 8, 2}, {7, 10, 2}, {7, 12, 2}, {7, 14, 2},
    {8, 0, 2}, {8, 2, 2}, {8, 4, 2}, {8, 6, 2}, {8, 8, 2}, {8, 10, 2}, {8, 12, 2}, {8, 14, 2},
    {9, 0, 2}, {9, 2, 2}, {9, 4, 2}, {9, 6, 2}, {9, 8, 2}, {9, 10, 2}, {9, 12, 2}, {9, 14, 2}};

#include "mlp_mfma16.inc"
