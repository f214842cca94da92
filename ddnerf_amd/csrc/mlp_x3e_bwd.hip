// K2 "x3" backward-data with EXACT records (DDNERF_X3_WGRAD=exact): the x3 backward-data chain that writes every outgoing delta as blocked
// hi/lo words (see mlp_x3e_fwd_train.hip); round 2's kernel, frozen (mlp_mfma16_hilo.inc).  The step plan is mlp_x3_bwd.hip's.
#include "common.h"
#define M16_PLANES 2
#define M16_BWD
#define M16_SYM(x) ddnerf_mlp_x3e_##x
#define M16_KERNEL mlp_x3e_bwd16_kernel
#define M16_FEAT_T float
#define M16_PACK_KERNEL mlp_x3e_pack_t16_kernel
#define M16_PACK_SYM ddnerf_mlp_x3e_pack_t
#define M16_PACKED_BYTES_SYM ddnerf_mlp_x3e_packed_t_bytes
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

#include "mlp_mfma16_hilo.inc"
