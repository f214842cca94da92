// K2 "x3" training forward with the view-direction columns once per RAY (round 5): mlp_x3_fwd_train.hip's kernel -- same body, same weight
// image, same records and sign words -- whose view-direction group of a sample comes from the row of its ray in the per-ray table [n,32] of
// ddnerf_encode_rays (as mlp_x3_fwd_rays.hip's inference kernel takes it) instead of columns 96..127 of the sample's own feature row.
// Entry point ddnerf_mlp_x3_forward_train_rays; outputs and records bit for bit ddnerf_mlp_x3_forward_train's on full rows.
#include "common.h"
#define M16_PLANES 2
#define M16_TRAIN
#define M16_RAY_DIRS
#define M16_NO_PACK
#define M16_SYM(x) ddnerf_mlp_x3tr_##x
#define M16_KERNEL mlp_x3_fwd16_train_rays_kernel
#define M16_FEAT_T float
#include "mlp_x3_stages.h"

#include "mlp_mfma16.inc"

DDN_EXPORT int ddnerf_mlp_x3_forward_train_rays(const float *feat, const float *dirs, int S, const void *packed, int depth_head, float *raw,
                                                float *acts, void *bits, long M, long ld, ddnerf_stream_t stream) {
    return ddnerf_mlp_x3tr_forward_train_rays(feat, dirs, S, packed, depth_head, raw, acts, bits, M, ld, stream);
}
