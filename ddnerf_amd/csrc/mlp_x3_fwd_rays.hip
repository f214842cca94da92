// K2 "x3" (inference) with the view-direction columns once per RAY: mlp_x3_fwd.hip's kernel -- the same body, mlp_mfma16.inc, the same weight
// image (ddnerf_mlp_x3_pack) -- whose view-direction group of a sample is fetched from the row of its ray in a per-ray table [n,32]
// (ddnerf_encode_rays: the reference encodes a ray's direction once and broadcasts it, models/models.py:128-133) instead of columns 96..127 of
// the sample's own feature row.  Entry point ddnerf_mlp_x3_forward_rays; outputs bit for bit ddnerf_mlp_x3_forward's on full rows.
#include "common.h"
#define M16_PLANES 2
#define M16_RAY_DIRS
#define M16_SYM(x) ddnerf_mlp_x3r_##x
#define M16_KERNEL mlp_x3_fwd16_rays_kernel
#define M16_FEAT_T float
#define M16_PACK_KERNEL mlp_x3r_pack16_kernel
#include "mlp_x3_stages.h"
#include "mlp_mfma16.inc"

// (the public name; the pack / packed_bytes twins of this translation unit are not exported through the header: the image is ddnerf_mlp_x3_pack's)
DDN_EXPORT int ddnerf_mlp_x3_forward_rays(const float *feat, const float *dirs, int S, const void *packed, int depth_head, float *raw, long M,
                                          ddnerf_stream_t stream) {
    return ddnerf_mlp_x3r_forward_rays(feat, dirs, S, packed, depth_head, raw, M, stream);
}
