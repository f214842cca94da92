// K2 (fp32): the whole 8x256 MLP (MipNeRFModel.forward / DepthMipNeRFModel.forward,
// models/base_architectures.py:40-61, 103-126) as ONE kernel on the fp32-input matrix cores
// (v_mfma_f32_32x32x2_f32: bit-for-bit a k-ordered fmaf chain, so the fp32 parity bar holds).
//
// Formulation: every layer is computed TRANSPOSED,  H_out^T[out, sample] = W[out, in] * H_in^T[in, sample].
//   * A operand  = a 32-row block of W (out-features x in-features), staged through LDS and shared by the
//                  four waves of the workgroup;
//   * B operand  = the previous layer's output, which in the MFMA accumulator layout already has the sample
//                  on the lane and the feature on the register -- exactly what the next MFMA wants as B.
// So activations NEVER leave the register file: no LDS round trip, no HBM round trip between the 12 layers.
// A wave owns 32 samples (one 32-wide column block): 8 accumulator tiles (256 features) in, 8 out.
// A workgroup = 4 waves = 128 samples, one workgroup per CU (the kernel wants the full 512-register file).
//
// MFMA lane maps (32x32x2 f32): A[i = lane&31][k = lane>>5], B[k = lane>>5][j = lane&31],
// D[row = (r&3) + 8*(r>>2) + 4*(lane>>5)][col = lane&31], r = 0..15.
// Feeding D register r of block b' back as B means: k-slot h (= lane>>5) carries in-feature
// 32b' + (r&3) + 8(r>>2) + 4h.  For four consecutive r (r = 4g..4g+3) lane half h therefore needs the four
// CONSECUTIVE in-features 32b' + 8g + 4h .. +3 of W's row: one ds_read_b128 of the natural [out][in] layout.
//
// Weights are repacked once per update into the exact LDS image of each 32-row slice (row stride K+4 floats:
// the 16-byte pad makes the b128 fragment reads bank-conflict free), stored in consumption order, so staging
// is a linear copy; the 32 bias values of the slice's rows follow its last row inside the same image (the layers' bias blocks at the
// end of the buffer stay: the training kernels read those).  Slices are double-buffered: slice s+1 is fetched into registers while slice s feeds the
// MFMAs, written to the other LDS buffer behind its later MFMAs, one barrier per slice (in front of its last four MFMAs).
#include "mlp_f32_common.h"

// ---- layer schedule -------------------------------------------------------------------------------
// kind: 0 first layer (K=96 from xyz features)   1 hidden (K=256)   2 skip layer (K=352 = xyz96 + hidden256)
//       3 dir+alpha layer (K=288 = feat256 + dir27 + 5 zero, 160 rows = 128 dir + alpha + 31 zero)
//       4 heads (K=128, 32 rows: rgb 0..2, mu 4, sigma 5)
#define NLAYERS 11
static constexpr int kLayerK[NLAYERS] = {96, 256, 256, 256, 256, 352, 256, 256, 256, 288, 128};
static constexpr int kLayerNB[NLAYERS] = {8, 8, 8, 8, 8, 8, 8, 8, 8, 5, 1};

struct PackPlan {
    int slice_off[NLAYERS];  // float offset of the layer's first slice in the packed buffer
    int bias_off[NLAYERS];   // float offset of the layer's bias block (NB*32 floats)
    int w_src[13];           // float offsets of the 13 weight matrices in the flat parameter buffer
    int b_src[13];           //   "      of the 13 bias vectors
    int total;               // floats in the packed buffer
};

static PackPlan make_plan(int depth_head) {
    PackPlan p;
    static const int nout[13] = {256, 256, 256, 256, 256, 256, 256, 256, 256, 1, 128, 3, 2};
    static const int nin[13] = {96, 256, 256, 256, 256, 352, 256, 256, 256, 256, 283, 128, 128};
    int off = 0;
    for (int l = 0; l < 13; ++l) {
        p.w_src[l] = off;
        off += nout[l] * nin[l];
        p.b_src[l] = off;
        off += nout[l];
        if (l == 11 && !depth_head) {  // no fc_mu_sigma in MipNeRFModel
            p.w_src[12] = p.b_src[12] = -1;
            break;
        }
    }
    off = 0;
    for (int l = 0; l < NLAYERS; ++l) {
        p.slice_off[l] = off;
        off += kLayerNB[l] * slice_floats(kLayerK[l]);
    }
    for (int l = 0; l < NLAYERS; ++l) {
        p.bias_off[l] = off;
        off += kLayerNB[l] * 32;
    }
    p.total = off;
    return p;
}

DDN_EXPORT size_t ddnerf_mlp_f32_packed_floats(int depth_head) { return (size_t)make_plan(depth_head).total; }

// value of packed layer `l`, out-row `o`, in-column `c` (c < K), read from the flat parameter buffer
__device__ __forceinline__ float src_weight(const float *__restrict__ P, const PackPlan &pl, int l, int o, int c) {
    if (l <= 8) return P[pl.w_src[l] + o * kLayerK[l] + c];  // layers_xyz.0..7, fc_feat: K == in_features
    if (l == 9) {
        if (o < 128) return c < 283 ? P[pl.w_src[10] + o * 283 + c] : 0.0f;  // layers_dir.0 on cat(feat, dirs)
        if (o == 128) return c < 256 ? P[pl.w_src[9] + c] : 0.0f;            // fc_alpha on feat
        return 0.0f;
    }
    if (o < 3) return P[pl.w_src[11] + o * 128 + c];                                     // fc_rgb
    if ((o == 4 || o == 5) && pl.w_src[12] >= 0) return P[pl.w_src[12] + (o - 4) * 128 + c];  // fc_mu_sigma
    return 0.0f;
}
__device__ __forceinline__ float src_bias(const float *__restrict__ P, const PackPlan &pl, int l, int o) {
    if (l <= 8) return P[pl.b_src[l] + o];
    if (l == 9) return o < 128 ? P[pl.b_src[10] + o] : (o == 128 ? P[pl.b_src[9]] : 0.0f);
    if (o < 3) return P[pl.b_src[11] + o];
    if ((o == 4 || o == 5) && pl.b_src[12] >= 0) return P[pl.b_src[12] + (o - 4)];
    return 0.0f;
}

__global__ void mlp_f32_pack_kernel(const float *__restrict__ P, PackPlan pl, float *__restrict__ packed) {
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= pl.total) return;
    int l;
    if (idx >= pl.bias_off[0]) {
        for (l = NLAYERS - 1; l > 0; --l)
            if (idx >= pl.bias_off[l]) break;
        packed[idx] = src_bias(P, pl, l, idx - pl.bias_off[l]);
        return;
    }
    for (l = NLAYERS - 1; l > 0; --l)
        if (idx >= pl.slice_off[l]) break;
    const int ld = kLayerK[l] + 4, local = idx - pl.slice_off[l];
    const int sl = local / slice_floats(kLayerK[l]), within = local % slice_floats(kLayerK[l]);
    float v = 0.0f;
    if (within < 32 * ld) {
        const int o = 32 * sl + within / ld, c = within % ld;
        if (c < kLayerK[l]) v = src_weight(P, pl, l, o, c);
    } else if (within < 32 * ld + 32) {
        v = src_bias(P, pl, l, 32 * sl + within - 32 * ld);  // the slice's bias tile rides in its image (round 4): it reaches LDS with the rows
    }
    packed[idx] = v;
}

DDN_EXPORT int ddnerf_mlp_f32_pack(const float *params, int depth_head, float *packed, ddnerf_stream_t stream) {
    DDN_REQUIRE(params && packed, DDNERF_E_ARG);
    DDN_REQUIRE(ddn_aligned(packed, 16), DDNERF_E_ALIGN);
    PackPlan pl = make_plan(depth_head);
    hipLaunchKernelGGL(mlp_f32_pack_kernel, dim3((pl.total + 255) / 256), dim3(256), 0, (hipStream_t)stream, params, pl,
                       packed);
    return ddn_launch_status();
}

#include "mlp_f32_fwd.inc"

template <bool DEPTH>
__global__ __launch_bounds__(256, 1) void mlp_f32_fwd_kernel(const float *__restrict__ feat, const float *__restrict__ packed, PackPlan pl,
                                                             float *__restrict__ raw, long M) {
    __shared__ __attribute__((aligned(16))) float lds[F32_FWD_LDS_FLOATS];
    mlp_f32_forward_tiles<DEPTH>(lds, feat, packed, 4u * (unsigned)pl.bias_off[0], raw, M, NoRecord{});
}
// the same with the view-direction columns from a per-ray table (ddnerf_mlp_f32_forward_rays)
template <bool DEPTH>
__global__ __launch_bounds__(256, 1) void mlp_f32_fwd_rays_kernel(const float *__restrict__ feat, const float *__restrict__ dirs, unsigned magic,
                                                                  const float *__restrict__ packed, PackPlan pl, float *__restrict__ raw, long M) {
    __shared__ __attribute__((aligned(16))) float lds[F32_FWD_LDS_FLOATS];
    mlp_f32_forward_tiles<DEPTH>(lds, feat, packed, 4u * (unsigned)pl.bias_off[0], raw, M, NoRecord{}, F32Dirs{dirs, magic});
}

#ifdef F32_STAMP_TILE
DDN_EXPORT int ddnerf_debug_f32_tile_stamps(unsigned long long *host_out) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_f32_tile), sizeof(g_f32_tile));
}
#endif
#ifdef F32_STAMP
DDN_EXPORT int ddnerf_debug_f32_stamps(unsigned long long *host_out) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_f32_stamps), sizeof(g_f32_stamps));
}
#endif

DDN_EXPORT int ddnerf_mlp_f32_forward(const float *feat, const float *packed, int depth_head, float *raw, long M,
                                      ddnerf_stream_t stream) {
    DDN_REQUIRE(feat && packed && raw, DDNERF_E_ARG);
    DDN_REQUIRE(M > 0, DDNERF_E_ARG);
    DDN_REQUIRE(ddn_aligned(feat, 16) && ddn_aligned(packed, 16) && ddn_aligned(raw, 16), DDNERF_E_ALIGN);
    PackPlan pl = make_plan(depth_head);
    const long ntiles = (M + 127) / 128, cus = ddn_cu_count();
    dim3 grid((unsigned)(ntiles < cus ? ntiles : cus));   // persistent: the kernel needs a CU's whole register file and most of its LDS
    if (depth_head)
        hipLaunchKernelGGL(mlp_f32_fwd_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, feat, packed, pl, raw, M);
    else
        hipLaunchKernelGGL(mlp_f32_fwd_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, feat, packed, pl, raw, M);
    return ddn_launch_status();
}

// feat [n*S,128] with columns 96..127 NOT read; dirs [n,32]: the rays' view-direction columns (ddnerf_encode_rays).  Same outputs as
// ddnerf_mlp_f32_forward on rows that carry the columns.  n * S^2 < 2^32 (the ray of a sample comes from a 32-bit multiply-high).
DDN_EXPORT int ddnerf_mlp_f32_forward_rays(const float *feat, const float *dirs, int S, const float *packed, int depth_head, float *raw, long M,
                                           ddnerf_stream_t stream) {
    DDN_REQUIRE(feat && dirs && packed && raw, DDNERF_E_ARG);
    DDN_REQUIRE(M > 0 && S > 0 && M % S == 0, DDNERF_E_ARG);
    DDN_REQUIRE((unsigned long long)M * (unsigned long long)S < (1ull << 32), DDNERF_E_RANGE);
    DDN_REQUIRE(ddn_aligned(feat, 16) && ddn_aligned(dirs, 16) && ddn_aligned(packed, 16) && ddn_aligned(raw, 16), DDNERF_E_ALIGN);
    PackPlan pl = make_plan(depth_head);
    const long ntiles = (M + 127) / 128, cus = ddn_cu_count();
    dim3 grid((unsigned)(ntiles < cus ? ntiles : cus));
    const unsigned magic = (unsigned)(((1ull << 32) + (unsigned)S - 1) / (unsigned)S);   // (S = 1: 2^32 does not fit -- the rows ARE the rays then)
    DDN_REQUIRE(S > 1, DDNERF_E_RANGE);
    if (depth_head)
        hipLaunchKernelGGL(mlp_f32_fwd_rays_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, feat, dirs, magic, packed, pl, raw, M);
    else
        hipLaunchKernelGGL(mlp_f32_fwd_rays_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, feat, dirs, magic, packed, pl, raw, M);
    return ddn_launch_status();
}
