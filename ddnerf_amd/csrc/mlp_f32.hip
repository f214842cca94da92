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
// is a linear copy.  Slices are double-buffered: slice s+1 is fetched into registers while slice s feeds the
// MFMAs, written to the other LDS buffer afterwards, one barrier per slice.
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

// One layer: NB slices.  `wp` walks the packed buffer (slices are stored in consumption order); while slice s
// is multiplied, slice s+1 (possibly the next layer's first one, NEXT_K wide; NEXT_K = 0: none) is fetched.
// PAR = parity of the LDS buffer that holds this layer's first slice.
template <int KIND, int K, int NB, int NEXT_K, int PAR>
__device__ __forceinline__ void layer(const float *__restrict__ &wp, const float *__restrict__ bias, float *lds,
                                      const f32x16 (&Breg)[12], f32x16 (&out)[8], int tid, int lane) {
    constexpr int N4 = slice_floats(K) / 4;  // float4 pieces per slice of this layer
    constexpr int NEXT_N4 = NEXT_K > 0 ? slice_floats(NEXT_K) / 4 : 0;
    const int h = lane >> 5;
    f32x16 bcur = bias_tile(bias, h);  // one exposed fetch per layer; every later tile's bias is fetched a slice ahead
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const float *cur = lds + ((PAR + b) & 1) * MAX_SLICE_FLOATS;
        float *nxt = lds + ((PAR + b + 1) & 1) * MAX_SLICE_FLOATS;
        wp += 4 * N4;  // now points at the slice after the current one
        f32x16 bnext;
        auto init = [&](f32x16 &a) {
            a = bcur;
            if (b + 1 < NB) bnext = bias_tile(bias + 32 * (b + 1), h);
        };
        auto post = [](f32x16 &) {};
        if (b + 1 < NB) slice_step_hooks<KIND, K, N4>(wp, cur, nxt, Breg, out[b], tid, lane, init, [](int, int) {}, post);
        else slice_step_hooks<KIND, K, NEXT_N4>(wp, cur, nxt, Breg, out[b], tid, lane, init, [](int, int) {}, post);
        if (b + 1 < NB) bcur = bnext;
    }
}

// load feature columns [32*b0, 32*(b0+nb)) of this lane's sample into Breg[dst..] (B layout, see header)
template <int DST, int B0, int NBLK>
__device__ __forceinline__ void load_features(const float *__restrict__ frow, int h, f32x16 (&Breg)[12]) {
#pragma unroll
    for (int b = 0; b < NBLK; ++b) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 v = *(const f32x4 *)(frow + 32 * (B0 + b) + 8 * g + 4 * h);
            Breg[DST + b][4 * g + 0] = v.x;
            Breg[DST + b][4 * g + 1] = v.y;
            Breg[DST + b][4 * g + 2] = v.z;
            Breg[DST + b][4 * g + 3] = v.w;
        }
    }
}

template <bool DEPTH>
__global__ __launch_bounds__(256, 1) void mlp_f32_fwd_kernel(const float *__restrict__ feat,
                                                             const float *__restrict__ packed, PackPlan pl,
                                                             float *__restrict__ raw, long M) {
    __shared__ __attribute__((aligned(16))) float lds[2 * MAX_SLICE_FLOATS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, h = lane >> 5;
    const long m = (long)blockIdx.x * 128 + wave * 32 + j;
    const bool valid = m < M;
    const float *frow = feat + (size_t)(valid ? m : M - 1) * DDNERF_FEAT_LD;

    f32x16 Breg[12];
    f32x16 out[8];

    // stage slice 0 synchronously, fetch the sample's 128 features into B layout meanwhile
    const float *wp = packed;
    {
        constexpr int ROUNDS = slice_floats(96) / 1024;
        f32x4 pf[ROUNDS];
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) pf[r] = *(const f32x4 *)(wp + 4 * (size_t)(r * 256 + tid));
        load_features<8, 0, 3>(frow, h, Breg);  // xyz features; dead again after layer 0
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) *(f32x4 *)(lds + 4 * (r * 256 + tid)) = pf[r];
    }
    __syncthreads();

    const float *bias = packed;  // + pl.bias_off[l]
    // layer 0: 96 -> 256, ReLU                                             base_architectures.py:42-43
    layer<0, 96, 8, 256, 0>(wp, bias + pl.bias_off[0], lds, Breg, out, tid, lane);
#pragma unroll
    for (int b = 0; b < 8; ++b) Breg[b] = relu16(out[b]);
    // layers 1..4: 256 -> 256, ReLU                                        :44-49
    for (int l = 1; l <= 3; ++l) {
        layer<1, 256, 8, 256, 0>(wp, bias + pl.bias_off[l], lds, Breg, out, tid, lane);
#pragma unroll
        for (int b = 0; b < 8; ++b) Breg[b] = relu16(out[b]);
    }
    layer<1, 256, 8, 352, 0>(wp, bias + pl.bias_off[4], lds, Breg, out, tid, lane);
#pragma unroll
    for (int b = 0; b < 8; ++b) Breg[b] = relu16(out[b]);
    // layer 5: cat(xyz, x) 352 -> 256, ReLU                                :45-46
    load_features<8, 0, 3>(frow, h, Breg);  // re-fetched (L2) instead of held in 48 registers across layers 1-4
    __builtin_amdgcn_sched_barrier(0);
    layer<2, 352, 8, 256, 0>(wp, bias + pl.bias_off[5], lds, Breg, out, tid, lane);
#pragma unroll
    for (int b = 0; b < 8; ++b) Breg[b] = relu16(out[b]);
    // layers 6, 7 (ReLU) and fc_feat (no activation)                       :47-50
    for (int l = 6; l <= 7; ++l) {
        layer<1, 256, 8, 256, 0>(wp, bias + pl.bias_off[l], lds, Breg, out, tid, lane);
#pragma unroll
        for (int b = 0; b < 8; ++b) Breg[b] = relu16(out[b]);
    }
    layer<1, 256, 8, 288, 0>(wp, bias + pl.bias_off[8], lds, Breg, out, tid, lane);
#pragma unroll
    for (int b = 0; b < 8; ++b) Breg[b] = out[b];
    // layers_dir.0 on cat(feat, dirs) + fc_alpha on feat: 288 -> 160       :51-56
    load_features<11, 3, 1>(frow, h, Breg);  // view-dir columns 96..127
    __builtin_amdgcn_sched_barrier(0);
    layer<3, 288, 5, 128, 0>(wp, bias + pl.bias_off[9], lds, Breg, out, tid, lane);
    const float alpha = out[4][0];  // row 128 = block 4, register 0, lane half 0
#pragma unroll
    for (int b = 0; b < 4; ++b) Breg[b] = relu16(out[b]);
    // fc_rgb (+ fc_mu_sigma): 128 -> 32 rows                               :60 / :123-124
    layer<4, 128, 1, 0, 1>(wp, bias + pl.bias_off[10], lds, Breg, out, tid, lane);

    if (valid) {
        if (DEPTH) {
            float *o = raw + (size_t)m * 6;
            if (h == 0) {
                *(float2 *)(o) = make_float2(out[0][0], out[0][1]);
                *(float2 *)(o + 2) = make_float2(out[0][2], alpha);
            } else {
                *(float2 *)(o + 4) = make_float2(out[0][0], out[0][1]);  // rows 4, 5 = raw mu, raw sigma
            }
        } else if (h == 0) {
            *(f32x4 *)(raw + (size_t)m * 4) = f32x4{out[0][0], out[0][1], out[0][2], alpha};
        }
    }
}

DDN_EXPORT int ddnerf_mlp_f32_forward(const float *feat, const float *packed, int depth_head, float *raw, long M,
                                      ddnerf_stream_t stream) {
    DDN_REQUIRE(feat && packed && raw, DDNERF_E_ARG);
    DDN_REQUIRE(M > 0, DDNERF_E_ARG);
    DDN_REQUIRE(ddn_aligned(feat, 16) && ddn_aligned(packed, 16) && ddn_aligned(raw, 16), DDNERF_E_ALIGN);
    PackPlan pl = make_plan(depth_head);
    dim3 grid((unsigned)((M + 127) / 128));
    if (depth_head)
        hipLaunchKernelGGL(mlp_f32_fwd_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, feat, packed, pl, raw, M);
    else
        hipLaunchKernelGGL(mlp_f32_fwd_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, feat, packed, pl, raw, M);
    return ddn_launch_status();
}
