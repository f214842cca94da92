// K2 ("x3"): the whole 8x256 MLP (models/base_architectures.py:40-61, 103-126) as ONE kernel on the bf16 matrix
// cores at fp32-class accuracy.
//
// Every fp32 number splits exactly into hi + lo + r with hi = bf16(x), lo = bf16(x - hi), |r| <= 2^-17 |x|, so
//     w * a  =  hi_w hi_a  +  hi_w lo_a  +  lo_w hi_a  +  O(2^-16 |w a|):
// three v_mfma_f32_32x32x16_bf16 (fp32 accumulation) per 16 input features replace the eight v_mfma_f32_32x32x2_f32
// of the exact fp32 kernel -- 5.3x fewer matrix-pipe cycles.  Measured against an fp64 evaluation of the same network
// the outputs are off by ~1e-6 absolute (fp32 kernel: 6e-8; plain bf16: 6e-4): two orders of magnitude inside the
// 1e-4 RGB/depth parity bar, so this is the fast path of the fp32 tier, not a reduced-precision tier.
//
// Formulation as in mlp_bf16.hip: H_out^T[out, sample] = W[out, in] * H_in^T[in, sample]; A = W slices (hi and lo
// images) from LDS, B = the previous layer's activations (hi and lo files) in registers; the accumulator layout is the
// next layer's B layout in the permuted k order baked into the weight packing.  A wave owns 32 samples (the hi + lo
// files of this and the next layer are 4 x 64 registers: one wave per SIMD), 4 waves = 128 samples per workgroup
// share every LDS-staged weight byte; per k-step a wave issues 2 ds_read_b128 and 3 MFMAs (96 matrix-pipe cycles), so
// LDS reads sit at a third of the array's rate and the accumulator -> hi/lo re-pack (8 VALU ops per value pair) has
// 48 MFMAs per tile to hide behind.
#include "mlp_x3_common.h"

struct PlanX {
    int w_src[13];
    int b_src[13];
    int total_bytes;
};

static PlanX make_plan_x(int depth_head) {
    PlanX p;
    static const int nout[13] = {256, 256, 256, 256, 256, 256, 256, 256, 256, 1, 128, 3, 2};
    static const int nin[13] = {96, 256, 256, 256, 256, 352, 256, 256, 256, 256, 283, 128, 128};
    int off = 0;
    for (int l = 0; l < 13; ++l) {
        p.w_src[l] = off;
        off += nout[l] * nin[l];
        p.b_src[l] = off;
        off += nout[l];
        if (l == 11 && !depth_head) {
            p.w_src[12] = p.b_src[12] = -1;
            break;
        }
    }
    p.total_bytes = x3_total_bytes<X3FwdPlan>();
    return p;
}

DDN_EXPORT size_t ddnerf_mlp_x3_train_packed_bytes(int depth_head) { return (size_t)make_plan_x(depth_head).total_bytes; }

// same source mapping as the fp32 / bf16 kernels (see mlp_f32.hip)
__device__ __forceinline__ float srcw(const float *__restrict__ P, const PlanX &pl, int l, int o, int c) {
    if (l <= 8) return P[pl.w_src[l] + o * X3FwdPlan::K[l] + c];
    if (l == 9) {
        if (o < 128) return c < 283 ? P[pl.w_src[10] + o * 283 + c] : 0.0f;
        if (o == 128) return c < 256 ? P[pl.w_src[9] + c] : 0.0f;
        return 0.0f;
    }
    if (o < 3) return P[pl.w_src[11] + o * 128 + c];
    if ((o == 4 || o == 5) && pl.w_src[12] >= 0) return P[pl.w_src[12] + (o - 4) * 128 + c];
    return 0.0f;
}
__device__ __forceinline__ float srcb(const float *__restrict__ P, const PlanX &pl, int l, int o) {
    if (l <= 8) return P[pl.b_src[l] + o];
    if (l == 9) return o < 128 ? P[pl.b_src[10] + o] : (o == 128 ? P[pl.b_src[9]] : 0.0f);
    if (o < 3) return P[pl.b_src[11] + o];
    if ((o == 4 || o == 5) && pl.b_src[12] >= 0) return P[pl.b_src[12] + (o - 4)];
    return 0.0f;
}

__global__ void mlp_x3_pack_kernel(const float *__restrict__ P, PlanX pl, unsigned short *__restrict__ packed) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;  // one thread per 16-bit word of the packed buffer
    if (idx >= pl.total_bytes / 2) return;
    packed[idx] = x3_pack_word<X3FwdPlan>(idx, [&](int l, int o, int c) { return srcw(P, pl, l, o, c); },
                                          [&](int l, int o) { return srcb(P, pl, l, o); });
}

DDN_EXPORT int ddnerf_mlp_x3_train_pack(const float *params, int depth_head, void *packed, ddnerf_stream_t stream) {
    DDN_REQUIRE(params && packed, DDNERF_E_ARG);
    DDN_REQUIRE(ddn_aligned(packed, 16), DDNERF_E_ALIGN);
    PlanX pl = make_plan_x(depth_head);
    int threads = pl.total_bytes / 2;
    hipLaunchKernelGGL(mlp_x3_pack_kernel, dim3((threads + 255) / 256), dim3(256), 0, (hipStream_t)stream, params, pl,
                       (unsigned short *)packed);
    return ddn_launch_status();
}

// ---- fused forward ----------------------------------------------------------------------------------------
// MODE 0: inference (ddnerf_mlp_x3_forward).  MODE 1: the training forward, which also records every layer's output
// as a record of blocked hi/lo words (`acts`, mlp_x3_common.h; row map of mlp_f32_train.hip) and its sign bits (`bits`) for
// the backward pass.
template <bool DEPTH_HEAD, int MODE, int PFD>
__global__ __launch_bounds__(X3_WG_THREADS, 1) void mlp_x3_fwd_kernel(const float *__restrict__ feat,
                                                                      const char *__restrict__ packed,
                                                                      float *__restrict__ raw, float *__restrict__ acts,
                                                                      unsigned short *__restrict__ bits, long M, long ld) {
    __shared__ __attribute__((aligned(16))) char lds[2 * X3_STAGE_BYTES_MAX];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, h = lane >> 5;
    const long m = (long)blockIdx.x * X3_WG_SAMPLES + wave * 32 + j;  // < ld always in MODE 1 (ld is a multiple of 128)
    const float *frow = feat + (size_t)(m < M ? m : M - 1) * DDNERF_FEAT_LD;
    bf16x8 HAh[16], HAl[16], HBh[16], HBl[16], Xh[8], Xl[8];
    f32x16 keep;
#ifdef X3_NO_REPACK
    for (int r = 0; r < 16; ++r) keep[r] = 0.f;
#endif
    const char *wp = packed;
    x3_dma_stage(wp, lds, x3_stage_bytes<X3FwdPlan>(0, 0), wave, lane);
    // fp32 features, natural column order: position j of lane half h in k-step g is column 16g + 8(j>>2) + 4h + (j&3)
    auto load_x = [&](auto g0c, auto g1c, bool record) {  // feature groups [g0, g1) (re-fetched when needed again, not held)
        constexpr int g0 = decltype(g0c)::value, g1 = decltype(g1c)::value;
#pragma unroll
        for (int g = g0; g < g1; ++g) {
            const f32x4 a = *(const f32x4 *)(frow + 16 * g + 4 * h), b = *(const f32x4 *)(frow + 16 * g + 8 + 4 * h);
            split_quad(a, b, Xh[g], Xl[g]);
            if (MODE == 1 && record) {  // the input columns, transposed, are operands of the weight gradients too
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    __builtin_nontemporal_store(x3_word(a[c]), (unsigned *)acts + x3_rec_index(2432 + 16 * g + 4 * h + c, m));
                    __builtin_nontemporal_store(x3_word(b[c]), (unsigned *)acts + x3_rec_index(2432 + 16 * g + 8 + 4 * h + c, m));
                }
            }
        }
    };
    load_x(std::integral_constant<int, 0>{}, std::integral_constant<int, 6>{}, true);  // xyz: dead again after layer 0
    auto io = [&](int row0) { return x3_tile_io(acts, bits, nullptr, ld, m, h, row0); };
    using PL = X3FwdPlan;
    // parity of the LDS buffer holding a layer's first stage: layers 0..8 have an even number of stages (2, 4, 4, 4, 4, 8,
    // 4, 4, 4) so layers 0..9 start in buffer 0; the dir layer has 5, so the heads start in buffer 1
    x3_layer<PL, 0, 0, 1, 0, 8, true, MODE, PFD, true, true>(wp, lds, HAh, HAl, Xh, Xl, HAh, HAl, keep, wave, lane, io(0));
    x3_layer<PL, 1, 1, 2, 0, 8, false, MODE, PFD, true>(wp, lds, HAh, HAl, Xh, Xl, HBh, HBl, keep, wave, lane, io(256));
    x3_layer<PL, 2, 1, 3, 0, 8, true, MODE, PFD, true>(wp, lds, HBh, HBl, Xh, Xl, HAh, HAl, keep, wave, lane, io(512));
    x3_layer<PL, 3, 1, 4, 0, 8, false, MODE, PFD, true>(wp, lds, HAh, HAl, Xh, Xl, HBh, HBl, keep, wave, lane, io(768));
    x3_layer<PL, 4, 1, 5, 0, 8, true, MODE, PFD, true>(wp, lds, HBh, HBl, Xh, Xl, HAh, HAl, keep, wave, lane, io(1024));
    load_x(std::integral_constant<int, 0>{}, std::integral_constant<int, 6>{}, false);
    x3_layer<PL, 5, 2, 6, 0, 8, false, MODE, PFD, true>(wp, lds, HAh, HAl, Xh, Xl, HBh, HBl, keep, wave, lane, io(1280));  // cat(xyz, h)
    x3_layer<PL, 6, 1, 7, 0, 8, true, MODE, PFD, true>(wp, lds, HBh, HBl, Xh, Xl, HAh, HAl, keep, wave, lane, io(1536));
    x3_layer<PL, 7, 1, 8, 0, 8, false, MODE, PFD, true>(wp, lds, HAh, HAl, Xh, Xl, HBh, HBl, keep, wave, lane, io(1792));
    x3_layer<PL, 8, 1, 9, 0, 8, true, MODE, PFD, false>(wp, lds, HBh, HBl, Xh, Xl, HAh, HAl, keep, wave, lane, io(2048));  // fc_feat: no activation
    load_x(std::integral_constant<int, 6>{}, std::integral_constant<int, 8>{}, true);                                     // view-dir columns
    x3_layer<PL, 9, 3, 10, 0, 4, false, MODE, PFD, true>(wp, lds, HAh, HAl, Xh, Xl, HBh, HBl, keep, wave, lane, io(2304));  // dir layer + alpha row
    const float alpha = keep[0];  // row 128 = block 4, register 0, lane half 0
    x3_layer<PL, 10, 4, -1, 1, 0, true, MODE, PFD, false>(wp, lds, HBh, HBl, Xh, Xl, HAh, HAl, keep, wave, lane, io(0));  // heads

    if (m < M) {
        if (DEPTH_HEAD) {
            float *op = raw + (size_t)m * 6;
            if (h == 0) {
                *(float2 *)(op) = make_float2(keep[0], keep[1]);
                *(float2 *)(op + 2) = make_float2(keep[2], alpha);
            } else {
                *(float2 *)(op + 4) = make_float2(keep[0], keep[1]);  // rows 4, 5 = raw mu, raw sigma
            }
        } else if (h == 0) {
            *(f32x4 *)(raw + (size_t)m * 4) = f32x4{keep[0], keep[1], keep[2], alpha};
        }
    }
}

#ifndef X3_PFD
#define X3_PFD 6  // weight pieces in flight per wave (inference)
#endif
#ifndef X3_PFD_TRAIN
#define X3_PFD_TRAIN 14
#endif
// vmcnt retires in order: the staging ring must also cover the acknowledgement of the activation stores

// The training forward: also writes acts (2560 rows x ld samples, blocked hi/lo words; row map of mlp_f32_train.hip) and
// bits [160, ld] (u16: word (tile * 2 + lane half) of a sample holds the signs of that lane's 16 values of the tile).
DDN_EXPORT int ddnerf_mlp_x3_forward_train(const float *feat, const void *packed, int depth_head, float *raw, float *acts,
                                           void *bits, long M, long ld, ddnerf_stream_t stream) {
    DDN_REQUIRE(feat && packed && raw && acts && bits, DDNERF_E_ARG);
    DDN_REQUIRE(M > 0, DDNERF_E_ARG);
    DDN_REQUIRE(ld >= M && ld % 128 == 0, DDNERF_E_RANGE);
    DDN_REQUIRE(ddn_aligned(feat, 16) && ddn_aligned(packed, 16) && ddn_aligned(raw, 16), DDNERF_E_ALIGN);
    dim3 grid((unsigned)((M + X3_WG_SAMPLES - 1) / X3_WG_SAMPLES));
    if (depth_head)
        hipLaunchKernelGGL((mlp_x3_fwd_kernel<true, 1, X3_PFD_TRAIN>), grid, dim3(X3_WG_THREADS), 0, (hipStream_t)stream,
                           feat, (const char *)packed, raw, acts, (unsigned short *)bits, M, ld);
    else
        hipLaunchKernelGGL((mlp_x3_fwd_kernel<false, 1, X3_PFD_TRAIN>), grid, dim3(X3_WG_THREADS), 0, (hipStream_t)stream,
                           feat, (const char *)packed, raw, acts, (unsigned short *)bits, M, ld);
    return ddn_launch_status();
}
