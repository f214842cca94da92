// K2 ("x3") backward-data: ONE kernel chains d(raw) back through all layers on the bf16 matrix cores with exact hi/lo
// operand splits (mlp_x3_common.h), mirroring mlp_f32_train.hip's fp32 pass:
//     delta_l^T = (W_{l+1}^T delta_{l+1}^T) * relu'(h_l)
// The accumulator tile of one step is re-packed into the hi/lo B files of the next, deltas never leave registers between
// layers; every delta tile is also stored into `deltas` (blocked hi/lo words, mlp_x3_common.h: the operand of the weight-gradient
// kernels).  relu' comes from the sign words the training forward recorded (2 bytes per lane per tile instead of the
// 64-byte fp32 tile: the pass then writes 5.4 GB per fine launch and reads almost nothing).
// Steps (K = rows of the incoming delta, NB = 32-row blocks of the outgoing one), as in mlp_f32_train.hip:
//   d0: heads^T          K=32  (the d(raw) tile)            -> d(dir hidden) 128 rows, masked
//   d1: [dir | alpha]^T  K=160 (d(dir hidden) + d(raw))     -> d(feat)       256 rows (fc_feat has no activation)
//   d2: fc_feat^T        K=256                              -> d(h7), masked
//   d3..d9: layers_xyz.{7..1}^T (layer 5: its hidden columns only)           -> d(h6) .. d(h0), masked
#include "mlp_x3_common.h"

#define ROW_X 2432
#define ROW_FEAT 2048
#define ROW_DIR 2304

struct X3BwdPlan {
    static constexpr int NL = 10;
    static constexpr int K[10] = {32, 160, 256, 256, 256, 256, 256, 256, 256, 256};
    static constexpr int NB[10] = {4, 8, 8, 8, 8, 8, 8, 8, 8, 8};
    // slices per stage: K=32 -> all 4 (20.5 KiB); K=160 -> 3 (63.4 KiB; stages of 3, 3, 2); K=256 -> 2
    static constexpr int SPS[10] = {4, 3, 2, 2, 2, 2, 2, 2, 2, 2};
};

struct SrcT {
    int w_src[13];
};

static SrcT make_src_t(int depth_head) {
    SrcT p;
    static const int nout[13] = {256, 256, 256, 256, 256, 256, 256, 256, 256, 1, 128, 3, 2};
    static const int nin[13] = {96, 256, 256, 256, 256, 352, 256, 256, 256, 256, 283, 128, 128};
    int off = 0;
    for (int l = 0; l < 13; ++l) {
        p.w_src[l] = off;
        off += nout[l] * nin[l] + nout[l];
        if (l == 11 && !depth_head) {
            p.w_src[12] = -1;
            break;
        }
    }
    return p;
}

DDN_EXPORT size_t ddnerf_mlp_x3_packed_t_bytes(int depth_head) {
    (void)depth_head;
    return (size_t)x3_total_bytes<X3BwdPlan>();
}

// A^T element of backward step d: row c (feature of the outgoing delta), column o (row of the incoming delta)
__device__ __forceinline__ float src_wt(const float *__restrict__ P, const SrcT &pl, int d, int c, int o) {
    if (d == 0) {  // heads: incoming rows = raw columns (0..2 rgb, 3 alpha [not an input of this step], 4,5 mu,sigma)
        if (o < 3) return P[pl.w_src[11] + o * 128 + c];
        if ((o == 4 || o == 5) && pl.w_src[12] >= 0) return P[pl.w_src[12] + (o - 4) * 128 + c];
        return 0.0f;
    }
    if (d == 1) {  // d(feat)[c] = sum_o W_dir[o][c] d(dir)[o]  +  W_alpha[c] d(raw)[3]
        if (o < 128) return P[pl.w_src[10] + o * 283 + c];
        if (o == 128 + 3) return P[pl.w_src[9] + c];
        return 0.0f;
    }
    if (d == 2) return P[pl.w_src[8] + o * 256 + c];  // fc_feat
    const int l = 10 - d;                              // d3 -> layers_xyz.7 ... d9 -> layers_xyz.1
    if (l == 5) return P[pl.w_src[5] + o * 352 + 96 + c];
    return P[pl.w_src[l] + o * 256 + c];
}

__global__ void mlp_x3_pack_t_kernel(const float *__restrict__ P, SrcT pl, unsigned short *__restrict__ packed, int words) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= words) return;
    packed[idx] = x3_pack_word<X3BwdPlan>(idx, [&](int d, int c, int o) { return src_wt(P, pl, d, c, o); },
                                          [](int, int) { return 0.0f; });
}

DDN_EXPORT int ddnerf_mlp_x3_pack_t(const float *params, int depth_head, void *packed_t, ddnerf_stream_t stream) {
    DDN_REQUIRE(params && packed_t, DDNERF_E_ARG);
    DDN_REQUIRE(ddn_aligned(packed_t, 16), DDNERF_E_ALIGN);
    SrcT pl = make_src_t(depth_head);
    const int words = x3_total_bytes<X3BwdPlan>() / 2;
    hipLaunchKernelGGL(mlp_x3_pack_t_kernel, dim3((words + 255) / 256), dim3(256), 0, (hipStream_t)stream, params, pl,
                       (unsigned short *)packed_t, words);
    return ddn_launch_status();
}

#ifndef X3_PFD_BWD
#define X3_PFD_BWD 14
#endif
// the staging ring also has to cover the sign-word loads and the delta stores (vmcnt is in order)

template <bool DEPTH>
__global__ __launch_bounds__(X3_WG_THREADS, 1) void mlp_x3_bwd_data_kernel(const float *__restrict__ g_raw,
                                                                           const char *__restrict__ packed_t,
                                                                           const unsigned short *__restrict__ bits,
                                                                           float *__restrict__ deltas, long M, long ld) {
    __shared__ __attribute__((aligned(16))) char lds[2 * X3_STAGE_BYTES_MAX];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, h = lane >> 5;
    const long m = (long)blockIdx.x * X3_WG_SAMPLES + wave * 32 + j;  // < ld (a multiple of 128)
    bf16x8 HAh[16], HAl[16], HBh[16], HBl[16], Xh[8], Xl[8];
    f32x16 keep;
    const char *wp = packed_t;
    x3_dma_stage(wp, lds, x3_stage_bytes<X3BwdPlan>(0, 0), wave, lane);
    {
        // the d(raw) tile as a 32-row B operand: row c = raw column c.  k-step 0, lane half h, position j is row
        // 8(j>>2) + 4h + (j&3): half 0 holds rows 0..3 (rgb, alpha), half 1 rows 4, 5 (mu, sigma); everything else is zero.
        // Stored transposed into deltas rows ROW_X.. as well: the head weight gradients contract over it.
        f32x4 a = {0.0f, 0.0f, 0.0f, 0.0f};
        const f32x4 z = {0.0f, 0.0f, 0.0f, 0.0f};
        if (m < M) {
            const float *g = g_raw + (size_t)m * (DEPTH ? 6 : 4);
            if (h == 0) a = f32x4{g[0], g[1], g[2], g[3]};
            else if (DEPTH) a = f32x4{g[4], g[5], 0.0f, 0.0f};
        }
        split_quad(a, z, Xh[0], Xl[0]);
        split_quad(z, z, Xh[1], Xl[1]);
#pragma unroll
        for (int r = 0; r < 16; ++r)
            __builtin_nontemporal_store(r < 4 ? x3_word(a[r & 3]) : 0u, (unsigned *)deltas + x3_rec_index(ROW_X + x3_tile_row(r, h), m));
    }
    auto io = [&](int row0) { return x3_tile_io(deltas, nullptr, bits, ld, m, h, row0); };
    using PL = X3BwdPlan;
    constexpr int PFD = X3_PFD_BWD;
    // LDS buffer parity of a step's first stage: d0 has one stage (buffer 0), d1 three (starts in 1), then every step has
    // four and starts in buffer 0
    x3_layer<PL, 0, 10, 1, 0, 4, true, 2, PFD, true, true>(wp, lds, HAh, HAl, Xh, Xl, HAh, HAl, keep, wave, lane, io(ROW_DIR));
    x3_layer<PL, 1, 11, 2, 1, 8, false, 2, PFD, false>(wp, lds, HAh, HAl, Xh, Xl, HBh, HBl, keep, wave, lane, io(ROW_FEAT));
    x3_layer<PL, 2, 1, 3, 0, 8, true, 2, PFD, true>(wp, lds, HBh, HBl, Xh, Xl, HAh, HAl, keep, wave, lane, io(256 * 7));
    x3_layer<PL, 3, 1, 4, 0, 8, false, 2, PFD, true>(wp, lds, HAh, HAl, Xh, Xl, HBh, HBl, keep, wave, lane, io(256 * 6));
    x3_layer<PL, 4, 1, 5, 0, 8, true, 2, PFD, true>(wp, lds, HBh, HBl, Xh, Xl, HAh, HAl, keep, wave, lane, io(256 * 5));
    x3_layer<PL, 5, 1, 6, 0, 8, false, 2, PFD, true>(wp, lds, HAh, HAl, Xh, Xl, HBh, HBl, keep, wave, lane, io(256 * 4));
    x3_layer<PL, 6, 1, 7, 0, 8, true, 2, PFD, true>(wp, lds, HBh, HBl, Xh, Xl, HAh, HAl, keep, wave, lane, io(256 * 3));
    x3_layer<PL, 7, 1, 8, 0, 8, false, 2, PFD, true>(wp, lds, HAh, HAl, Xh, Xl, HBh, HBl, keep, wave, lane, io(256 * 2));
    x3_layer<PL, 8, 1, 9, 0, 8, true, 2, PFD, true>(wp, lds, HBh, HBl, Xh, Xl, HAh, HAl, keep, wave, lane, io(256 * 1));
    x3_layer<PL, 9, 1, -1, 0, 8, false, 2, PFD, true>(wp, lds, HAh, HAl, Xh, Xl, HBh, HBl, keep, wave, lane, io(0));
}

DDN_EXPORT int ddnerf_mlp_x3_backward_data(const float *g_raw, const void *packed_t, const void *bits, int depth_head,
                                           float *deltas, long M, long ld, ddnerf_stream_t stream) {
    DDN_REQUIRE(g_raw && packed_t && bits && deltas, DDNERF_E_ARG);
    DDN_REQUIRE(M > 0, DDNERF_E_ARG);
    DDN_REQUIRE(ld >= M && ld % 128 == 0, DDNERF_E_RANGE);
    DDN_REQUIRE(ddn_aligned(packed_t, 16), DDNERF_E_ALIGN);
    dim3 grid((unsigned)((M + X3_WG_SAMPLES - 1) / X3_WG_SAMPLES));
    if (depth_head)
        hipLaunchKernelGGL(mlp_x3_bwd_data_kernel<true>, grid, dim3(X3_WG_THREADS), 0, (hipStream_t)stream, g_raw,
                           (const char *)packed_t, (const unsigned short *)bits, deltas, M, ld);
    else
        hipLaunchKernelGGL(mlp_x3_bwd_data_kernel<false>, grid, dim3(X3_WG_THREADS), 0, (hipStream_t)stream, g_raw,
                           (const char *)packed_t, (const unsigned short *)bits, deltas, M, ld);
    return ddn_launch_status();
}
