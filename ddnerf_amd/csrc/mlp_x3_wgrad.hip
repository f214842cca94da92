// K2b' weight gradients at fp32-class accuracy on the bf16 matrix cores ("bf16x3").
//
// dW[out][in] = sum_s delta^T[out][s] * act^T[in][s] is a contraction over up to 524,288 samples; on the fp32 MFMA it is
// compute-bound (wgrad_f32_kernel: 0.6 ms per 256x256 job).  Every fp32 operand splits exactly into hi + lo + r with
// hi = bf16(x), lo = bf16(x - hi), |r| <= 2^-17 |x|, so   a*b = hi_a*hi_b + hi_a*lo_b + lo_a*hi_b + O(2^-16 |a b|):
// three v_mfma_f32_32x32x16_bf16 (fp32 accumulation) replace eight v_mfma_f32_32x32x2_f32 -- 5.3x fewer matrix-pipe
// cycles at a relative product error of ~1.5e-5, far below the fp32 reassociation noise of a half-million-term sum --
// and the job becomes HBM-bound (it must read 2 x 256 rows x M floats once: 0.2 ms at M = 524,288).
//
// Operands are the transposed [feature][sample] fp32 matrices of mlp_f32_train.hip.  A workgroup walks its share of
// the samples in 32-sample tiles: global -> registers (prefetched one tile ahead: 64 KiB in flight per CU keeps HBM
// busy), split into hi / lo bf16 while parking in LDS (row stride 80 B: conflict-free ds_read_b128), then RG x CG waves
// each accumulate RT x CT 32x32 tiles.  Partial slabs per workgroup, reduced in a fixed order (mlp_f32_wgrad.hip).
//
// What limits it now is the HBM round trip per tile, not bandwidth: one tile in flight per workgroup gives 0.29 ms per
// 256x256 job at M = 524,288 where the bytes alone need 0.21.  Tried and measured slower (0.32 ms): 16-sample tiles by
// LDS-DMA into a three-deep raw fp32 ring + a conversion pass -- a 16-sample row segment is 64 B, half an HBM line, and
// 160 KiB of LDS has no room for a deeper ring of 32-sample tiles; a second register set for a second tile in flight
// does not fit either (244 of 256 registers are in use at 8 waves per CU); two 256-thread workgroups per CU, each with
// half of the out-rows and its own tile pipeline, made a training step 1.4 ms slower.  Round 2: a 4-wave configuration (one
// wave per SIMD, 4 x 4 accumulator tiles per wave) fits one tile in flight in 448 registers and made the step 0.8 ms slower;
// with two tiles in flight it needs more than the 512 registers (319 spills).
#include "mlp_f32_common.h"
#include "wgrad_reduce.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

#define X3_TILE 32               // samples per LDS tile (two k = 16 MFMA steps)
#define X3_ROWB 80               // bytes per LDS row: 32 bf16 + 16 B pad

__device__ __forceinline__ void split4(const f32x4 v, bf16x4 &hi, bf16x4 &lo) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const __bf16 h = (__bf16)v[c];  // round to nearest even
        hi[c] = h;
        lo[c] = (__bf16)(v[c] - (float)h);  // the subtraction is exact
    }
}

template <int RT, int CT, int RG, int CG>
__global__ __launch_bounds__(64 * RG * CG) void wgrad_x3_kernel(const float *__restrict__ dT, const float *__restrict__ aT,
                                                                 long M, long ld, int tiles_per_wg,
                                                                 float *__restrict__ slabs, float *__restrict__ bias_slabs) {
    constexpr int NOP = 32 * RT * RG, NIN = 32 * CT * CG, THREADS = 64 * RG * CG, ROWS = NOP + NIN;
    constexpr int PIECES = ROWS * 8, MAXR = (PIECES + THREADS - 1) / THREADS;
    extern __shared__ __attribute__((aligned(16))) char lds_raw[];
    char *hi_l = lds_raw, *lo_l = lds_raw + ROWS * X3_ROWB;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, h = lane >> 5;
    const int rg = wave % RG, cg = wave / RG;
    const long tile0 = (long)blockIdx.x * tiles_per_wg;
    const long ntiles_total = (M + X3_TILE - 1) / X3_TILE;
    const int ntiles = (int)max(0L, min((long)tiles_per_wg, ntiles_total - tile0));

    f32x4 pf[MAXR];
    float bsum[MAXR];
#pragma unroll
    for (int r = 0; r < MAXR; ++r) bsum[r] = 0.0f;
    f32x16 acc[RT][CT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[rt][ct][r] = 0.0f;

    auto fetch = [&](int t) {
        const long s0 = (tile0 + t) * X3_TILE;
#pragma unroll
        for (int r = 0; r < MAXR; ++r) {
            const int idx = r * THREADS + tid, row = idx >> 3, q = idx & 7;
            // ragged tail: ld is a multiple of 128 >= M; the pad columns of `deltas` are exact zeros (backward_data)
            // and those of `acts` finite (forward_train), so they contribute exactly nothing
            if ((r + 1) * THREADS <= PIECES || idx < PIECES)
                pf[r] = __builtin_nontemporal_load((const f32x4 *)((row < NOP ? dT + (size_t)row * ld : aT + (size_t)(row - NOP) * ld) + s0 + 4 * q));
        }
    };
    auto park = [&]() {
#pragma unroll
        for (int r = 0; r < MAXR; ++r) {
            const int idx = r * THREADS + tid, row = idx >> 3, q = idx & 7;
            if ((r + 1) * THREADS <= PIECES || idx < PIECES) {
                bf16x4 hi, lo;
                split4(pf[r], hi, lo);
                *(bf16x4 *)(hi_l + row * X3_ROWB + 8 * q) = hi;
                *(bf16x4 *)(lo_l + row * X3_ROWB + 8 * q) = lo;
                if (row < NOP) bsum[r] += (pf[r].x + pf[r].y) + (pf[r].z + pf[r].w);
            }
        }
    };

    if (ntiles > 0) fetch(0);
    for (int t = 0; t < ntiles; ++t) {
        __syncthreads();  // everyone is done reading the previous tile
        park();
        __syncthreads();
        if (t + 1 < ntiles) fetch(t + 1);  // in flight under this tile's MFMAs
#pragma unroll
        for (int s = 0; s < X3_TILE / 16; ++s) {
            bf16x8 ah[RT], al[RT];
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                const int off = ((rg * RT + rt) * 32 + i) * X3_ROWB + 32 * s + 16 * h;
                ah[rt] = *(const bf16x8 *)(hi_l + off);
                al[rt] = *(const bf16x8 *)(lo_l + off);
            }
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                const int off = (NOP + (cg * CT + ct) * 32 + i) * X3_ROWB + 32 * s + 16 * h;
                const bf16x8 bh = *(const bf16x8 *)(hi_l + off);
                const bf16x8 bl = *(const bf16x8 *)(lo_l + off);
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
                    // small terms first
                    acc[rt][ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[rt], bh, acc[rt][ct], 0, 0, 0);
                    acc[rt][ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[rt], bl, acc[rt][ct], 0, 0, 0);
                    acc[rt][ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[rt], bh, acc[rt][ct], 0, 0, 0);
                }
            }
        }
    }
    float *slab = slabs + (size_t)blockIdx.x * NOP * NIN;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                slab[(size_t)((rg * RT + rt) * 32 + tile_row(r, h)) * NIN + (cg * CT + ct) * 32 + i] = acc[rt][ct][r];
    if (bias_slabs) {
        // piece (row, q) is owned by the same thread in every tile; the 8 q-pieces of a row sit in 8 adjacent lanes
#pragma unroll
        for (int r = 0; r < MAXR; ++r) {
            const int idx = r * THREADS + tid, row = idx >> 3;
            float s = bsum[r];
            s += __shfl_xor(s, 1);
            s += __shfl_xor(s, 2);
            s += __shfl_xor(s, 4);
            if ((idx & 7) == 0 && row < NOP) bias_slabs[(size_t)blockIdx.x * NOP + row] = s;
        }
    }
}

// Same contract as ddnerf_mlp_f32_wgrad (mlp_f32_wgrad.hip); products carry a relative error of ~2^-16.
DDN_EXPORT int ddnerf_mlp_x3_wgrad(const float *deltas, int drow0, int n_out, const float *acts, int arow0, int n_in,
                                   int n_in_used, long M, long ld, float *dst, int dst_ld, int dst_col0,
                                   float *dst_bias, float *workspace, ddnerf_stream_t stream) {
    DDN_REQUIRE(deltas && acts && dst && workspace, DDNERF_E_ARG);
    DDN_REQUIRE(M > 0 && n_out > 0 && n_out <= 256 && n_in_used > 0 && n_in_used <= n_in, DDNERF_E_ARG);
    DDN_REQUIRE(n_in == 32 || n_in == 96 || n_in == 128 || n_in == 256, DDNERF_E_RANGE);
    DDN_REQUIRE(ld % 4 == 0, DDNERF_E_ALIGN);
    hipStream_t st = (hipStream_t)stream;
    const int n_out_pad = n_out > 128 ? 256 : (n_out > 32 ? 128 : 32);
    const long ntiles = (M + X3_TILE - 1) / X3_TILE;
    const int nwg = (int)(ntiles < 256 ? ntiles : 256);
    const int tiles_per_wg = (int)((ntiles + nwg - 1) / nwg);
    const float *dT = deltas + (size_t)drow0 * ld, *aT = acts + (size_t)arow0 * ld;
    float *slabs = workspace;
    const size_t slab_stride = (size_t)n_out_pad * n_in;
    float *bias_slabs = dst_bias ? workspace + (size_t)nwg * slab_stride : nullptr;
    const size_t lds = (size_t)(n_out_pad + n_in) * X3_ROWB * 2;
#define LAUNCH(RT, CT, RG, CG)                                                                                         \
    hipLaunchKernelGGL((wgrad_x3_kernel<RT, CT, RG, CG>), dim3(nwg), dim3(64 * RG * CG), lds, st, dT, aT, M, ld,       \
                       tiles_per_wg, slabs, bias_slabs)
    if (n_out_pad == 256) {
        if (n_in == 256) LAUNCH(2, 4, 4, 2); else if (n_in == 128) LAUNCH(1, 4, 8, 1); else if (n_in == 96) LAUNCH(1, 3, 8, 1); else LAUNCH(1, 1, 8, 1);
    } else if (n_out_pad == 128) {
        if (n_in == 256) LAUNCH(1, 4, 4, 2); else if (n_in == 128) LAUNCH(1, 2, 4, 2); else if (n_in == 96) LAUNCH(1, 3, 4, 1); else LAUNCH(1, 1, 4, 1);
    } else {
        if (n_in == 256) LAUNCH(1, 2, 1, 4); else if (n_in == 128) LAUNCH(1, 1, 1, 4); else if (n_in == 96) LAUNCH(1, 3, 1, 1); else LAUNCH(1, 1, 1, 1);
    }
#undef LAUNCH
    const int total = n_out * n_in_used;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((total + 63) / 64), dim3(256), 0, st, slabs, nwg, slab_stride, n_in, 0, 0,
                       n_out, n_in_used, dst, dst_ld, dst_col0);
    if (dst_bias)
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((n_out + 63) / 64), dim3(256), 0, st, bias_slabs, nwg,
                           (size_t)n_out_pad, 1, 0, 0, n_out, 1, dst_bias, 1, 0);
    return ddn_launch_status();
}
