// K2b weight gradients (fp32):  dW[out][in] = sum_s delta^T[out][s] * act^T[in][s]  on the fp32 matrix cores.
//
// Both operands are the transposed [feature][sample] matrices the fused training kernels stored (mlp_f32_train.hip),
// so the contraction (sample) axis is contiguous in BOTH: a workgroup walks its share of the samples in tiles of 32,
// loads each tile with full 128-byte row segments into LDS (register-prefetched one tile ahead), and every wave
// accumulates a [32 x 32*NCT] block of dW (<= 128 accumulator registers; 8 waves for 256 out-rows) in registers with v_mfma_f32_32x32x2_f32 (A = delta rows, B = input
// rows, k = 2 samples per instruction).  One workgroup per CU-sized share of the samples; each writes a partial slab,
// a second kernel adds the slabs in a fixed order (split-K without atomics: training stays reproducible).
// Bias gradients (row sums of delta) fall out of the A fragments on the VALU.
#include "mlp_f32_common.h"
#include "wgrad_reduce.h"

#define WG_TILE 32           // samples per LDS tile
#define WG_LDW (WG_TILE + 4)  // LDS row stride in floats: conflict-free ds_read_b128 of 4 samples x 32 rows

template <int ROWT, int NCT, int NOP, int THREADS>
__global__ __launch_bounds__(THREADS) void wgrad_f32_kernel(const float *__restrict__ dT, const float *__restrict__ aT,
                                                           long M, long ld, int tiles_per_wg,
                                                           float *__restrict__ slabs, float *__restrict__ bias_slabs) {
    constexpr int N_IN = 32 * NCT, n_out_pad = NOP;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *dl = lds;                        // [n_out_pad][WG_LDW]
    float *al = lds + n_out_pad * WG_LDW;   // [N_IN][WG_LDW]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, h = lane >> 5;
    const int RG = n_out_pad / (32 * ROWT);  // row groups (1, 2 or 4); waves beyond them split the samples
    constexpr int WAVES = THREADS / 64;
    const int rg = wave % RG, sub = wave / RG, nsub = WAVES / RG;
    const long tile0 = (long)blockIdx.x * tiles_per_wg;
    const long ntiles_total = (M + WG_TILE - 1) / WG_TILE;
    const int ntiles = (int)max(0L, min((long)tiles_per_wg, ntiles_total - tile0));
    constexpr int PIECES = (NOP + N_IN) * 8;                       // float4 pieces per tile
    constexpr int MAXR = (PIECES + THREADS - 1) / THREADS;         // copy rounds (the last one may be partial)
    f32x4 pf[MAXR];

    f32x16 acc[ROWT][NCT];
#pragma unroll
    for (int rt = 0; rt < ROWT; ++rt)
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[rt][ct][r] = 0.0f;
    float bsum[ROWT];
#pragma unroll
    for (int rt = 0; rt < ROWT; ++rt) bsum[rt] = 0.0f;

    auto fetch = [&](int t) {  // tile t of this workgroup -> registers
        const long s0 = (tile0 + t) * WG_TILE;
#pragma unroll
        for (int r = 0; r < MAXR; ++r) {
            const int idx = r * THREADS + tid, row = idx >> 3, q = idx & 7;
            // no masking of the ragged tail: ld is a multiple of 128 >= M, the pad columns of `deltas` are exact zeros
            // (backward_data) and those of `acts` are finite copies of the last sample (forward_train)
            if ((r + 1) * THREADS <= PIECES || idx < PIECES)
                pf[r] = *(const f32x4 *)((row < n_out_pad ? dT + (size_t)row * ld : aT + (size_t)(row - n_out_pad) * ld) + s0 + 4 * q);
        }
    };
    auto park = [&]() {
#pragma unroll
        for (int r = 0; r < MAXR; ++r) {
            const int idx = r * THREADS + tid, row = idx >> 3, q = idx & 7;
            if ((r + 1) * THREADS <= PIECES || idx < PIECES) *(f32x4 *)(lds + row * WG_LDW + 4 * q) = pf[r];
        }
    };

    if (ntiles > 0) fetch(0);
    for (int t = 0; t < ntiles; ++t) {
        __syncthreads();  // everyone is done reading the previous tile
        park();
        __syncthreads();
        if (t + 1 < ntiles) fetch(t + 1);  // in flight under this tile's MFMAs
        if (t % nsub == sub) {
#pragma unroll
            for (int s8 = 0; s8 < WG_TILE / 8; ++s8) {
                f32x4 a[ROWT];
#pragma unroll
                for (int rt = 0; rt < ROWT; ++rt) {
                    a[rt] = *(const f32x4 *)(dl + ((rg * ROWT + rt) * 32 + i) * WG_LDW + 8 * s8 + 4 * h);
                    bsum[rt] += (a[rt].x + a[rt].y) + (a[rt].z + a[rt].w);
                }
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct) {
                    const f32x4 b = *(const f32x4 *)(al + (ct * 32 + i) * WG_LDW + 8 * s8 + 4 * h);
#pragma unroll
                    for (int c = 0; c < 4; ++c)
#pragma unroll
                        for (int rt = 0; rt < ROWT; ++rt)
                            acc[rt][ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[rt][c], b[c], acc[rt][ct], 0, 0, 0);
                }
            }
        }
    }
    // partial slab of this (workgroup, sample sub-chunk): [n_out_pad][N_IN]
    float *slab = slabs + ((size_t)blockIdx.x * nsub + sub) * (size_t)n_out_pad * N_IN;
#pragma unroll
    for (int rt = 0; rt < ROWT; ++rt)
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                slab[(size_t)((rg * ROWT + rt) * 32 + tile_row(r, h)) * N_IN + ct * 32 + i] = acc[rt][ct][r];
    if (bias_slabs) {
#pragma unroll
        for (int rt = 0; rt < ROWT; ++rt) {
            float s = bsum[rt] + __shfl_xor(bsum[rt], 32);
            if (h == 0) bias_slabs[((size_t)blockIdx.x * nsub + sub) * n_out_pad + (rg * ROWT + rt) * 32 + i] = s;
        }
    }
}

DDN_EXPORT size_t ddnerf_mlp_f32_wgrad_workspace_floats(long M) {
    // worst job: 256 x 352 slab (+ 256 bias) per workgroup, at most 4 sample sub-chunks per workgroup
    long ntiles = (M + WG_TILE - 1) / WG_TILE;
    long nwg = ntiles < 256 ? ntiles : 256;
    return (size_t)nwg * 4 * (32 * 256 + 32) + (size_t)nwg * (256 * 256 + 256) + 1024;
}

// One weight-gradient job: rows [drow0, drow0 + n_out) of `deltas` against rows [arow0, arow0 + n_in) of `acts`
// (n_in a multiple of 32: 32, 96, 128, 256).  Result rows/cols are written (not accumulated) into
//   dst[r * dst_ld + dst_col0 + c],  r < n_out, c < n_in_used;   dst_bias[r] = sum_s delta[r][s]   (may be NULL).
DDN_EXPORT int ddnerf_mlp_f32_wgrad(const float *deltas, int drow0, int n_out, const float *acts, int arow0, int n_in,
                                    int n_in_used, long M, long ld, float *dst, int dst_ld, int dst_col0,
                                    float *dst_bias, float *workspace, ddnerf_stream_t stream) {
    DDN_REQUIRE(deltas && acts && dst && workspace, DDNERF_E_ARG);
    DDN_REQUIRE(M > 0 && n_out > 0 && n_out <= 256 && n_in_used > 0 && n_in_used <= n_in, DDNERF_E_ARG);
    DDN_REQUIRE(n_in == 32 || n_in == 96 || n_in == 128 || n_in == 256, DDNERF_E_RANGE);
    DDN_REQUIRE(ld % 4 == 0, DDNERF_E_ALIGN);
    hipStream_t st = (hipStream_t)stream;
    const int n_out_pad = n_out > 128 ? 256 : (n_out > 32 ? 128 : 32);
    const int threads = n_out_pad == 256 ? 512 : 256;
    const int nsub = (threads / 64) / (n_out_pad / 32);
    const long ntiles = (M + WG_TILE - 1) / WG_TILE;
    const int nwg = (int)(ntiles < 256 ? ntiles : 256);
    const int tiles_per_wg = (int)((ntiles + nwg - 1) / nwg);
    const float *dT = deltas + (size_t)drow0 * ld, *aT = acts + (size_t)arow0 * ld;
    float *slabs = workspace;
    const size_t slab_stride = (size_t)n_out_pad * n_in;
    float *bias_slabs = dst_bias ? workspace + (size_t)nwg * nsub * slab_stride : nullptr;
    const size_t lds = (size_t)(n_out_pad + n_in) * WG_LDW * sizeof(float);
    dim3 grid(nwg), block(threads);
#define LAUNCH(CT, NP, TH) hipLaunchKernelGGL((wgrad_f32_kernel<1, CT, NP, TH>), grid, block, lds, st, dT, aT, M, ld, \
                                              tiles_per_wg, slabs, bias_slabs)
    const int nct = n_in / 32;
    if (n_out_pad == 256) {
        if (nct == 8) LAUNCH(8, 256, 512); else if (nct == 3) LAUNCH(3, 256, 512); else if (nct == 4) LAUNCH(4, 256, 512); else LAUNCH(1, 256, 512);
    } else if (n_out_pad == 128) {
        if (nct == 8) LAUNCH(8, 128, 256); else if (nct == 3) LAUNCH(3, 128, 256); else if (nct == 4) LAUNCH(4, 128, 256); else LAUNCH(1, 128, 256);
    } else {
        if (nct == 8) LAUNCH(8, 32, 256); else if (nct == 3) LAUNCH(3, 32, 256); else if (nct == 4) LAUNCH(4, 32, 256); else LAUNCH(1, 32, 256);
    }
#undef LAUNCH
    const int nslabs = nwg * nsub;
    int total = n_out * n_in_used;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((total + 63) / 64), dim3(256), 0, st, slabs, nslabs, slab_stride, n_in,
                       0, 0, n_out, n_in_used, dst, dst_ld, dst_col0);
    if (dst_bias)
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((n_out + 63) / 64), dim3(256), 0, st, bias_slabs, nslabs,
                           (size_t)n_out_pad, 1, 0, 0, n_out, 1, dst_bias, 1, 0);
    return ddn_launch_status();
}
