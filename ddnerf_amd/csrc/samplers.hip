// K4a / K4b: hierarchical inverse-CDF samplers, one 64-lane wave per ray.
// Compiled with -ffp-contract=off.  The bin index of every draw must equal the reference's bit for bit
// (given identical weights and u): the blurred pdf, its ATen-ordered sum, the double-accumulated cumsum
// and the `u >= cdf[j]` comparisons are therefore evaluated exactly as torch's CPU kernels do.
#include "common.h"

#define SMP_WAVES 4

// shared front half: weights row -> cdf[0..nc] in LDS      models/samplers.py:69-91 == :130-152
__device__ __forceinline__ void build_cdf(const float *__restrict__ w, int nc, int pdf_padding, float *wp, float *cdf,
                                          int lane, bool live) {
    if (live) {
        for (int j = lane; j < nc; j += 64) {
            float c = w[j], prev = w[j > 0 ? j - 1 : 0], next = w[j < nc - 1 ? j + 1 : nc - 1];
            float v;
            if (pdf_padding) {
                float m0 = fmaxf(prev, c), m1 = fmaxf(c, next);              // weights_max[j], [j+1]   :75
                v = 0.5f * (m0 + m1) + 0.01f;                                // :76, :79
            } else {
                v = ((0.8f * c + 0.1f * prev) + 0.1f * next) + 0.01f;        // :85
            }
            wp[j] = v;
        }
    }
    ddn_wave_sync();
    float sum = ddn_aten_sum_wave(wp, nc, lane);                             // :87 torch.sum order
    if (live)
        for (int j = lane; j < nc; j += 64) wp[j] = wp[j] / sum;             // pdf
    ddn_wave_sync();
    if (live && lane == 0) {  // torch.cumsum: double running sum, fp32 prefixes; clamp at 1   :88-91
        double a = 0.0;
        cdf[0] = 0.0f;
        for (int j = 0; j < nc - 1; ++j) {
            a += (double)wp[j];
            cdf[j + 1] = fminf(1.0f, (float)a);
        }
        cdf[nc] = 1.0f;
    }
    ddn_wave_sync();
}

// k = last j in [0,len) with cdf[j] <= u  (mask = u >= cdf[j] is a prefix because cdf is non-decreasing)
__device__ __forceinline__ int last_le(const float *cdf, int len, float u) {
    int lo = 0, hi = len;
    while (lo < hi) {
        int mid = (lo + hi) >> 1;
        if (u >= cdf[mid]) lo = mid + 1;
        else hi = mid;
    }
    return lo > 0 ? lo - 1 : 0;
}

__device__ __forceinline__ float make_u(const float *u_base, const float *rnd, float div, int s, bool clamp_lo) {
    float u = u_base[s];
    if (rnd) {
        u = u + rnd[s] / div;                                                // :102 / :165
        u = fminf(u, 0.9999f);                                               // :104 / :169
        if (clamp_lo) u = fmaxf(u, 0.0f);                                    // :171
    }
    return u;
}

// ---------------------------------------------------------------------------------------------------
// a11  sample_pdf   models/samplers.py:64-121
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sample_pdf_kernel(const float *__restrict__ bins,
                                                         const float *__restrict__ weights,
                                                         const float *__restrict__ u_base, const float *__restrict__ rnd,
                                                         float div, float *__restrict__ samples, int n, int nc, int ns,
                                                         int pdf_padding) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int ray = blockIdx.x * SMP_WAVES + wave;
    const bool live = ray < n;
    float *wp = smem + (size_t)wave * (2 * nc + 2);
    float *cdf = wp + nc;
    const size_t r = live ? ray : 0;
    build_cdf(weights + r * nc, nc, pdf_padding, wp, cdf, lane, live);
    if (!live) return;
    const float *b = bins + r * (nc + 1);
    for (int s = lane; s < ns; s += 64) {
        float u = make_u(u_base, rnd ? rnd + r * ns : nullptr, div, s, false);
        int k = last_le(cdf, nc + 1, u), k1 = k < nc ? k + 1 : nc;
        float t = (u - cdf[k]) / (cdf[k1] - cdf[k]);
        if (t != t) t = 0.0f;                                                // nan_to_num(., 0)  :118
        t = fminf(fmaxf(t, 0.0f), 1.0f);                                     // clip (also maps +-inf)
        samples[r * ns + s] = b[k] + t * (b[k1] - b[k]);                     // :119
    }
}

DDN_EXPORT int ddnerf_sample_pdf(const float *bins, const float *weights, const float *u_base, const float *rnd,
                                 float *samples, int n, int nc, int ns, int pdf_padding, ddnerf_stream_t stream) {
    DDN_REQUIRE(bins && weights && u_base && samples, DDNERF_E_ARG);
    DDN_REQUIRE(n > 0 && nc > 1 && ns > 0, DDNERF_E_ARG);  // the reference raises for a single coarse cell
    size_t lds = (size_t)SMP_WAVES * (2 * nc + 2) * sizeof(float);
    DDN_REQUIRE(lds <= 64 * 1024, DDNERF_E_RANGE);
    float div = (float)((double)ns + 1e-5);  // (1/s)+1e-5 with s = 1/ns, cast to the tensor dtype
    hipLaunchKernelGGL(sample_pdf_kernel, dim3((n + SMP_WAVES - 1) / SMP_WAVES), dim3(256), lds, (hipStream_t)stream,
                       bins, weights, u_base, rnd, div, samples, n, nc, ns, pdf_padding);
    return ddn_launch_status();
}

// One padded row (NE * 64 values in LDS, +inf behind the real ones) sorted ascending in registers by a bitonic network;
// the first ns values go to `dst`.  A compare-exchange keeps min(a, c) at the lower index of an ascending pair -- the values
// are never NaN, and equal values are interchangeable, so this is the swap rule of the LDS network.
template <int NE>
__device__ __forceinline__ void sort_row_regs(const float *row, float *__restrict__ dst, int ns, int lane, bool live) {
    float v[NE];
#pragma unroll
    for (int e = 0; e < NE; ++e) v[e] = row[e * 64 + lane];
#pragma unroll
    for (int k = 2; k <= NE * 64; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
            if (j >= 64) {
#pragma unroll
                for (int e = 0; e < NE; ++e) {
                    const int e2 = e ^ (j >> 6);
                    if (e2 > e) {
                        const bool up = ((e * 64) & k) == 0;  // (bit k of the index lies in the register number: k >= 128 here)
                        const float a = v[e], c = v[e2];
                        v[e] = up ? fminf(a, c) : fmaxf(a, c);
                        v[e2] = up ? fmaxf(a, c) : fminf(a, c);
                    }
                }
            } else {
#pragma unroll
                for (int e = 0; e < NE; ++e) {
                    const float a = v[e], c = __shfl_xor(a, j);
                    const bool lower = (lane & j) == 0, up = ((e * 64 + lane) & k) == 0;
                    v[e] = (lower == up) ? fminf(a, c) : fmaxf(a, c);
                }
            }
        }
    }
    if (live) {
#pragma unroll
        for (int e = 0; e < NE; ++e)
            if (e * 64 + lane < ns) dst[e * 64 + lane] = v[e];
    }
}

// ---------------------------------------------------------------------------------------------------
// a12  sample_pdf_with_mu_sigma   models/samplers.py:124-215
// Inverse CDF over bins, then the truncated in-cell Gaussian through erfinv; endpoints overwritten with
// near/far; torch.sort as an in-LDS bitonic network over the next power of two (+inf padding).
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sample_pdf_mu_sigma_kernel(
    const float *__restrict__ bins, const float *__restrict__ weights, const float *__restrict__ mus,
    const float *__restrict__ sigmas, const float *__restrict__ part, const float *__restrict__ left,
    const float *__restrict__ u_base, const float *__restrict__ rnd, float div, float near_, float far_,
    float *__restrict__ samples, int32_t *__restrict__ bins_ind, int n, int nc, int ns, int npad, int pdf_padding) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int ray = blockIdx.x * SMP_WAVES + wave;
    const bool live = ray < n;
    float *wp = smem + (size_t)wave * (2 * nc + 2 + npad);
    float *cdf = wp + nc;
    float *out = cdf + nc + 2;
    const size_t r = live ? ray : 0;
    build_cdf(weights + r * nc, nc, pdf_padding, wp, cdf, lane, live);
    const float *b = bins + r * (nc + 1);
    const float *mu = mus + r * nc, *sg = sigmas + r * nc, *pi_ = part + r * nc, *lt = left + r * nc;
    if (live) {
        for (int s = lane; s < npad; s += 64) {
            float v = __builtin_inff();
            if (s < ns) {
                float u = make_u(u_base, rnd ? rnd + r * ns : nullptr, div, s, true);
                float z, t, g0, g1;
                int ki = 0;
                if (nc == 1) {                                               // :185-190
                    z = u * pi_[0] + lt[0];
                    z = ddn_norm_icdf(z);
                    t = z * sg[0] + mu[0];
                    g0 = b[0];
                    g1 = b[1];
                } else {
                    int k = last_le(cdf, nc + 1, u), k1 = k < nc ? k + 1 : nc;
                    ki = k;
                    while (ki > 0 && b[ki - 1] == b[k]) --ki;                // torch.max: first index of the max
                    if (ki > nc - 1) ki = nc - 1;
                    z = ((u - cdf[k]) / (cdf[k1] - cdf[k])) * pi_[ki] + lt[ki];  // :198
                    z = fminf(z, 0.999f);                                    // :199
                    z = ddn_norm_icdf(z);                                    // :204
                    t = z * sg[ki] + mu[ki];
                    g0 = b[k];
                    g1 = b[k1];
                }
                t = fminf(fmaxf(t, 0.0f), 0.99999f);                         // :206
                v = g0 + t * (g1 - g0);                                      // :208
                if (s == ns - 1) v = far_;                                   // :210
                if (s == 0) v = near_;                                       // :211
                if (bins_ind) bins_ind[r * ns + s] = ki;
            }
            out[s] = v;
        }
    }
    ddn_wave_sync();
    // torch.sort(dim=1), ascending.  Rows of 64 .. 512 padded elements are sorted in REGISTERS: element e * 64 + lane lives in
    // register e of the lane, so an exchange at distance j < 64 is one cross-lane shuffle and one at j >= 64 a register pair --
    // the in-LDS network below spent ~100 cycles of LDS round trip on each of its 36 stages x 4 passes (kernel 33 -> 20 us at 4096 rays x 129 samples).
    if (npad == 64 || npad == 128 || npad == 256 || npad == 512) {
        if (npad == 64) sort_row_regs<1>(out, samples + r * ns, ns, lane, live);
        else if (npad == 128) sort_row_regs<2>(out, samples + r * ns, ns, lane, live);
        else if (npad == 256) sort_row_regs<4>(out, samples + r * ns, ns, lane, live);
        else sort_row_regs<8>(out, samples + r * ns, ns, lane, live);
        return;
    }
    // bitonic network in LDS (any other row length)
    for (int k = 2; k <= npad; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            if (live) {
                for (int i = lane; i < npad; i += 64) {
                    int ixj = i ^ j;
                    if (ixj > i) {
                        float a = out[i], c = out[ixj];
                        bool up = (i & k) == 0;
                        if ((a > c) == up) {
                            out[i] = c;
                            out[ixj] = a;
                        }
                    }
                }
            }
            ddn_wave_sync();
        }
    }
    if (live)
        for (int s = lane; s < ns; s += 64) samples[r * ns + s] = out[s];
}

DDN_EXPORT int ddnerf_sample_pdf_mu_sigma(const float *bins, const float *weights, const float *mus,
                                          const float *sigmas, const float *part, const float *left,
                                          const float *u_base, const float *rnd, float near_, float far_,
                                          float *samples, int32_t *bins_ind, int n, int nc, int ns, int pdf_padding,
                                          ddnerf_stream_t stream) {
    DDN_REQUIRE(bins && weights && mus && sigmas && part && left && u_base && samples, DDNERF_E_ARG);
    DDN_REQUIRE(n > 0 && nc > 0 && ns > 1, DDNERF_E_ARG);
    int npad = 2;
    while (npad < ns) npad <<= 1;
    size_t lds = (size_t)SMP_WAVES * (2 * nc + 2 + npad) * sizeof(float);
    DDN_REQUIRE(lds <= 64 * 1024, DDNERF_E_RANGE);
    float div = (float)((double)ns + 1e-5);
    hipLaunchKernelGGL(sample_pdf_mu_sigma_kernel, dim3((n + SMP_WAVES - 1) / SMP_WAVES), dim3(256), lds,
                       (hipStream_t)stream, bins, weights, mus, sigmas, part, left, u_base, rnd, div, near_, far_,
                       samples, bins_ind, n, nc, ns, npad, pdf_padding);
    return ddn_launch_status();
}
