// Device code of the hierarchical samplers (K4a / K4b), shared by samplers.hip (the stand-alone kernels) and composite.hip (the coarse
// pass of DDNerfModel's render path, which samples the fine fenceposts in the same launch that composites the coarse ones).
// Include only from translation units compiled with -ffp-contract=off: the bin index of every draw must equal the reference's bit
// for bit (models/samplers.py).
#pragma once
#include "common.h"

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
    if (live && lane == 0) ddn_chain_cdf(wp, cdf, nc);  // torch.cumsum: double running sum, fp32 prefixes; clamp at 1   :88-91
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

// a12  sample_pdf_with_mu_sigma (models/samplers.py:124-215) for ONE ray on one 64-lane wave: inverse CDF over the bins, then the
// truncated in-cell Gaussian through erfinv; endpoints overwritten with near / far; torch.sort.  `w`, `mu`, `sg`, `pi_`, `lt`: the ray's
// nc weights and (smoothed) head values -- global or LDS; scratch in LDS: wp [nc], cdf [nc + 2], out [npad].
__device__ __forceinline__ void dd_sample_row(const float *w, const float *b, const float *mu, const float *sg, const float *pi_, const float *lt,
                                              const float *__restrict__ u_base, const float *rnd_row, float div, float near_, float far_,
                                              float *__restrict__ samples_row, int32_t *__restrict__ bins_ind_row, int nc, int ns, int npad,
                                              int pdf_padding, float *wp, float *cdf, float *out, int lane, bool live) {
    build_cdf(w, nc, pdf_padding, wp, cdf, lane, live);
    if (live) {
        for (int s = lane; s < npad; s += 64) {
            float v = __builtin_inff();
            if (s < ns) {
                float u = make_u(u_base, rnd_row, div, s, true);
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
                if (bins_ind_row) bins_ind_row[s] = ki;
            }
            out[s] = v;
        }
    }
    ddn_wave_sync();
    // torch.sort(dim=1), ascending.  Rows of 64 .. 512 padded elements are sorted in REGISTERS: element e * 64 + lane lives in
    // register e of the lane, so an exchange at distance j < 64 is one cross-lane shuffle and one at j >= 64 a register pair --
    // the in-LDS network below spent ~100 cycles of LDS round trip on each of its 36 stages x 4 passes (kernel 33 -> 20 us at 4096 rays x 129 samples).
    if (npad == 64 || npad == 128 || npad == 256 || npad == 512) {
        if (npad == 64) sort_row_regs<1>(out, samples_row, ns, lane, live);
        else if (npad == 128) sort_row_regs<2>(out, samples_row, ns, lane, live);
        else if (npad == 256) sort_row_regs<4>(out, samples_row, ns, lane, live);
        else sort_row_regs<8>(out, samples_row, ns, lane, live);
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
        for (int s = lane; s < ns; s += 64) samples_row[s] = out[s];
}
