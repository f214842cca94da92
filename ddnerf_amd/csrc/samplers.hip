// K4a / K4b: hierarchical inverse-CDF samplers, one 64-lane wave per ray.
// Compiled with -ffp-contract=off.  The bin index of every draw must equal the reference's bit for bit
// (given identical weights and u): the blurred pdf, its ATen-ordered sum, the double-accumulated cumsum
// and the `u >= cdf[j]` comparisons are therefore evaluated exactly as torch's CPU kernels do.
#include "common.h"
#include "sampler_device.h"

#define SMP_WAVES 4

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
    dd_sample_row(weights + r * nc, bins + r * (nc + 1), mus + r * nc, sigmas + r * nc, part + r * nc, left + r * nc, u_base,
                  rnd ? rnd + r * ns : nullptr, div, near_, far_, samples + r * ns, bins_ind ? bins_ind + r * ns : nullptr, nc, ns, npad,
                  pdf_padding, wp, cdf, out, lane, live);
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
