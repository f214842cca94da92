// K3: alpha compositing (exclusive product scan along the ray) and the DD head.
// Compiled with -ffp-contract=off so that every product/sum rounds like the reference's ATen CPU ops.
#include "common.h"
#include "sampler_device.h"

// ---------------------------------------------------------------------------------------------------
// a10  volume_render_radiance_field  general_utils/volume_rendering_utils.py:6-85
//      cumprod_exclusive             general_utils/nerf_helpers.py:43-64
//
// One 64-lane wave per ray, four rays per 256-thread block.  Lanes stride over the S samples of the ray;
// per-ray scratch lives in LDS.  The transmittance scan reproduces torch.cumprod's CPU arithmetic
// (running product in double, each prefix rounded to fp32) by letting one lane run the S-step chain;
// the three row sums (acc, depth, corrected depth) reproduce torch.sum's 8-lane x 4-accumulator order
// on 8 lanes of the wave.  Rays are independent, so the cost of those short serial chains is hidden by
// the other waves resident on the CU.
// ---------------------------------------------------------------------------------------------------
#define COMP_WAVES 4

// HEAD (the coarse pass of DDNerfModel, render path): the same launch also evaluates the DD head of its rays (dd_head_kernel's
// arithmetic, element for element; `raw` has 6 columns, `mus` is an OUTPUT), writes the per-block partial sums of the two L2
// regularisers and the per-bin flags / per-ray counts of the level-0 records (dd_records_count_kernel's arithmetic: its row sum
// IS this kernel's wsum).  keep_out (the fine pass): the dp loss's row filter torch.sum(w1) > 1e-10 (dpl_keep_kernel: again wsum).
struct HeadOut {
    float *mus, *sigmas, *left, *part, *ssig, *sleft, *spart, *partials;
    unsigned char *rec_flags;
    int *rec_counts;
    float smooth;
};
// HEAD + samples != nullptr: the launch also draws the fine pass's fenceposts of its rays (sample_pdf_with_mu_sigma,
// models/samplers.py:124-215, called at models/models.py:227-237 with exactly what this kernel has just computed: the returned weights, the
// head's mus and its SMOOTHED sigmas / part-inside / left tails) -- dd_sample_row of sampler_device.h, the stand-alone sampler kernel's
// code, fed from LDS instead of from the arrays this kernel also writes to memory.  One launch and five [n, nc] reads less per chunk.
// noise == NULL and ng.on: the density noise randn * std of volume_rendering_utils.py:29-37 is drawn INSIDE the kernel instead of being
// read from a tensor a generator launch wrote: element (ray, j) of this pass is Philox4x32-10 under the key (seed ^ a constant),
// counter (ng.base + ray * S + j, offset), two of its words through Box-Muller -- a pure function of (seed, offset, element), so a
// replay of the same torch generator state reproduces it.  The host takes (seed, offset) from torch's CUDA generator and advances
// its offset like any torch random kernel would (ddnerf_amd/models.py).  ddnerf_debug_philox_normal writes the same values to memory
// (the tests composite once with the in-kernel noise and once with that tensor: bit-identical outputs).
struct NoiseGen {
    unsigned long long seed, offset, base;
    float std;
    int on;
};
__device__ __forceinline__ void philox_round(unsigned (&c)[4], unsigned k0, unsigned k1) {
    const unsigned long long p0 = (unsigned long long)0xD2511F53u * c[0], p1 = (unsigned long long)0xCD9E8D57u * c[2];
    const unsigned n0 = (unsigned)(p1 >> 32) ^ c[1] ^ k0, n1 = (unsigned)p1, n2 = (unsigned)(p0 >> 32) ^ c[3] ^ k1, n3 = (unsigned)p0;
    c[0] = n0;
    c[1] = n1;
    c[2] = n2;
    c[3] = n3;
}
__device__ __forceinline__ float philox_normal(const NoiseGen &ng, unsigned long long elem) {
    const unsigned long long key = ng.seed ^ 0x9E3779B97F4A7C15ull, e = ng.base + elem;
    unsigned c[4] = {(unsigned)e, (unsigned)(e >> 32), (unsigned)ng.offset, (unsigned)(ng.offset >> 32)};
    unsigned k0 = (unsigned)key, k1 = (unsigned)(key >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        philox_round(c, k0, k1);
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    const float u1 = ((float)c[0] + 0.5f) * 2.3283064e-10f;   // (0, 1]: 2^-32 (x + 1/2), rounded -- never 0
    const float u2 = ((float)c[1] + 0.5f) * 2.3283064e-10f;
    return sqrtf(-2.0f * logf(u1)) * cosf(6.28318548f * u2) * ng.std;
}

struct SampleArgs {
    const float *u_base, *rnd;
    float *samples;
    float div, near_, far_;
    int ns, npad, pdf_padding;
};

template <bool HEAD>
__global__ __launch_bounds__(256) void composite_fwd_kernel(
    const float *__restrict__ raw, int ldr, const float *__restrict__ t_vals, const float *__restrict__ rays,
    const float *__restrict__ noise, const float *__restrict__ mus, int n, int S, int flags,
    float *__restrict__ rgb_map, float *__restrict__ disp, float *__restrict__ acc, float *__restrict__ weights,
    float *__restrict__ depth, float *__restrict__ cdisp, float *__restrict__ rgb_out, int *__restrict__ keep_out, HeadOut ho,
    SampleArgs sa, NoiseGen ng) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int ray = blockIdx.x * COMP_WAVES + wave;
    const bool live = ray < n;
    constexpr int NARR = HEAD ? 12 : 8;
    const int per_wave = NARR * S + ((HEAD && sa.samples) ? 2 * S + 2 + sa.npad : 0);
    float *alpha = smem + (size_t)wave * per_wave;  // [S]
    float *trans = alpha + S;                    // [S]  (1-alpha+1e-10), then exclusive transmittance
    float *wpre = trans + S;                     // [S]  weights before the blender epsilon
    float *wpost = wpre + S;                     // [S]  returned weights
    float *tmp = wpost + S;                      // [S]
    float *rgbs = tmp + S;                       // [3S]
    float *musl = rgbs + 3 * S;                  // [S]  HEAD: the head's mus of this ray
    float *ssigl = musl + S, *spartl = ssigl + S, *sleftl = spartl + S;   // [S] each, HEAD: the smoothed head values (the sampler's inputs)
    const bool white = flags & DDNERF_COMP_WHITE_BKGD, blender = flags & DDNERF_COMP_BLENDER;
    const float *t = t_vals + (size_t)(live ? ray : 0) * (S + 1);

    float sq_m = 0.0f, sq_s = 0.0f;
    if (live) {
        const float *d = rays + 12 * (size_t)ray + 3;
        const float dn = sqrtf((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]);  // :23
        for (int j = lane; j < S; j += 64) {
            const size_t m = (size_t)ray * S + j;
            const float *r = raw + m * ldr;
            float delta = (t[j + 1] - t[j]) * dn;                       // :21-23
            float dens = r[3] + (noise ? noise[m] : (ng.on ? philox_normal(ng, m) : 0.0f));   // :40
            float sig = ddn_softplus(dens - 1.0f);                      // :41
            float a = 1.0f - expf(-sig * delta);                        // :42
            alpha[j] = a;
            trans[j] = 1.0f - a + 1e-10f;                               // :43
            for (int k = 0; k < 3; ++k) {
                float c = ddn_sigmoid(r[k]) * 1.002f - 0.001f;          // :25-27
                rgbs[3 * j + k] = c;
                if (rgb_out) rgb_out[m * 3 + k] = c;
            }
            if constexpr (HEAD) {  // models/models.py:242-260, 266-273 (dd_head_kernel)
                const float rm = r[4], rs = r[5];
                const float mu = ddn_sigmoid(rm), sg = ddn_sigmoid(rs) + 0.001f;
                ho.mus[m] = mu;
                musl[j] = mu;
                ho.sigmas[m] = sg;
                sq_m = sq_m + fabsf(rm) * fabsf(rm);
                sq_s = sq_s + fabsf(rs) * fabsf(rs);
                const float l = ddn_norm_cdf((0.0f - mu) / sg);
                ho.left[m] = l;
                ho.part[m] = ddn_norm_cdf((1.0f - mu) / sg) - l;
                const float ss = sg * ho.smooth;
                ho.ssig[m] = ss;
                ssigl[j] = ss;
                const float sl = ddn_norm_cdf((0.0f - mu) / ss);
                ho.sleft[m] = sl;
                sleftl[j] = sl;
                const float sp = ddn_norm_cdf((1.0f - mu) / ss) - sl;
                ho.spart[m] = sp;
                spartl[j] = sp;
            }
        }
    }
    if constexpr (HEAD) {  // per-block partial sums of the regularisers: dd_head_kernel's 256-thread tree (at S = 64 the very same
        // 256 elements on the same threads: bit-identical partials); they are added up by the records kernel that follows
        __shared__ float red[2][256];
        red[0][threadIdx.x] = sq_m;
        red[1][threadIdx.x] = sq_s;
        __syncthreads();
        for (int st = 128; st > 0; st >>= 1) {
            if ((int)threadIdx.x < st) {
                red[0][threadIdx.x] += red[0][threadIdx.x + st];
                red[1][threadIdx.x] += red[1][threadIdx.x + st];
            }
            __syncthreads();
        }
        if (threadIdx.x == 0) {
            ho.partials[2 * blockIdx.x] = red[0][0];
            ho.partials[2 * blockIdx.x + 1] = red[1][0];
        }
    }
    ddn_wave_sync();
#ifndef COMP_EXP_NOCHAIN   // (timing-only experiment builds, tools/comp_variants.py: what does the serial chain cost?)
    if (live && lane == 0) ddn_chain_cumprod_exclusive(trans, S);  // double accumulator, fp32 prefixes
#endif
    ddn_wave_sync();
    if (live) {
        for (int j = lane; j < S; j += 64) {
            float w = alpha[j] * trans[j];                              // :43
            wpre[j] = w;
            float w2 = (blender && j == S - 1) ? w + 1e-10f : w;        // :52-56
            wpost[j] = w2;
            weights[(size_t)ray * S + j] = w2;
        }
    }
    ddn_wave_sync();
    float c_sum = 0.0f;
#ifndef COMP_EXP_NORGBSUM
    // rgb_map = sum_j w_j * rgb_j, j ascending (:47-48): the products by the whole wave, in place (rgbs is not read again), the three sums by
    // three lanes
    if (live)
        for (int j = lane; j < S; j += 64) {
            const float w = wpre[j];
            rgbs[3 * j] = w * rgbs[3 * j];
            rgbs[3 * j + 1] = w * rgbs[3 * j + 1];
            rgbs[3 * j + 2] = w * rgbs[3 * j + 2];
        }
    ddn_wave_sync();
    if (live && lane < 3) c_sum = ddn_chain_sum_strided(rgbs + lane, 3, S);
#endif
    float wsum = ddn_aten_sum_wave(wpost, S, lane);                     // :58 and :70 (same operand)
    if constexpr (HEAD) {  // models/models.py:292-295 (dd_records_count_kernel): bins with w / sum(w) > 0.1
        if (live) {
            int cnt = 0;
            for (int j = lane; j < S; j += 64) {
                const int f = (wpost[j] / wsum) > 0.1f;  // NaN (an all-zero row) compares false, like torch
                ho.rec_flags[(size_t)ray * S + j] = (unsigned char)f;
                cnt += f;
            }
            for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
            if (lane == 0) ho.rec_counts[ray] = cnt;
        }
    }
    if (live) {
        for (int j = lane; j < S; j += 64) {
            float pdf = blender ? wpost[j] / wsum : wpost[j];           // :58 / :61
            alpha[j] = pdf;                                             // reuse as pdf
            tmp[j] = pdf * ((t[j + 1] + t[j]) / 2.0f);                  // :19, :66
        }
    }
    ddn_wave_sync();
    float dm = ddn_aten_sum_wave(tmp, S, lane);                         // :68
    ddn_wave_sync();
    float cdm = 0.0f;
    const bool has_mus = HEAD || mus != nullptr;
    if (has_mus) {                                                      // :77-83
        if (live)
            for (int j = lane; j < S; j += 64) {
                const float mu = HEAD ? musl[j] : mus[(size_t)ray * S + j];
                tmp[j] = alpha[j] * (t[j] + mu * (t[j + 1] - t[j]));
            }
        ddn_wave_sync();
        cdm = ddn_aten_sum_wave(tmp, S, lane);
    }
    if (live) {
        if (lane < 3) rgb_map[3 * (size_t)ray + lane] = white ? c_sum + (1.0f - wsum) : c_sum;  // :73-74
        if (lane == 0) {
            acc[ray] = wsum;
            disp[ray] = 1.0f / fmaxf(1e-10f, dm / wsum);                // :71
            if (has_mus) {
                if (cdisp) cdisp[ray] = 1.0f / fmaxf(1e-10f, cdm / wsum);  // :82
                depth[ray] = cdm;                                       // :83
            } else {
                depth[ray] = dm;
            }
            // models/dd_utils.py:16 (dpl_keep_kernel): the filter applies to dataset type "blender" only -- its own flag bit
            if (keep_out) keep_out[ray] = (flags & DDNERF_COMP_DP_FILTER) ? (wsum > 1e-10f ? 1 : 0) : 1;
        }
    }
    if constexpr (HEAD) {
        if (sa.samples) {   // the fine fenceposts of this ray (the arrays read here were last written before the wave syncs above)
            ddn_wave_sync();
            float *wp = alpha + NARR * S, *cdf = wp + S, *out = cdf + S + 2;
            const size_t r = live ? ray : 0;
            dd_sample_row(wpost, t, musl, ssigl, spartl, sleftl, sa.u_base, sa.rnd ? sa.rnd + r * sa.ns : nullptr, sa.div, sa.near_, sa.far_,
                          sa.samples + r * sa.ns, (int32_t *)nullptr, S, sa.ns, sa.npad, sa.pdf_padding, wp, cdf, out, lane, live);
        }
    }
}

DDN_EXPORT int ddnerf_composite_forward(const float *raw, int ldr, const float *t_vals, const float *rays,
                                        const float *noise, const float *mus, int n, int S, int flags, float *rgb_map,
                                        float *disp, float *acc, float *weights, float *depth, float *cdisp,
                                        float *rgb, ddnerf_stream_t stream) {
    DDN_REQUIRE(raw && t_vals && rays && rgb_map && disp && acc && weights && depth, DDNERF_E_ARG);
    DDN_REQUIRE(n > 0 && S > 0, DDNERF_E_ARG);
    DDN_REQUIRE(ldr >= 4, DDNERF_E_RANGE);
    size_t lds = (size_t)COMP_WAVES * 8 * S * sizeof(float);
    DDN_REQUIRE(lds <= 64 * 1024, DDNERF_E_RANGE);  // S <= 512
    hipLaunchKernelGGL(composite_fwd_kernel<false>, dim3((n + COMP_WAVES - 1) / COMP_WAVES), dim3(256), lds,
                       (hipStream_t)stream, raw, ldr, t_vals, rays, noise, mus, n, S, flags, rgb_map, disp, acc,
                       weights, depth, cdisp, rgb, (int *)nullptr, HeadOut{}, SampleArgs{}, NoiseGen{});
    return ddn_launch_status();
}

// The fine pass of DDNerfModel (render path): compositing + the dp loss's row filter in one launch.  keep [n] int32 is the
// first array of a ddnerf_dp_loss_workspace_bytes(n) workspace, handed to ddnerf_dp_loss_forward_kept.
DDN_EXPORT int ddnerf_composite_forward_keep_rng(const float *raw, int ldr, const float *t_vals, const float *rays, const float *noise,
                                                 const float *mus, int n, int S, int flags, float *rgb_map, float *disp, float *acc,
                                                 float *weights, float *depth, float *cdisp, void *dp_workspace, unsigned long long noise_seed,
                                                 unsigned long long noise_offset, unsigned long long noise_base, float noise_std,
                                                 ddnerf_stream_t stream) {
    DDN_REQUIRE(raw && t_vals && rays && rgb_map && disp && acc && weights && depth && dp_workspace, DDNERF_E_ARG);
    DDN_REQUIRE(n > 0 && S > 0, DDNERF_E_ARG);
    DDN_REQUIRE(ldr >= 4, DDNERF_E_RANGE);
    size_t lds = (size_t)COMP_WAVES * 8 * S * sizeof(float);
    DDN_REQUIRE(lds <= 64 * 1024, DDNERF_E_RANGE);
    const NoiseGen ng{noise_seed, noise_offset, noise_base, noise_std, (!noise && noise_std > 0.0f) ? 1 : 0};
    hipLaunchKernelGGL(composite_fwd_kernel<false>, dim3((n + COMP_WAVES - 1) / COMP_WAVES), dim3(256), lds, (hipStream_t)stream, raw, ldr,
                       t_vals, rays, noise, mus, n, S, flags, rgb_map, disp, acc, weights, depth, cdisp, (float *)nullptr,
                       (int *)dp_workspace, HeadOut{}, SampleArgs{}, ng);
    return ddn_launch_status();
}
DDN_EXPORT int ddnerf_composite_forward_keep(const float *raw, int ldr, const float *t_vals, const float *rays, const float *noise,
                                             const float *mus, int n, int S, int flags, float *rgb_map, float *disp, float *acc,
                                             float *weights, float *depth, float *cdisp, void *dp_workspace, ddnerf_stream_t stream) {
    return ddnerf_composite_forward_keep_rng(raw, ldr, t_vals, rays, noise, mus, n, S, flags, rgb_map, disp, acc, weights, depth, cdisp, dp_workspace,
                                             0ull, 0ull, 0ull, 0.0f, stream);
}

// the values philox_normal() gives elements [0, count) under (seed, offset, base, std): what the compositing kernels add to the density
// when they draw their noise themselves (test hook; also a way to materialise that noise)
__global__ void philox_normal_kernel(float *__restrict__ out, size_t count, NoiseGen ng) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) out[i] = philox_normal(ng, i);
}
DDN_EXPORT int ddnerf_debug_philox_normal(float *out, long count, unsigned long long seed, unsigned long long offset, unsigned long long base,
                                          float std, ddnerf_stream_t stream) {
    DDN_REQUIRE(out && count > 0, DDNERF_E_ARG);
    hipLaunchKernelGGL(philox_normal_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, (hipStream_t)stream, out, (size_t)count,
                       NoiseGen{seed, offset, base, std, 1});
    return ddn_launch_status();
}

// ---------------------------------------------------------------------------------------------------
// a8  DD head   models/models.py:242-260 (raw sigma) and :266-273 (smoothed sigma)
// Elementwise part + the two L2 regularisers (sum of squares over the whole chunk divided by the number
// of rays).  Reduction is two-stage and order-fixed (no atomics) so training is reproducible.
// ---------------------------------------------------------------------------------------------------
#define DDH_BLOCK 256

__global__ __launch_bounds__(DDH_BLOCK) void dd_head_kernel(const float *__restrict__ raw6, size_t count, float smooth,
                                                            float *__restrict__ mus, float *__restrict__ sigmas,
                                                            float *__restrict__ left, float *__restrict__ part,
                                                            float *__restrict__ ssig, float *__restrict__ sleft,
                                                            float *__restrict__ spart, float *__restrict__ partials) {
    __shared__ float red[2][DDH_BLOCK];
    size_t i = (size_t)blockIdx.x * DDH_BLOCK + threadIdx.x;
    float sq_m = 0.0f, sq_s = 0.0f;
    if (i < count) {
        float rm = raw6[6 * i + 4], rs = raw6[6 * i + 5];
        float mu = ddn_sigmoid(rm), sg = ddn_sigmoid(rs) + 0.001f;       // :245-246
        mus[i] = mu;
        sigmas[i] = sg;
        sq_m = fabsf(rm) * fabsf(rm);                                     // :249
        sq_s = fabsf(rs) * fabsf(rs);                                     // :248
        float l = ddn_norm_cdf((0.0f - mu) / sg);                         // :254-257
        left[i] = l;
        part[i] = ddn_norm_cdf((1.0f - mu) / sg) - l;                     // :258
        float ss = sg * smooth;                                           // :268
        ssig[i] = ss;
        float sl = ddn_norm_cdf((0.0f - mu) / ss);                        // :270-272
        sleft[i] = sl;
        spart[i] = ddn_norm_cdf((1.0f - mu) / ss) - sl;                   // :273
    }
    red[0][threadIdx.x] = sq_m;
    red[1][threadIdx.x] = sq_s;
    __syncthreads();
    for (int s = DDH_BLOCK / 2; s > 0; s >>= 1) {
        if (threadIdx.x < s) {
            red[0][threadIdx.x] += red[0][threadIdx.x + s];
            red[1][threadIdx.x] += red[1][threadIdx.x + s];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        partials[2 * blockIdx.x] = red[0][0];
        partials[2 * blockIdx.x + 1] = red[1][0];
    }
}

__global__ void dd_head_finish_kernel(const float *__restrict__ partials, int nblocks, int n, float dist_reg,
                                      float *__restrict__ scal) {
    // one wave; lane-strided double partials, then a fixed-order tree
    double m = 0.0, s = 0.0;
    for (int b = threadIdx.x; b < nblocks; b += 64) {
        m += (double)partials[2 * b];
        s += (double)partials[2 * b + 1];
    }
    for (int o = 32; o > 0; o >>= 1) {
        m += __shfl_down(m, o);
        s += __shfl_down(s, o);
    }
    if (threadIdx.x == 0) {
        float ml = (float)m / (float)n, sl = (float)s / (float)n;        // :248-249 (rays in chunk)
        scal[0] = ml;
        scal[1] = sl;
        scal[2] = dist_reg * ml;                                          // :251
        scal[3] = dist_reg * sl;                                          // :252
    }
}

DDN_EXPORT size_t ddnerf_dd_head_workspace_floats(int n, int nc) {
    if (n <= 0 || nc <= 0) return 0;
    size_t count = (size_t)n * nc;
    return 2 * ((count + DDH_BLOCK - 1) / DDH_BLOCK);
}

DDN_EXPORT int ddnerf_dd_head(const float *raw6, int n, int nc, float smooth, float dist_reg, float *mus,
                              float *sigmas, float *left, float *part, float *ssig, float *sleft, float *spart,
                              float *scal, float *workspace, ddnerf_stream_t stream) {
    DDN_REQUIRE(raw6 && mus && sigmas && left && part && ssig && sleft && spart && scal && workspace, DDNERF_E_ARG);
    DDN_REQUIRE(n > 0 && nc > 0, DDNERF_E_ARG);
    size_t count = (size_t)n * nc;
    int nblocks = (int)((count + DDH_BLOCK - 1) / DDH_BLOCK);
    hipLaunchKernelGGL(dd_head_kernel, dim3(nblocks), dim3(DDH_BLOCK), 0, (hipStream_t)stream, raw6, count, smooth, mus,
                       sigmas, left, part, ssig, sleft, spart, workspace);
    hipLaunchKernelGGL(dd_head_finish_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, workspace, nblocks, n, dist_reg,
                       scal);
    return ddn_launch_status();
}

// ---------------------------------------------------------------------------------------------------
// K3b  backward of volume_render_radiance_field w.r.t. the raw network outputs.
// Upstream gradients that exist in the reference's graph: d(rgb_map) from the MSE losses and
// d(weights) from the dp loss (coarse level; NULL otherwise).  disp / acc / depth / corrected_disp are not
// differentiated (nothing in the reference's losses reads them).
//   c_jk = sigmoid(r_jk)*1.002 - 0.001;  sigma_j = softplus(r_j3 + noise - 1);  a_j = 1 - exp(-sigma_j d_j)
//   T_j = prod_{i<j} (1 - a_i + 1e-10);  w_j = a_j T_j;  rgb_map_k = sum_j w_j c_jk (+ 1 - sum_j w_j if white)
//   dL/dw_j  = sum_k G_k c_jk + gw_j - (white ? sum_k G_k : 0)
//   dL/da_j  = T_j dL/dw_j - (sum_{i>j} dL/dw_i w_i) / (1 - a_j + 1e-10)
//   dL/dr_j3 = dL/da_j * d_j (1 - a_j) * sigmoid(r_j3 + noise - 1);   dL/dr_jk = G_k w_j * 1.002 s(1-s)
// One wave per ray; the suffix sum is a wave-level reverse scan.  g_raw has row stride ldr; columns >= 4 are
// written as zero (so the DD-head gradient can simply be added).
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void composite_bwd_kernel(
    const float *__restrict__ raw, int ldr, const float *__restrict__ t_vals, const float *__restrict__ rays,
    const float *__restrict__ noise, int n, int S, int flags, const float *__restrict__ g_rgb_map,
    const float *__restrict__ g_weights, float *__restrict__ g_raw) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int ray = blockIdx.x * COMP_WAVES + wave;
    const bool live = ray < n;
    float *alpha = smem + (size_t)wave * 4 * S;  // [S]
    float *trans = alpha + S;                    // [S] exclusive transmittance
    float *dw = trans + S;                       // [S] dL/dw
    float *suf = dw + S;                         // [S] dL/dw_j * w_j, then its exclusive suffix sum
    const bool white = flags & DDNERF_COMP_WHITE_BKGD;
    const size_t r0 = live ? ray : 0;
    const float *t = t_vals + r0 * (S + 1);
    const float *d = rays + 12 * r0 + 3;
    const float dn = sqrtf((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]);
    const float G0 = g_rgb_map[3 * r0], G1 = g_rgb_map[3 * r0 + 1], G2 = g_rgb_map[3 * r0 + 2];
    const float Gsum = white ? (G0 + G1) + G2 : 0.0f;

    if (live)
        for (int j = lane; j < S; j += 64) {
            const size_t m = r0 * S + j;
            float dens = raw[m * ldr + 3] + (noise ? noise[m] : 0.0f);
            float sig = ddn_softplus(dens - 1.0f);
            float a = 1.0f - expf(-sig * ((t[j + 1] - t[j]) * dn));
            alpha[j] = a;
            trans[j] = 1.0f - a + 1e-10f;
        }
    ddn_wave_sync();
    if (live && lane == 0) ddn_chain_cumprod_exclusive(trans, S);  // same transmittance arithmetic as the forward kernel
    ddn_wave_sync();
    if (live)
        for (int j = lane; j < S; j += 64) {
            const size_t m = r0 * S + j;
            const float *r = raw + m * ldr;
            float w = alpha[j] * trans[j];
            float s0 = ddn_sigmoid(r[0]), s1 = ddn_sigmoid(r[1]), s2 = ddn_sigmoid(r[2]);
            float c0 = s0 * 1.002f - 0.001f, c1 = s1 * 1.002f - 0.001f, c2 = s2 * 1.002f - 0.001f;
            float dLdw = (G0 * c0 + G1 * c1 + G2 * c2) - Gsum + (g_weights ? g_weights[m] : 0.0f);
            dw[j] = dLdw;
            suf[j] = dLdw * w;
            float *g = g_raw + m * ldr;
            g[0] = G0 * w * 1.002f * s0 * (1.0f - s0);
            g[1] = G1 * w * 1.002f * s1 * (1.0f - s1);
            g[2] = G2 * w * 1.002f * s2 * (1.0f - s2);
            for (int k = 4; k < ldr; ++k) g[k] = 0.0f;
        }
    ddn_wave_sync();
    // exclusive suffix sum of suf[] by the wave: chunks of 64 from the back, carry across chunks
    float carry = 0.0f;
    const int nchunk = (S + 63) / 64;
    for (int c = nchunk - 1; c >= 0; --c) {
        int j = c * 64 + lane;
        float v = (live && j < S) ? suf[j] : 0.0f;
        float inc = v;  // inclusive suffix within the chunk
        for (int o = 1; o < 64; o <<= 1) {
            float up = __shfl_down(inc, o);
            if (lane + o < 64) inc += up;
        }
        float excl = inc - v + carry;
        if (live && j < S) suf[j] = excl;
        carry += __shfl(inc, 0);
    }
    ddn_wave_sync();
    if (live)
        for (int j = lane; j < S; j += 64) {
            const size_t m = r0 * S + j;
            float a = alpha[j], om = 1.0f - a + 1e-10f;
            float dLda = trans[j] * dw[j] - suf[j] / om;
            float dens = raw[m * ldr + 3] + (noise ? noise[m] : 0.0f);
            float delta = (t[j + 1] - t[j]) * dn;
            g_raw[m * ldr + 3] = dLda * delta * (1.0f - a) * ddn_sigmoid(dens - 1.0f);
        }
}

DDN_EXPORT int ddnerf_composite_backward(const float *raw, int ldr, const float *t_vals, const float *rays,
                                         const float *noise, int n, int S, int flags, const float *g_rgb_map,
                                         const float *g_weights, float *g_raw, ddnerf_stream_t stream) {
    DDN_REQUIRE(raw && t_vals && rays && g_rgb_map && g_raw, DDNERF_E_ARG);
    DDN_REQUIRE(n > 0 && S > 0, DDNERF_E_ARG);
    DDN_REQUIRE(ldr >= 4, DDNERF_E_RANGE);
    size_t lds = (size_t)COMP_WAVES * 4 * S * sizeof(float);
    DDN_REQUIRE(lds <= 64 * 1024, DDNERF_E_RANGE);
    hipLaunchKernelGGL(composite_bwd_kernel, dim3((n + COMP_WAVES - 1) / COMP_WAVES), dim3(256), lds,
                       (hipStream_t)stream, raw, ldr, t_vals, rays, noise, n, S, flags, g_rgb_map, g_weights, g_raw);
    return ddn_launch_status();
}

// ---------------------------------------------------------------------------------------------------
// backward of the DD head w.r.t. raw[..., 4:6]  (models/models.py:245-252)
//   mus = sigmoid(rm); sigmas = sigmoid(rs) + 1e-3; mus_loss = sum rm^2 / n; sig_loss = sum rs^2 / n;
//   mus_reg = c mus_loss; sig_reg = c sig_loss.     g_scal = upstream grads of {mus_loss, sig_loss, mus_reg, sig_reg}
// Adds into g_raw6[..., 4] and [..., 5] (the compositing backward has written zeros there).
// ---------------------------------------------------------------------------------------------------
__global__ void dd_head_bwd_kernel(const float *__restrict__ raw6, size_t count, int n, float dist_reg,
                                   const float *__restrict__ g_mus, const float *__restrict__ g_sigmas,
                                   const float *__restrict__ g_scal, float *__restrict__ g_raw6) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    float rm = raw6[6 * i + 4], rs = raw6[6 * i + 5];
    float mu = ddn_sigmoid(rm), sg = ddn_sigmoid(rs);
    float cm = 0.0f, cs = 0.0f;
    if (g_scal) {
        cm = (g_scal[0] + dist_reg * g_scal[2]) * 2.0f / (float)n;
        cs = (g_scal[1] + dist_reg * g_scal[3]) * 2.0f / (float)n;
    }
    g_raw6[6 * i + 4] += (g_mus ? g_mus[i] * mu * (1.0f - mu) : 0.0f) + cm * rm;
    g_raw6[6 * i + 5] += (g_sigmas ? g_sigmas[i] * sg * (1.0f - sg) : 0.0f) + cs * rs;
}

DDN_EXPORT int ddnerf_dd_head_backward(const float *raw6, int n, int nc, float dist_reg, const float *g_mus,
                                       const float *g_sigmas, const float *g_scal, float *g_raw6,
                                       ddnerf_stream_t stream) {
    DDN_REQUIRE(raw6 && g_raw6, DDNERF_E_ARG);
    DDN_REQUIRE(n > 0 && nc > 0, DDNERF_E_ARG);
    size_t count = (size_t)n * nc;
    hipLaunchKernelGGL(dd_head_bwd_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, (hipStream_t)stream, raw6,
                       count, n, dist_reg, g_mus, g_sigmas, g_scal, g_raw6);
    return ddn_launch_status();
}

// ---------------------------------------------------------------------------------------------------
// DD records (models/models.py:292-295): mus / sigmas / smoothed sigmas of the coarse bins whose level-0 pdf
// w / sum(w) exceeds 0.1, as flat tensors in row-major order (what `x[pdf > 0.1]` returns).  The sizes are data
// dependent; three small kernels replace the dozen torch launches of sum, divide, compare, nonzero and three gathers:
//   count: wave per ray -- the row sum in torch.sum's order, IEEE divide, one flag byte per bin, the row's count
//   scan : one block    -- exclusive scan of the row counts, the total
//   write: wave per ray -- ordered writes of the three records
// The caller allocates the outputs at capacity n * nc and reads `total` when it needs the shapes (its only host sync).
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void dd_records_count_kernel(const float *__restrict__ w, int n, int nc,
                                                              unsigned char *__restrict__ flags, int *__restrict__ counts) {
    const int lane = threadIdx.x & 63, ray = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (ray >= n) return;
    const float *row = w + (size_t)ray * nc;
    const float s = ddn_aten_sum_wave(row, nc, lane);
    int cnt = 0;
    for (int j = lane; j < nc; j += 64) {
        const int f = (row[j] / s) > 0.1f;  // NaN (an all-zero row) compares false, like torch
        flags[(size_t)ray * nc + j] = (unsigned char)f;
        cnt += f;
    }
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
    if (lane == 0) counts[ray] = cnt;
}


// Every block finds the output offset of its own rays by counting what the rays in front of it keep (integer sums over the
// count kernel's per-ray counts: no scan launch in between); the block that owns the last ray also writes the total.
// (One block more than the rays need, when `partials` is given: its first wave adds up the regulariser partial sums the fused
// coarse kernel left behind -- dd_head_finish_kernel's arithmetic -- into scal[4].)
__global__ __launch_bounds__(256) void dd_records_write_kernel(const unsigned char *__restrict__ flags,
                                                              const int *__restrict__ counts, const float *__restrict__ mus,
                                                              const float *__restrict__ sigmas, const float *__restrict__ ssig,
                                                              int n, int nc, float *__restrict__ out_mus,
                                                              float *__restrict__ out_sigmas, float *__restrict__ out_ssig,
                                                              int *__restrict__ total, const float *__restrict__ partials,
                                                              int npartials, float dist_reg, float *__restrict__ scal) {
    __shared__ int part[4];
    if (partials && blockIdx.x == gridDim.x - 1) {
        if (threadIdx.x < 64) {
            double m = 0.0, s2 = 0.0;
            for (int b = threadIdx.x; b < npartials; b += 64) {
                m += (double)partials[2 * b];
                s2 += (double)partials[2 * b + 1];
            }
            for (int o = 32; o > 0; o >>= 1) {
                m += __shfl_down(m, o);
                s2 += __shfl_down(s2, o);
            }
            if (threadIdx.x == 0) {
                float ml = (float)m / (float)n, sl = (float)s2 / (float)n;
                scal[0] = ml;
                scal[1] = sl;
                scal[2] = dist_reg * ml;
                scal[3] = dist_reg * sl;
            }
        }
        return;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, ray = blockIdx.x * 4 + wave;
    const int first = blockIdx.x * 4;
    int before = 0;
    for (int i = threadIdx.x; i < first; i += 256) before += counts[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) before += __shfl_xor(before, o);
    if (lane == 0) part[wave] = before;
    __syncthreads();
    int base = part[0] + part[1] + part[2] + part[3];
    for (int i = first; i < ray && i < n; ++i) base += counts[i];
    if (ray == n - 1 && lane == 0) *total = base + counts[ray];
    if (ray >= n) return;
    for (int j0 = 0; j0 < nc; j0 += 64) {
        const int j = j0 + lane;
        const bool f = j < nc && flags[(size_t)ray * nc + j];
        const unsigned long long m = __ballot(f);
        if (f) {
            const int dst = base + __popcll(m & ((1ull << lane) - 1ull));
            const size_t src = (size_t)ray * nc + j;
            out_mus[dst] = mus[src];
            out_sigmas[dst] = sigmas[src];
            out_ssig[dst] = ssig[src];
        }
        base += __popcll(m);
    }
}

DDN_EXPORT size_t ddnerf_dd_records_workspace_bytes(int n, int nc) {
    return (size_t)n * nc + (size_t)(2 * n + 1) * sizeof(int) + 16;
}

DDN_EXPORT int ddnerf_dd_records(const float *weights, const float *mus, const float *sigmas, const float *ssig, int n, int nc,
                                 float *out_mus, float *out_sigmas, float *out_ssig, int *total, void *workspace,
                                 ddnerf_stream_t stream) {
    DDN_REQUIRE(weights && mus && sigmas && ssig && out_mus && out_sigmas && out_ssig && total && workspace, DDNERF_E_ARG);
    DDN_REQUIRE(n > 0 && nc > 0, DDNERF_E_ARG);
    hipStream_t st = (hipStream_t)stream;
    int *counts = (int *)workspace, *offsets = counts + n;
    unsigned char *flags = (unsigned char *)(offsets + n + 1);
    dim3 grid((unsigned)((n + 3) / 4));
    hipLaunchKernelGGL(dd_records_count_kernel, grid, dim3(256), 0, st, weights, n, nc, flags, counts);
    hipLaunchKernelGGL(dd_records_write_kernel, grid, dim3(256), 0, st, flags, counts, mus, sigmas, ssig, n, nc, out_mus,
                       out_sigmas, out_ssig, total, (const float *)nullptr, 0, 0.0f, (float *)nullptr);
    return ddn_launch_status();
}

// The coarse pass of DDNerfModel behind the MLP, render path (models/models.py:242-295), as TWO launches instead of five:
//   1. the DD head (ddnerf_dd_head), the compositing with the head's mus (ddnerf_composite_forward) and the flags / counts of
//      the level-0 records, wave per ray;
//   2. the records' ordered writes (ddnerf_dd_records) and the regularisers' final sums -> scal[4].
// Outputs are bit for bit those of the three separate entry points at nc = 64 (the regulariser partial sums cover the same 256
// elements per block; another nc is another -- equally fixed -- summation order).  workspace:
// ddnerf_dd_coarse_workspace_bytes(n, nc).
DDN_EXPORT size_t ddnerf_dd_coarse_workspace_bytes(int n, int nc) {
    if (n <= 0 || nc <= 0) return 0;
    return ddnerf_dd_records_workspace_bytes(n, nc) + 16 + 2 * sizeof(float) * (size_t)((n + COMP_WAVES - 1) / COMP_WAVES);
}

// (noise == NULL and noise_std > 0: the density noise is drawn in the kernel, see NoiseGen)
// ... and with samples != NULL also the fine pass's fenceposts [n, ns] (ddnerf_sample_pdf_mu_sigma's arguments u_base [ns], rnd [n, ns] or
// NULL, near_, far_, pdf_padding; bit for bit that entry point's output on this launch's weights / mus / smoothed head values).
DDN_EXPORT int ddnerf_dd_coarse_sample_forward(const float *raw6, const float *t_vals, const float *rays, const float *noise, int n, int nc,
                                               int flags, float smooth, float dist_reg, float *mus, float *sigmas, float *left,
                                               float *part, float *ssig, float *sleft, float *spart, float *scal, float *rgb_map,
                                               float *disp, float *acc, float *weights, float *depth, float *cdisp, float *rec_mus,
                                               float *rec_sigmas, float *rec_ssig, int *rec_total, void *workspace, const float *u_base,
                                               const float *rnd, float near_, float far_, float *samples, int ns, int pdf_padding,
                                               unsigned long long noise_seed, unsigned long long noise_offset, unsigned long long noise_base,
                                               float noise_std, ddnerf_stream_t stream) {
    DDN_REQUIRE(raw6 && t_vals && rays && mus && sigmas && left && part && ssig && sleft && spart && scal, DDNERF_E_ARG);
    DDN_REQUIRE(rgb_map && disp && acc && weights && depth && cdisp && rec_mus && rec_sigmas && rec_ssig && rec_total && workspace, DDNERF_E_ARG);
    DDN_REQUIRE(n > 0 && nc > 0, DDNERF_E_ARG);
    SampleArgs sa{};
    size_t per_wave = (size_t)12 * nc;
    if (samples) {
        DDN_REQUIRE(u_base && ns > 1, DDNERF_E_ARG);
        int npad = 2;
        while (npad < ns) npad <<= 1;
        sa = SampleArgs{u_base, rnd, samples, (float)((double)ns + 1e-5), near_, far_, ns, npad, pdf_padding};
        per_wave += (size_t)2 * nc + 2 + npad;
    }
    const size_t lds = (size_t)COMP_WAVES * per_wave * sizeof(float);
    DDN_REQUIRE(lds <= 60 * 1024, DDNERF_E_RANGE);
    hipStream_t st = (hipStream_t)stream;
    int *counts = (int *)workspace, *offsets = counts + n;
    unsigned char *rflags = (unsigned char *)(offsets + n + 1);
    const int nblocks = (n + COMP_WAVES - 1) / COMP_WAVES;
    float *partials = (float *)(((size_t)(rflags + (size_t)n * nc) + 15) & ~(size_t)15);
    HeadOut ho = {mus, sigmas, left, part, ssig, sleft, spart, partials, rflags, counts, smooth};
    const NoiseGen ng{noise_seed, noise_offset, noise_base, noise_std, (!noise && noise_std > 0.0f) ? 1 : 0};
    hipLaunchKernelGGL(composite_fwd_kernel<true>, dim3(nblocks), dim3(256), lds, st, raw6, 6, t_vals, rays, noise, (const float *)nullptr, n,
                       nc, flags, rgb_map, disp, acc, weights, depth, cdisp, (float *)nullptr, (int *)nullptr, ho, sa, ng);
    hipLaunchKernelGGL(dd_records_write_kernel, dim3((unsigned)((n + 3) / 4 + 1)), dim3(256), 0, st, rflags, counts, mus, sigmas, ssig, n, nc,
                       rec_mus, rec_sigmas, rec_ssig, rec_total, partials, nblocks, dist_reg, scal);
    return ddn_launch_status();
}

DDN_EXPORT int ddnerf_dd_coarse_forward(const float *raw6, const float *t_vals, const float *rays, const float *noise, int n, int nc,
                                        int flags, float smooth, float dist_reg, float *mus, float *sigmas, float *left,
                                        float *part, float *ssig, float *sleft, float *spart, float *scal, float *rgb_map,
                                        float *disp, float *acc, float *weights, float *depth, float *cdisp, float *rec_mus,
                                        float *rec_sigmas, float *rec_ssig, int *rec_total, void *workspace, ddnerf_stream_t stream) {
    return ddnerf_dd_coarse_sample_forward(raw6, t_vals, rays, noise, n, nc, flags, smooth, dist_reg, mus, sigmas, left, part, ssig, sleft, spart, scal,
                                           rgb_map, disp, acc, weights, depth, cdisp, rec_mus, rec_sigmas, rec_ssig, rec_total, workspace,
                                           (const float *)nullptr, (const float *)nullptr, 0.0f, 0.0f, (float *)nullptr, 0, 0, 0ull, 0ull, 0ull, 0.0f, stream);
}
