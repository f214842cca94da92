// Ray packing, first-cycle sampling and the fused cone-cast + IPE + view-dir encoding kernel (K1).
// Compiled with -ffp-contract=off: the Gaussian mean feeds sin(x * 2^l) with l up to 15, so the
// reference's operation order must be kept bit for bit (SURVEY.md 7 "hard parts").
#include "mlp_bf16_common.h"   // (korder32: the 16-bit rows' column order; includes common.h)

// ---------------------------------------------------------------------------------------------------
// a1  GeneralMipNerfModel.get_rays_batches   models/models.py:144-162
// ---------------------------------------------------------------------------------------------------
__global__ void pack_rays_kernel(const float *__restrict__ ro, const float *__restrict__ rd,
                                 const float *__restrict__ rad, float near_, float far_, float *__restrict__ rays,
                                 int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float d0 = rd[3 * i], d1 = rd[3 * i + 1], d2 = rd[3 * i + 2];
    float nrm = sqrtf((d0 * d0 + d1 * d1) + d2 * d2);
    float *r = rays + 12 * (size_t)i;
    r[0] = ro[3 * i]; r[1] = ro[3 * i + 1]; r[2] = ro[3 * i + 2];
    r[3] = d0; r[4] = d1; r[5] = d2;
    r[6] = rad[i];
    r[7] = near_; r[8] = far_;
    r[9] = d0 / nrm; r[10] = d1 / nrm; r[11] = d2 / nrm;
}

DDN_EXPORT int ddnerf_pack_rays(const float *origins, const float *directions, const float *radii, float near_,
                                float far_, float *rays, int n, ddnerf_stream_t stream) {
    DDN_REQUIRE(origins && directions && radii && rays, DDNERF_E_ARG);
    DDN_REQUIRE(n > 0, DDNERF_E_ARG);
    hipLaunchKernelGGL(pack_rays_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, origins, directions,
                       radii, near_, far_, rays, n);
    return ddn_launch_status();
}

// ---------------------------------------------------------------------------------------------------
// a2  sample_first_cycle   models/samplers.py:30-62.  One thread per fencepost.
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ float first_cycle_t(float nr, float fr, float tl, int lindisp) {
    if (lindisp == 2) return tl;                                 // :45-49 get_combined_samples: the row holds absolute depths
    if (!lindisp) return nr * (1.0f - tl) + fr * tl;             // :40
    return 1.0f / (1.0f / nr * (1.0f - tl) + 1.0f / fr * tl);    // :42
}

__global__ void first_cycle_kernel(const float *__restrict__ rays, const float *__restrict__ t_lin,
                                   const float *__restrict__ t_rand, float *__restrict__ t_vals, int n, int nc,
                                   int lindisp) {
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    int s1 = nc + 1;
    if (idx >= (size_t)n * s1) return;
    int i = (int)(idx / s1), j = (int)(idx % s1);
    float nr = rays[12 * (size_t)i + 7], fr = rays[12 * (size_t)i + 8];
    float t = first_cycle_t(nr, fr, t_lin[j], lindisp);
    if (t_rand) {  // :52-60
        float lower, upper;
        if (j == 0) lower = t;
        else lower = 0.5f * (t + first_cycle_t(nr, fr, t_lin[j - 1], lindisp));
        if (j == nc) upper = t;
        else upper = 0.5f * (first_cycle_t(nr, fr, t_lin[j + 1], lindisp) + t);
        t = lower + (upper - lower) * t_rand[idx];
        if (j == 0) t = nr;
        if (j == nc) t = fr;
    }
    t_vals[idx] = t;
}

DDN_EXPORT int ddnerf_sample_first_cycle(const float *rays, const float *t_lin, const float *t_rand, float *t_vals,
                                         int n, int nc, int lindisp, ddnerf_stream_t stream) {
    DDN_REQUIRE(rays && t_lin && t_vals, DDNERF_E_ARG);
    DDN_REQUIRE(n > 0 && nc > 0, DDNERF_E_ARG);
    DDN_REQUIRE(lindisp >= 0 && lindisp <= 2, DDNERF_E_RANGE);
    size_t total = (size_t)n * (nc + 1);
    hipLaunchKernelGGL(first_cycle_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       rays, t_lin, t_rand, t_vals, n, nc, lindisp);
    return ddn_launch_status();
}

// a1 + a2 in one launch (a ray batch that is ONE chunk: the benchmark's and most training batches): thread (ray i, fencepost j)
// computes t_vals[i][j] from the scalar near / far the packed row would hold, the j = 0 thread also packs row i.  Same
// arithmetic, same outputs as the two kernels above.
__global__ void pack_first_cycle_kernel(const float *__restrict__ ro, const float *__restrict__ rd, const float *__restrict__ rad,
                                        float near_, float far_, const float *__restrict__ t_lin, const float *__restrict__ t_rand,
                                        float *__restrict__ rays, float *__restrict__ t_vals, int n, int nc, int lindisp) {
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    int s1 = nc + 1;
    if (idx >= (size_t)n * s1) return;
    int i = (int)(idx / s1), j = (int)(idx % s1);
    if (j == 0) {
        float d0 = rd[3 * i], d1 = rd[3 * i + 1], d2 = rd[3 * i + 2];
        float nrm = sqrtf((d0 * d0 + d1 * d1) + d2 * d2);
        float *r = rays + 12 * (size_t)i;
        r[0] = ro[3 * i]; r[1] = ro[3 * i + 1]; r[2] = ro[3 * i + 2];
        r[3] = d0; r[4] = d1; r[5] = d2;
        r[6] = rad[i];
        r[7] = near_; r[8] = far_;
        r[9] = d0 / nrm; r[10] = d1 / nrm; r[11] = d2 / nrm;
    }
    const float nr = near_, fr = far_;
    float t = first_cycle_t(nr, fr, t_lin[j], lindisp);
    if (t_rand) {
        float lower, upper;
        if (j == 0) lower = t;
        else lower = 0.5f * (t + first_cycle_t(nr, fr, t_lin[j - 1], lindisp));
        if (j == nc) upper = t;
        else upper = 0.5f * (first_cycle_t(nr, fr, t_lin[j + 1], lindisp) + t);
        t = lower + (upper - lower) * t_rand[idx];
        if (j == 0) t = nr;
        if (j == nc) t = fr;
    }
    t_vals[idx] = t;
}

__device__ __forceinline__ unsigned short f32_to_bf16(float f) {
    __bf16 b = (__bf16)f;  // v_cvt_pk_bf16_f32: round-to-nearest-even, NaN preserved
    return __builtin_bit_cast(unsigned short, b);
}
// 16-bit feature rows: KIND 1 = bf16, KIND 2 = fp16 (the fp16 MLP tier: 11 significant bits; |features| <= 1 and the view direction's
// components are far inside fp16's range, tiny damped values go through its subnormals or to zero: absolute error < 6e-8)
template <int KIND>
__device__ __forceinline__ unsigned short f32_to_h16(float f) {
    if constexpr (KIND == 2) return __builtin_bit_cast(unsigned short, (_Float16)f);  // round-to-nearest-even
    else return f32_to_bf16(f);
}


// ---------------------------------------------------------------------------------------------------
// The per-RAY table of the fused encoder + MLP kernel (mlp_bf16_g2e.hip): what encode_kernel<1> derives from a packed ray row once
// per sample (phase 1: d^2, the null-space factor 1 - d^2 / |d|^2 of lift_gaussian, general_utils/math_utils.py:34-54) or once per
// block (phase 1b: the view directions' positional encoding, general_utils/nerf_helpers.py:127-171), once per ray: 16 floats
// [o 0:3 | d 3:6 | radius^2 6 | d^2 7:10 | 1 - d^2/|d|^2 10:13 | 0 0 0], then the ray's 32 view-direction columns as a bf16 / fp16 row in
// MFMA k-order (the same arithmetic, operation for operation: the fused kernel's outputs equal the two-launch path's bit for bit).
// ---------------------------------------------------------------------------------------------------
#define DDN_RAY_TABLE_FLOATS 32
template <int KIND>   // 1: the view-direction row as bf16, 2: as fp16 (the two 16-bit MLP tiers)
__device__ __forceinline__ void ray_table_row(const float *__restrict__ r, float *__restrict__ t) {
    const float d0 = r[3], d1 = r[4], d2 = r[5];
    const float q0 = d0 * d0, q1 = d1 * d1, q2 = d2 * d2;
    const float dmag = fmaxf(1e-10f, (q0 + q1) + q2);
    t[0] = r[0], t[1] = r[1], t[2] = r[2];
    t[3] = d0, t[4] = d1, t[5] = d2;
    t[6] = r[6] * r[6];
    t[7] = q0, t[8] = q1, t[9] = q2;
    t[10] = 1.0f - q0 / dmag, t[11] = 1.0f - q1 / dmag, t[12] = 1.0f - q2 / dmag;
    t[13] = t[14] = t[15] = 0.0f;
    float dirv[32];
#pragma unroll
    for (int k = 27; k < 32; ++k) dirv[k] = 0.0f;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float v = r[9 + a];
        dirv[a] = v;
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            const float x = v * (float)(1 << f);
            dirv[3 + f * 6 + a] = __builtin_amdgcn_sinf(x * 0.15915494f);                    // (encode_kernel<1>: fast_sin)
            dirv[3 + f * 6 + 3 + a] = __builtin_amdgcn_sinf((x + 1.57079637f) * 0.15915494f);
        }
    }
    unsigned short *row = (unsigned short *)(t + 16);
#pragma unroll
    for (int p = 0; p < 32; ++p) row[p] = f32_to_h16<KIND>(dirv[korder32(p)]);
}

template <int KIND>
__global__ void ray_table_kernel(const float *__restrict__ rays, int n, float *__restrict__ table) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) ray_table_row<KIND>(rays + 12 * (size_t)i, table + DDN_RAY_TABLE_FLOATS * (size_t)i);
}

DDN_EXPORT size_t ddnerf_ray_table_bytes(int n) { return (size_t)(n > 0 ? n : 0) * DDN_RAY_TABLE_FLOATS * sizeof(float); }
DDN_EXPORT int ddnerf_ray_table(const float *rays, int n, int feat_dtype, void *table, ddnerf_stream_t stream) {
    DDN_REQUIRE(rays && table, DDNERF_E_ARG);
    DDN_REQUIRE(n > 0, DDNERF_E_ARG);
    DDN_REQUIRE(feat_dtype == 1 || feat_dtype == 2, DDNERF_E_RANGE);
    DDN_REQUIRE(ddn_aligned(table, 128), DDNERF_E_ALIGN);
    if (feat_dtype == 1)
        hipLaunchKernelGGL(ray_table_kernel<1>, dim3((n + 127) / 128), dim3(128), 0, (hipStream_t)stream, rays, n, (float *)table);
    else
        hipLaunchKernelGGL(ray_table_kernel<2>, dim3((n + 127) / 128), dim3(128), 0, (hipStream_t)stream, rays, n, (float *)table);
    return ddn_launch_status();
}

// a1 + a2 + the ray table in ONE launch (the head of a one-chunk render pass on the fused bf16 path): pack_first_cycle_kernel without
// jitter, whose j = 0 thread also writes its ray's table row from the values it has just packed -- rays, t_vals and table bit for bit
// those of ddnerf_pack_rays_first_cycle (t_rand = NULL) followed by ddnerf_ray_table.
template <int KIND>
__global__ void pack_first_cycle_table_kernel(const float *__restrict__ ro, const float *__restrict__ rd, const float *__restrict__ rad,
                                              float near_, float far_, const float *__restrict__ t_lin, float *__restrict__ rays,
                                              float *__restrict__ t_vals, float *__restrict__ table, int n, int nc, int lindisp) {
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    int s1 = nc + 1;
    if (idx >= (size_t)n * s1) return;
    int i = (int)(idx / s1), j = (int)(idx % s1);
    if (j == 0) {
        float d0 = rd[3 * i], d1 = rd[3 * i + 1], d2 = rd[3 * i + 2];
        float nrm = sqrtf((d0 * d0 + d1 * d1) + d2 * d2);
        float r[12];
        r[0] = ro[3 * i]; r[1] = ro[3 * i + 1]; r[2] = ro[3 * i + 2];
        r[3] = d0; r[4] = d1; r[5] = d2;
        r[6] = rad[i];
        r[7] = near_; r[8] = far_;
        r[9] = d0 / nrm; r[10] = d1 / nrm; r[11] = d2 / nrm;
        float *dst = rays + 12 * (size_t)i;
#pragma unroll
        for (int k = 0; k < 12; ++k) dst[k] = r[k];
        ray_table_row<KIND>(r, table + DDN_RAY_TABLE_FLOATS * (size_t)i);
    }
    t_vals[idx] = first_cycle_t(near_, far_, t_lin[j], lindisp);
}

DDN_EXPORT int ddnerf_pack_rays_first_cycle_table(const float *origins, const float *directions, const float *radii, float near_, float far_,
                                                  const float *t_lin, float *rays, float *t_vals, int feat_dtype, void *table, int n, int nc,
                                                  int lindisp, ddnerf_stream_t stream) {
    DDN_REQUIRE(origins && directions && radii && t_lin && rays && t_vals && table, DDNERF_E_ARG);
    DDN_REQUIRE(n > 0 && nc > 0, DDNERF_E_ARG);
    DDN_REQUIRE(lindisp >= 0 && lindisp <= 2 && (feat_dtype == 1 || feat_dtype == 2), DDNERF_E_RANGE);
    DDN_REQUIRE(ddn_aligned(table, 128), DDNERF_E_ALIGN);
    size_t total = (size_t)n * (nc + 1);
    if (feat_dtype == 1)
        hipLaunchKernelGGL(pack_first_cycle_table_kernel<1>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, origins,
                           directions, radii, near_, far_, t_lin, rays, t_vals, (float *)table, n, nc, lindisp);
    else
        hipLaunchKernelGGL(pack_first_cycle_table_kernel<2>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, origins,
                           directions, radii, near_, far_, t_lin, rays, t_vals, (float *)table, n, nc, lindisp);
    return ddn_launch_status();
}

DDN_EXPORT int ddnerf_pack_rays_first_cycle(const float *origins, const float *directions, const float *radii, float near_, float far_,
                                            const float *t_lin, const float *t_rand, float *rays, float *t_vals, int n, int nc,
                                            int lindisp, ddnerf_stream_t stream) {
    DDN_REQUIRE(origins && directions && radii && t_lin && rays && t_vals, DDNERF_E_ARG);
    DDN_REQUIRE(n > 0 && nc > 0, DDNERF_E_ARG);
    DDN_REQUIRE(lindisp >= 0 && lindisp <= 2, DDNERF_E_RANGE);
    size_t total = (size_t)n * (nc + 1);
    hipLaunchKernelGGL(pack_first_cycle_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, origins,
                       directions, radii, near_, far_, t_lin, t_rand, rays, t_vals, n, nc, lindisp);
    return ddn_launch_status();
}

// ---------------------------------------------------------------------------------------------------
// K1  encode: cast_rays -> conical_frustum_to_gaussian(stable) | cylinder_to_gaussian -> lift_gaussian(diag)
//     -> integrated_pos_enc(16 octaves) -> cat(view-dir positional encoding)
//     general_utils/math_utils.py:7-166, general_utils/nerf_helpers.py:127-171, models/models.py:124-133
//
// Work split: one thread per (sample, octave*3+axis) pair, 48 threads per sample: the thread evaluates the
// shared damping exp(-yv/2) once and both the sin and the cos-block feature.  A 256-thread block covers
// 16 samples x 16 "slots"(= 3 features each); the per-sample Gaussian (mean[3], cov[3]) is computed once per
// sample by 16 threads and shared through LDS.  Rows are written 128 columns wide so that the MLP kernel can
// fetch them as aligned 16-byte pieces; stores are issued column-contiguous per sample (coalesced 512 B rows).
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ void gaussian_of_interval(float t0, float t1, float rad, int cylinder, float &t_mean,
                                                     float &t_var, float &r_var) {
    if (!cylinder) {
        float mu = (t0 + t1) / 2.0f, hw = (t1 - t0) / 2.0f;
        float mu2 = mu * mu, hw2 = hw * hw, hw4 = hw2 * hw2;
        float den = 3.0f * mu2 + hw2;
        t_mean = mu + (2.0f * mu * hw2) / den;                                                      // :78
        t_var = hw2 / 3.0f - 0.266666681f * ((hw4 * (12.0f * mu2 - hw2)) / (den * den));            // :79-80
        r_var = (rad * rad) * ((mu2 / 4.0f + 0.416666657f * hw2) - 0.266666681f * hw4 / den);       // :81-82
    } else {
        float dt = t1 - t0;
        t_mean = (t0 + t1) / 2.0f;   // :107
        r_var = (rad * rad) / 4.0f;  // :108
        t_var = (dt * dt) / 12.0f;   // :109
    }
}

// torch.remainder(x, T) for T = 100 * fp32(pi) and the |x| < 2^21 this encoder produces (means up to ~30 x 2^15),
// bit-exact without the library's iterative fmodf: q = trunc(|x| * INV_UP) with INV_UP a hair ABOVE 1/T never
// underestimates the quotient and overestimates it by at most 1; |x| - q T is a multiple of 2^-15 below 512 in magnitude,
// so the fma delivers it exactly; a negative result means q was one too large (+T, exact again).  The sign of x and the
// reference's final `+ T` for negative remainders (an fp32 add that rounds) follow.  Verified bit-for-bit against
// torch.remainder on 7 M values including near-multiples of T (tests/test_oracle_vs_golden.py keeps the recipe).
__device__ __forceinline__ float remainder_pos(float x, float T) {
    const float a = fabsf(x);
    const float q = __builtin_truncf(a * 0.0031830994f);  // fp32(1/T * (1 + 3 * 2^-24)), rounded up
    float r = __builtin_fmaf(-q, T, a);
    if (r < 0.0f) r += T;
    r = __builtin_copysignf(r, x);  // = fmodf(x, T)
    if (r != 0.0f && r < 0.0f) r += T;
    return r;
}

// sin(x + quadrant * pi/2) for |x| <= 100 pi + 2: three-term Cody-Waite reduction by pi/2 (k <= 201, so k * HI and k * MID
// are exact: both constants carry 13 significant bits) and the Cephes single-precision kernels on [-pi/4, pi/4].
// Max error 9.2e-8 absolute (1.5 ulp at 1) over the whole range -- the library sinf's class -- at a third of its
// instructions (no large-argument path, no branches).  cos(x) = quadrant 1.
__device__ __forceinline__ float enc_sin(float x, int quadrant = 0) {
    const float k = __builtin_rintf(x * 0.63661975f);
    float r = __builtin_fmaf(-k, 1.5705566f, x);
    r = __builtin_fmaf(-k, 0.00023967028f, r);
    r = __builtin_fmaf(-k, 1.5893255e-08f, r);
    const float z = r * r;
    float ps = __builtin_fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f);
    ps = __builtin_fmaf(ps, z, -1.6666654611e-1f);
    const float sn = __builtin_fmaf(ps * z, r, r);
    float pc = __builtin_fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f);
    pc = __builtin_fmaf(pc, z, 4.166664568298827e-2f);
    const float cs = __builtin_fmaf(pc * z, z, __builtin_fmaf(-0.5f, z, 1.0f));
    const int n = (int)k + quadrant;
    const float v = (n & 1) ? cs : sn;
    return (n & 2) ? -v : v;
}

// exp(v) for v <= 0: n = rint(v log2 e), two-term reduction by ln 2 (13-bit HI: n * HI exact), Cephes degree-6 kernel,
// ldexp (handles the denormal tail).  Max relative error 1.2e-7 on the normal range.
__device__ __forceinline__ float enc_exp_neg(float v) {
    if (!(v > -104.0f)) return v != v ? v : 0.0f;  // below the smallest denormal (the high octaves of a wide Gaussian)
    const float n = __builtin_rintf(v * 1.44269502f);
    float r = __builtin_fmaf(-n, 0.69311523f, v);
    r = __builtin_fmaf(-n, 3.1946183e-05f, r);
    float p = __builtin_fmaf(1.9875691500e-4f, r, 1.3981999507e-3f);
    p = __builtin_fmaf(p, r, 8.3334519073e-3f);
    p = __builtin_fmaf(p, r, 4.1665795894e-2f);
    p = __builtin_fmaf(p, r, 1.6666665459e-1f);
    p = __builtin_fmaf(p, r, 5.0000001201e-1f);
    const float e = __builtin_fmaf(p * r, r, r + 1.0f);
    return ldexpf(e, (int)n);
}

__device__ __forceinline__ float safe_sin(float x) {
    const float T = 314.159271f;  // 100 * fp32(pi), general_utils/math_utils.py:155
    float xr = (fabsf(x) < T) ? x : remainder_pos(x, T);
    return enc_sin(xr);
}

// bf16 feature rows (feat_dtype 1) keep 8 significant bits, so their sin / exp come from the hardware transcendentals
// (v_sin_f32 takes revolutions, |x| / 2 pi <= 51 here leaves 18 fraction bits; v_exp_f32 is 2^x): absolute error < 2e-6,
// 1/2000 of a bf16 ulp at 1 -- a few results per thousand land on the other side of a rounding boundary (1 bf16 ulp).
// A third of the instructions of the exact kernels above: the encoder is VALU-bound.
__device__ __forceinline__ float fast_sin(float x) { return __builtin_amdgcn_sinf(x * 0.15915494f); }
__device__ __forceinline__ float fast_exp_neg(float v) { return __builtin_amdgcn_exp2f(v * 1.44269502f); }
template <bool FAST>
__device__ __forceinline__ float safe_sin_t(float x) {
    if constexpr (!FAST) return safe_sin(x);
    const float T = 314.159271f;
    const float xr = (fabsf(x) < T) ? x : remainder_pos(x, T);
    return fast_sin(xr);
}

#ifndef ENC_SPB
#define ENC_SPB 32  // samples per 256-thread block (measured on the bf16 rows, fine / coarse pass: 16 -> 54 / 32 us, 32 -> 48 / 26, 64 -> 53 / 30)
#endif

// FIRST (the coarse pass of a ray batch that is one chunk, no first-cycle jitter): the launch ALSO packs the rays and samples the first
// cycle -- a1 + a2, models/models.py:144-162 and models/samplers.py:30-62 -- from the raw origins / directions / radii: `rays` and `t_vals`
// are then OUTPUTS (bit for bit pack_first_cycle_kernel's), written by the threads that need the values anyway, and the features come
// from the same values.  One launch less at the head of every chunk.
struct FirstCycle {
    const float *ro, *rd, *rad, *t_lin;
    float near_, far_;
    int lindisp;
};

template <int KIND, bool FIRST>   // feature rows: 0 fp32 (natural column order), 1 bf16, 2 fp16 (both in MFMA k-order)
__global__ __launch_bounds__(256) void encode_kernel(const float *__restrict__ rays_in, const float *__restrict__ t_vals_in,
                                                     void *__restrict__ feat_, int n, int S, int cylinder, FirstCycle fc,
                                                     float *__restrict__ rays_out, float *__restrict__ t_out, float *__restrict__ dirs_out) {
    // dirs_out != NULL (fp32 rows only; ddnerf_encode_rays): the 32 view-direction columns are written ONCE PER RAY into dirs_out [n,32] -- the
    // reference computes them per ray and broadcasts (models/models.py:128-133) -- and columns 96..127 of the feature rows are left untouched
    const float *__restrict__ rays = FIRST ? rays_out : rays_in;       // (FIRST: only this block's own writes are read back, see phase 1)
    const float *__restrict__ t_vals = FIRST ? t_out : t_vals_in;
    __shared__ float g_mean[ENC_SPB][4];
    __shared__ float g_cov[ENC_SPB][4];
    __shared__ __attribute__((aligned(16))) float row[ENC_SPB][100];   // the IPE columns of the block's samples (stride 100: the 16
                                                                       // samples of one column sit in 16 different banks)
    __shared__ __attribute__((aligned(16))) float dirv[ENC_SPB][32];   // columns 96..127 of the block's RAYS (a ray's samples share them)
    __shared__ int ray_of[ENC_SPB];                                    // sample -> its ray's row of dirv
    constexpr bool BF16 = KIND != 0;   // (the 16-bit rows: hardware transcendentals, k-order stores)
    const size_t M = (size_t)n * S;
    const size_t m0 = (size_t)blockIdx.x * ENC_SPB;
    const int tid = threadIdx.x;

    // phase 1: per-sample Gaussian (sample = e/3, axis = e%3)
    for (int e = tid; e < ENC_SPB * 3; e += 256) {
        int ls = e / 3, a = e % 3;
        size_t m = m0 + ls;
        if (m < M) {
            int i = (int)(m / S), j = (int)(m % S);
            float t0v, t1v, rad_v, d0, d1, d2, oa;
            if constexpr (FIRST) {
                d0 = fc.rd[3 * (size_t)i];
                d1 = fc.rd[3 * (size_t)i + 1];
                d2 = fc.rd[3 * (size_t)i + 2];
                rad_v = fc.rad[i];
                oa = fc.ro[3 * (size_t)i + a];
                t0v = first_cycle_t(fc.near_, fc.far_, fc.t_lin[j], fc.lindisp);
                t1v = first_cycle_t(fc.near_, fc.far_, fc.t_lin[j + 1], fc.lindisp);
                if (a == 0) {   // this sample's fencepost (and the ray's last one), the ray's packed row: pack_first_cycle_kernel's values
                    t_out[(size_t)i * (S + 1) + j] = t0v;
                    if (j == S - 1) t_out[(size_t)i * (S + 1) + S] = t1v;
                    if (j == 0) {
                        const float nrm = sqrtf((d0 * d0 + d1 * d1) + d2 * d2);
                        float *r = rays_out + 12 * (size_t)i;
                        r[0] = fc.ro[3 * (size_t)i]; r[1] = fc.ro[3 * (size_t)i + 1]; r[2] = fc.ro[3 * (size_t)i + 2];
                        r[3] = d0; r[4] = d1; r[5] = d2;
                        r[6] = rad_v;
                        r[7] = fc.near_; r[8] = fc.far_;
                        r[9] = d0 / nrm; r[10] = d1 / nrm; r[11] = d2 / nrm;
                    }
                }
            } else {
                const float *r = rays + 12 * (size_t)i;
                const float *t = t_vals + (size_t)i * (S + 1) + j;
                t0v = t[0];
                t1v = t[1];
                rad_v = r[6];
                d0 = r[3], d1 = r[4], d2 = r[5];
                oa = r[a];
            }
            float tm, tv, rv;
            gaussian_of_interval(t0v, t1v, rad_v, cylinder, tm, tv, rv);
            float q0 = d0 * d0, q1 = d1 * d1, q2 = d2 * d2;
            float dmag = fmaxf(1e-10f, (q0 + q1) + q2);                       // :38
            float da = a == 0 ? d0 : (a == 1 ? d1 : d2), qa = da * da;
            float nul = 1.0f - qa / dmag;                                     // :42
            if (a == 0) ray_of[ls] = i - (int)(m0 / S);
            g_mean[ls][a] = da * tm + oa;                                     // :36, :30
            g_cov[ls][a] = tv * qa + rv * nul;                                // :43-45
        }
    }
    // phase 1b: view-direction encoding: 27 values per RAY (the same for all of a ray's samples: a block of 16 consecutive samples
    // touches one ray when S is a multiple of 16, at most 16); walked from the far end of the block so that it lands on other waves
    // than phase 1
    const int ray0 = (int)(m0 / S);
    const int nrays = (int)(((m0 + ENC_SPB < M ? m0 + ENC_SPB : M) - 1) / S) - ray0 + 1;
    {
        for (int t2 = 255 - tid; t2 < nrays * 12; t2 += 256) {
            int lr = t2 / 12, k = t2 % 12;  // k: 0..11 -> (freq f = k/3, axis a = k%3), emits sin and cos
            int f = k / 3, a = k % 3;
            float v;
            if constexpr (FIRST) {   // the packed row's unit direction d / ||d|| (pack_rays_kernel's arithmetic)
                const float *dd = fc.rd + 3 * (size_t)(ray0 + lr);
                v = dd[a] / sqrtf((dd[0] * dd[0] + dd[1] * dd[1]) + dd[2] * dd[2]);
            } else {
                v = rays[12 * (size_t)(ray0 + lr) + 9 + a];
            }
            float x = v * (float)(1 << f);                                // nerf_helpers.py:163-165
            dirv[lr][3 + f * 6 + a] = BF16 ? fast_sin(x) : enc_sin(x);                       // |x| <= 8
            dirv[lr][3 + f * 6 + 3 + a] = BF16 ? fast_sin(x + 1.57079637f) : enc_sin(x, 1);  // cos
            if (f == 0) dirv[lr][a] = v;                                  // include_input
            if (k < 5) dirv[lr][27 + k] = 0.0f;                           // pad columns
        }
    }
    __syncthreads();
    // phase 2: IPE.  Thread -> (sample ls = tid % 32, q = tid / 32): the six pairs 6q .. 6q + 5 = octaves 2q and 2q + 1 of the three axes.
    // The sample's Gaussian comes out of LDS ONCE (six reads, one wait) and the six pairs are straight-line code: no per-pair index
    // arithmetic, no LDS round trip per pair (the loop this replaces spent ~45 instructions and two exposed LDS reads per pair; the
    // arithmetic per value is unchanged: same outputs).  A wave = two values of q = four consecutive octaves: the low ones, whose
    // arguments stay below 100 pi, skip safe_sin's remainder as a wave.
    static_assert(ENC_SPB == 32, "phase 2 maps 256 threads onto 32 samples x 8 octave pairs");
    {
        const int ls = tid & (ENC_SPB - 1), q = tid / ENC_SPB;
        const bool live = m0 + ls < M;
        float mean[3], cov[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            mean[a] = live ? g_mean[ls][a] : 0.0f;
            cov[a] = live ? g_cov[ls][a] : 0.0f;
        }
        const float T = 314.159271f;                                          // 100 * fp32(pi), general_utils/math_utils.py:155
        float ys[6], yc[6], damp[6];
        bool big = false;
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const float s = (float)(1 << (2 * q + k / 3));
            const int a = k % 3;
            const float y = mean[a] * s;                                      // :130
            const float yv = cov[a] * (s * s);                                // :131
            damp[k] = BF16 ? fast_exp_neg(-0.5f * yv) : enc_exp_neg(-0.5f * yv);   // :149  (yv >= 0)
            ys[k] = y;
            yc[k] = y + 1.57079637f;                                          // :143  y + 0.5*fp32(pi)
            if constexpr (KIND != 0) {
                // 16-bit rows (round 5: fp16 rows too -- that tier is no parity tier, it misses 1e-4 on trained weights whatever the remainder --,
                // so that both 16-bit tiers share ONE recipe with the encoder inside their MLP kernels): torch.remainder(x, T) as x - floor(x / T) T -- one fma, EXACT whenever the quotient is right (the true
                // remainder is representable); a quotient off by one near a multiple of T, and |x| < T with x < 0 (where the reference
                // leaves x alone), move the argument by T = 100 pi + 5.6e-6: 5.6e-6 in the sine; the reference's own rounding of its
                // `fmod + T` for negative x (<= 1.5e-5) is not reproduced.  <= 2e-5 in all, 1/200 of a bf16 ulp at 1 -- and no compare,
                // no branch: a third of the instructions of the exact recipe below.
                ys[k] = __builtin_fmaf(-__builtin_floorf(ys[k] * 0.0031830988f), T, ys[k]);
                yc[k] = __builtin_fmaf(-__builtin_floorf(yc[k] * 0.0031830988f), T, yc[k]);
            } else {
                big |= !(fabsf(ys[k]) < T) || !(fabsf(yc[k]) < T);
            }
        }
        if constexpr (KIND == 0) {
            if (__builtin_amdgcn_ballot_w64(big) != 0) {
#pragma unroll
                for (int k = 0; k < 6; ++k) {
                    ys[k] = (fabsf(ys[k]) < T) ? ys[k] : remainder_pos(ys[k], T);
                    yc[k] = (fabsf(yc[k]) < T) ? yc[k] : remainder_pos(yc[k], T);
                }
            }
        }
        if (live) {
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                row[ls][6 * q + k] = damp[k] * (BF16 ? fast_sin(ys[k]) : enc_sin(ys[k]));
                row[ls][48 + 6 * q + k] = damp[k] * (BF16 ? fast_sin(yc[k]) : enc_sin(yc[k]));
            }
        }
    }
    __syncthreads();
    // phase 3: coalesced row stores (16 rows x 128 columns)
    if (BF16) {
        // bf16 rows are stored in the MFMA k-order of the 16x16x32 MLP kernel (mlp_bf16_common.h korder32): inside every 32 columns,
        // position 8g + e holds column 16(e>>2) + 4g + (e&3).  A thread owns positions 8g .. 8g+7 of one 32-column group: columns
        // 4g .. 4g+3 and 16 + 4g .. +3, i.e. two aligned float4 reads from the LDS row and ONE 16-byte store (16 per row).
        char *feat = (char *)feat_;
        for (int e = tid; e < ENC_SPB * (DDNERF_FEAT_LD / 8); e += 256) {
            const int ls = e / (DDNERF_FEAT_LD / 8), j = e % (DDNERF_FEAT_LD / 8), grp = j >> 2, g = j & 3;
            if (m0 + ls < M) {
                const float *src = grp < 3 ? &row[ls][32 * grp] : &dirv[ray_of[ls]][0];
                const float4 a = *(const float4 *)(src + 4 * g), b = *(const float4 *)(src + 16 + 4 * g);
                uint4 v;
                v.x = (unsigned)f32_to_h16<KIND>(a.x) | ((unsigned)f32_to_h16<KIND>(a.y) << 16);
                v.y = (unsigned)f32_to_h16<KIND>(a.z) | ((unsigned)f32_to_h16<KIND>(a.w) << 16);
                v.z = (unsigned)f32_to_h16<KIND>(b.x) | ((unsigned)f32_to_h16<KIND>(b.y) << 16);
                v.w = (unsigned)f32_to_h16<KIND>(b.z) | ((unsigned)f32_to_h16<KIND>(b.w) << 16);
                *(uint4 *)(feat + ((m0 + ls) * DDNERF_FEAT_LD + 8 * j) * 2) = v;
            }
        }
    } else {
        float *feat = (float *)feat_;
        if (dirs_out) {   // the rays that START in this block (their sample 0 is one of its samples): their row of the per-ray table
            for (int e = tid; e < nrays * 8; e += 256) {
                const int lr = e >> 3, q = e & 7;
                if ((size_t)(ray0 + lr) * S >= m0) *(float4 *)(dirs_out + (size_t)(ray0 + lr) * 32 + 4 * q) = *(const float4 *)&dirv[lr][4 * q];
            }
        }
        for (int e = tid; e < ENC_SPB * DDNERF_FEAT_LD / 4; e += 256) {
            int ls = e / (DDNERF_FEAT_LD / 4), c = (e % (DDNERF_FEAT_LD / 4)) * 4;
            if (c >= 96 && dirs_out) continue;
            if (m0 + ls < M) {
                const float *src = c < 96 ? &row[ls][c] : &dirv[ray_of[ls]][c - 96];
                float4 v = *(const float4 *)src;
                *(float4 *)(feat + (m0 + ls) * DDNERF_FEAT_LD + c) = v;
            }
        }
    }
}

DDN_EXPORT int ddnerf_encode(const float *rays, const float *t_vals, void *feat, int n, int S, int ray_shape,
                             int feat_dtype, ddnerf_stream_t stream) {
    DDN_REQUIRE(rays && t_vals && feat, DDNERF_E_ARG);
    DDN_REQUIRE(n > 0 && S > 0, DDNERF_E_ARG);
    DDN_REQUIRE(ray_shape == 0 || ray_shape == 1, DDNERF_E_RANGE);
    DDN_REQUIRE(feat_dtype >= 0 && feat_dtype <= 2, DDNERF_E_RANGE);
    DDN_REQUIRE(ddn_aligned(feat, 16), DDNERF_E_ALIGN);
    size_t M = (size_t)n * S;
    dim3 grid((unsigned)((M + ENC_SPB - 1) / ENC_SPB));
    const FirstCycle none{};
    if (feat_dtype == 0)
        hipLaunchKernelGGL((encode_kernel<0, false>), grid, dim3(256), 0, (hipStream_t)stream, rays, t_vals, feat, n, S, ray_shape, none,
                           (float *)nullptr, (float *)nullptr, (float *)nullptr);
    else if (feat_dtype == 1)
        hipLaunchKernelGGL((encode_kernel<1, false>), grid, dim3(256), 0, (hipStream_t)stream, rays, t_vals, feat, n, S, ray_shape, none,
                           (float *)nullptr, (float *)nullptr, (float *)nullptr);
    else
        hipLaunchKernelGGL((encode_kernel<2, false>), grid, dim3(256), 0, (hipStream_t)stream, rays, t_vals, feat, n, S, ray_shape, none,
                           (float *)nullptr, (float *)nullptr, (float *)nullptr);
    return ddn_launch_status();
}

// a1 + a2 + (a3 + a4 + a5) in ONE launch: ddnerf_pack_rays_first_cycle without jitter (t_rand == NULL) followed by ddnerf_encode of its
// outputs -- rays [n,12] and t_vals [n,nc+1] are written, feat [n*nc,128] is the encoding of exactly those values (bit for bit the two
// entry points' outputs).  lindisp as ddnerf_pack_rays_first_cycle.
DDN_EXPORT int ddnerf_encode_first_cycle(const float *origins, const float *directions, const float *radii, float near_, float far_,
                                         const float *t_lin, int lindisp, float *rays, float *t_vals, void *feat, int n, int nc,
                                         int ray_shape, int feat_dtype, ddnerf_stream_t stream) {
    DDN_REQUIRE(origins && directions && radii && t_lin && rays && t_vals && feat, DDNERF_E_ARG);
    DDN_REQUIRE(n > 0 && nc > 0, DDNERF_E_ARG);
    DDN_REQUIRE(ray_shape == 0 || ray_shape == 1, DDNERF_E_RANGE);
    DDN_REQUIRE(lindisp >= 0 && lindisp <= 2, DDNERF_E_RANGE);
    DDN_REQUIRE(feat_dtype >= 0 && feat_dtype <= 2, DDNERF_E_RANGE);
    DDN_REQUIRE(ddn_aligned(feat, 16), DDNERF_E_ALIGN);
    size_t M = (size_t)n * nc;
    dim3 grid((unsigned)((M + ENC_SPB - 1) / ENC_SPB));
    const FirstCycle fc{origins, directions, radii, t_lin, near_, far_, lindisp};
    if (feat_dtype == 0)
        hipLaunchKernelGGL((encode_kernel<0, true>), grid, dim3(256), 0, (hipStream_t)stream, (const float *)nullptr, (const float *)nullptr, feat,
                           n, nc, ray_shape, fc, rays, t_vals, (float *)nullptr);
    else if (feat_dtype == 1)
        hipLaunchKernelGGL((encode_kernel<1, true>), grid, dim3(256), 0, (hipStream_t)stream, (const float *)nullptr, (const float *)nullptr, feat,
                           n, nc, ray_shape, fc, rays, t_vals, (float *)nullptr);
    else
        hipLaunchKernelGGL((encode_kernel<2, true>), grid, dim3(256), 0, (hipStream_t)stream, (const float *)nullptr, (const float *)nullptr, feat,
                           n, nc, ray_shape, fc, rays, t_vals, (float *)nullptr);
    return ddn_launch_status();
}

// a3 + a4 + a5 with the view-direction columns ONCE PER RAY (fp32 rows): feat [n*S,128] gets its columns 0..95 (96..127 are left untouched),
// dirs [n,32] one row per ray = what columns 96..127 of every row of that ray would hold (models/models.py:128-133: the reference encodes the
// ray's direction once and broadcasts it over the samples).  For ddnerf_mlp_f32_forward_rays / ddnerf_mlp_x3_forward_rays.
DDN_EXPORT int ddnerf_encode_rays(const float *rays, const float *t_vals, float *feat, float *dirs, int n, int S, int ray_shape,
                                  ddnerf_stream_t stream) {
    DDN_REQUIRE(rays && t_vals && feat && dirs, DDNERF_E_ARG);
    DDN_REQUIRE(n > 0 && S > 0, DDNERF_E_ARG);
    DDN_REQUIRE(ray_shape == 0 || ray_shape == 1, DDNERF_E_RANGE);
    DDN_REQUIRE(ddn_aligned(feat, 16) && ddn_aligned(dirs, 16), DDNERF_E_ALIGN);
    size_t M = (size_t)n * S;
    const FirstCycle none{};
    hipLaunchKernelGGL((encode_kernel<0, false>), dim3((unsigned)((M + ENC_SPB - 1) / ENC_SPB)), dim3(256), 0, (hipStream_t)stream, rays, t_vals,
                       (void *)feat, n, S, ray_shape, none, (float *)nullptr, (float *)nullptr, dirs);
    return ddn_launch_status();
}

// ... and with the rays packed and the first cycle sampled in the same launch (ddnerf_encode_first_cycle, fp32 rows)
DDN_EXPORT int ddnerf_encode_first_cycle_rays(const float *origins, const float *directions, const float *radii, float near_, float far_,
                                              const float *t_lin, int lindisp, float *rays, float *t_vals, float *feat, float *dirs, int n, int nc,
                                              int ray_shape, ddnerf_stream_t stream) {
    DDN_REQUIRE(origins && directions && radii && t_lin && rays && t_vals && feat && dirs, DDNERF_E_ARG);
    DDN_REQUIRE(n > 0 && nc > 0, DDNERF_E_ARG);
    DDN_REQUIRE(ray_shape == 0 || ray_shape == 1, DDNERF_E_RANGE);
    DDN_REQUIRE(lindisp >= 0 && lindisp <= 2, DDNERF_E_RANGE);
    DDN_REQUIRE(ddn_aligned(feat, 16) && ddn_aligned(dirs, 16), DDNERF_E_ALIGN);
    size_t M = (size_t)n * nc;
    const FirstCycle fc{origins, directions, radii, t_lin, near_, far_, lindisp};
    hipLaunchKernelGGL((encode_kernel<0, true>), dim3((unsigned)((M + ENC_SPB - 1) / ENC_SPB)), dim3(256), 0, (hipStream_t)stream,
                       (const float *)nullptr, (const float *)nullptr, (void *)feat, n, nc, ray_shape, fc, rays, t_vals, dirs);
    return ddn_launch_status();
}
