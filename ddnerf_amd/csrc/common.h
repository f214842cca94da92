// Shared host/device helpers of libddnerf_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ddnerf_hip.h"

#define DDN_EXPORT extern "C" __attribute__((visibility("default")))

static inline int ddn_launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? DDNERF_OK : (int)e;
}

#define DDN_REQUIRE(cond, code) \
    do {                        \
        if (!(cond)) return (code); \
    } while (0)

// CU count of the CURRENT device (a process may drive parts with different counts): looked up per device id, safe to call from
// several host threads at once (an entry is only ever written with its one value; the attribute query is cheap, unlike
// hipGetDeviceProperties).  Never 0: the persistent kernels' launchers divide by it.
static inline int ddn_cu_count() {
    static int table[64];   // (zero-initialised; int loads / stores of one aligned word)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    const bool tabled = dev >= 0 && dev < 64;
    int n = tabled ? __atomic_load_n(&table[dev], __ATOMIC_RELAXED) : 0;
    if (n <= 0) {
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        if (tabled) __atomic_store_n(&table[dev], n, __ATOMIC_RELAXED);
    }
    return n;
}

static inline bool ddn_aligned(const void *p, size_t a) { return (reinterpret_cast<uintptr_t>(p) % a) == 0; }

// ---- fp32 special functions, written with the reference's operation order -------------------------------

// general_utils/math_utils.py:193-199   0.5*(1+erf(x/sqrt2)), sqrt2 = fp32
__device__ __forceinline__ float ddn_norm_cdf(float x) {
    const float sqrt2 = 1.41421354f;
    return 0.5f * (1.0f + erff(x / sqrt2));
}

// torch.erfinv (ATen calc_erfinv<float>): rational start value + two Newton steps, float throughout.
__device__ __forceinline__ float ddn_erfinv(float y) {
    const float a0 = 0.886226899f, a1 = -1.645349621f, a2 = 0.914624893f, a3 = -0.140543331f;
    const float b0 = -2.118377725f, b1 = 1.442710462f, b2 = -0.329097515f, b3 = 0.012229801f;
    const float c0 = -1.970840454f, c1 = -1.624906493f, c2 = 3.429567803f, c3 = 1.641345311f;
    const float d0 = 3.543889200f, d1 = 1.637067800f;
    const float two_over_sqrtpi = 1.12837917f;  // 2.0f / float(sqrt(pi))
    float ya = fabsf(y), x;
    if (ya > 1.0f) return __builtin_nanf("");
    if (ya == 1.0f) return copysignf(__builtin_inff(), y);
    if (ya <= 0.7f) {
        float z = y * y;
        float num = (((a3 * z + a2) * z + a1) * z + a0);
        float dem = ((((b3 * z + b2) * z + b1) * z + b0) * z + 1.0f);
        x = y * num / dem;
    } else {
        float z = sqrtf(-logf((1.0f - ya) / 2.0f));
        float num = ((c3 * z + c2) * z + c1) * z + c0;
        float dem = (d1 * z + d0) * z + 1.0f;
        x = copysignf(num, y) / dem;
    }
    x = x - (erff(x) - y) / (two_over_sqrtpi * expf(-x * x));
    x = x - (erff(x) - y) / (two_over_sqrtpi * expf(-x * x));
    return x;
}

// general_utils/math_utils.py:202-208
__device__ __forceinline__ float ddn_norm_icdf(float x) {
    const float sqrt2 = 1.41421354f;
    return sqrt2 * ddn_erfinv(2.0f * x - 1.0f);
}

__device__ __forceinline__ float ddn_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }

// torch.nn.functional.softplus(beta=1, threshold=20)
__device__ __forceinline__ float ddn_softplus(float x) { return x > 20.0f ? x : log1pf(expf(x)); }

// torch.sum(x, dim=-1) on a contiguous fp32 row, ATen CPU order (8-lane vectors x 4 accumulators; rows
// shorter than 8 use four scalar accumulators).  Executed by ONE lane over an LDS/global row.
__device__ inline float ddn_aten_sum(const float *x, int n) {
    if (n < 8) {
        float a[4] = {0.f, 0.f, 0.f, 0.f};
        int q = n / 4;
        for (int i = 0; i < q; ++i)
            for (int k = 0; k < 4; ++k) a[k] = a[k] + x[4 * i + k];
        for (int i = 4 * q; i < n; ++i) a[0] = a[0] + x[i];
        return ((a[0] + a[1]) + a[2]) + a[3];
    }
    float P[4][8];
    int vs = n / 8, q = vs / 4;
    for (int k = 0; k < 4; ++k)
        for (int l = 0; l < 8; ++l) P[k][l] = 0.0f;
    for (int i = 0; i < q; ++i)
        for (int k = 0; k < 4; ++k)
            for (int l = 0; l < 8; ++l) P[k][l] = P[k][l] + x[(4 * i + k) * 8 + l];
    for (int i = 4 * q; i < vs; ++i)
        for (int l = 0; l < 8; ++l) P[0][l] = P[0][l] + x[i * 8 + l];
    float acc = 0.0f;
    for (int i = 8 * vs; i < n; ++i) acc = acc + x[i];
    for (int l = 0; l < 8; ++l) acc = acc + (((P[0][l] + P[1][l]) + P[2][l]) + P[3][l]);
    return acc;
}

// Barrier for LDS data that only ONE wave touches (the wave-per-ray kernels keep a region per wave): a wave's LDS operations
// execute in order, so all it takes is that the compiler neither moves accesses across this point nor keeps values in
// registers over it -- no s_barrier, the block's other waves run on.  (__syncthreads at these places made every wave of a
// 256-thread block wait for the slowest at each of up to 40 points per ray.)
__device__ __forceinline__ void ddn_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Same order, but the 8 vector lanes are 8 GPU lanes (lane l < 8 owns vector lane l); the final
// lane-0..7 chain runs on lane 0 after a gather.  Used where a whole wave is available.
// ---- the serial chains of the wave-per-ray kernels (ONE lane runs them: torch.cumprod / torch.cumsum / a j-ascending fp32 sum, step for step) ----
// Round 5, measured (tools/comp_variants.py, the fine compositing launch: 15.8 us, of which the cumprod chain 4.6 and the rgb sum 3.6):
// * a cumprod step is three double-class instructions on the chain's lane (v_cvt_f64_f32 of the factor, the dependent v_mul_f64, v_cvt_f32_f64 of
//   the prefix) at 10 - 16 cycles each (tools/calib/dp_chain.hip, profiles/r05_dp_chain_microbench.log: a dependent v_mul_f64 / v_add_f64 step costs
//   15 cycles with one lane active, 26 with the conversion of its result): ~70 cycles a step in the kernel, x 128 steps = the 4.6 us.  Batching the
//   loads in front of the steps changed nothing (the helpers below keep that form: one place for the arithmetic); taking the conversions OFF the
//   chain (the whole wave converting to double in front of it and back behind it, the chain a run of bare v_mul_f64 on doubles in LDS) made the
//   launch SLOWER (19.2 us) and was not kept.  All 4096 rays' chains already run side by side, four waves per SIMD.
// * the rgb sum was two dependent instructions a step on three lanes; the products are now formed in place by the whole wave and the chain is the
//   add alone: 3.6 -> 2.2 us.
// The same operations on the same values in the same order: bit-identical results.
#define DDN_CHAIN_BATCH 16

// torch.cumprod, exclusive (general_utils/nerf_helpers.py:43-64): x[j] <- fp32(prod_{i<j} x[i]), running product in double.  In place, one lane.
__device__ __forceinline__ void ddn_chain_cumprod_exclusive(float *x, int n) {
    double p = 1.0;
    int j = 0;
    for (; j + DDN_CHAIN_BATCH <= n; j += DDN_CHAIN_BATCH) {
        float v[DDN_CHAIN_BATCH], o[DDN_CHAIN_BATCH];
#pragma unroll
        for (int k = 0; k < DDN_CHAIN_BATCH; ++k) v[k] = x[j + k];
#pragma unroll
        for (int k = 0; k < DDN_CHAIN_BATCH; ++k) {
            o[k] = (float)p;
            p *= (double)v[k];
        }
#pragma unroll
        for (int k = 0; k < DDN_CHAIN_BATCH; ++k) x[j + k] = o[k];
    }
    for (; j < n; ++j) {
        const float v = x[j];
        x[j] = (float)p;
        p *= (double)v;
    }
}

// torch.cumsum (double running sum, fp32 prefixes) clamped at 1, as the samplers and the dp loss build their cdf (models/samplers.py:88-91,
// models/dd_utils.py:38-41): cdf[0] = 0, cdf[j + 1] = min(1, fp32(sum_{i <= j} p[i])) for j < nc - 1, cdf[nc] = 1.  p and cdf do not overlap.  One lane.
__device__ __forceinline__ void ddn_chain_cdf(const float *__restrict__ p, float *__restrict__ cdf, int nc) {
    double a = 0.0;
    cdf[0] = 0.0f;
    int j = 0;
    for (; j + DDN_CHAIN_BATCH <= nc - 1; j += DDN_CHAIN_BATCH) {
        float v[DDN_CHAIN_BATCH], o[DDN_CHAIN_BATCH];
#pragma unroll
        for (int k = 0; k < DDN_CHAIN_BATCH; ++k) v[k] = p[j + k];
#pragma unroll
        for (int k = 0; k < DDN_CHAIN_BATCH; ++k) {
            a += (double)v[k];
            o[k] = fminf(1.0f, (float)a);
        }
#pragma unroll
        for (int k = 0; k < DDN_CHAIN_BATCH; ++k) cdf[j + 1 + k] = o[k];
    }
    for (; j < nc - 1; ++j) {
        a += (double)p[j];
        cdf[j + 1] = fminf(1.0f, (float)a);
    }
    cdf[nc] = 1.0f;
}

// sum_j c[stride * j], j ascending, fp32 (the products w_j * rgb_j of torch.sum(w[..., None] * rgb, dim = -2) were formed in place by the whole wave)
__device__ __forceinline__ float ddn_chain_sum_strided(const float *__restrict__ c, int stride, int n) {
    float s = 0.0f;
    int j = 0;
    for (; j + DDN_CHAIN_BATCH <= n; j += DDN_CHAIN_BATCH) {
        float b[DDN_CHAIN_BATCH];
#pragma unroll
        for (int k = 0; k < DDN_CHAIN_BATCH; ++k) b[k] = c[stride * (j + k)];
#pragma unroll
        for (int k = 0; k < DDN_CHAIN_BATCH; ++k) s = s + b[k];
    }
    for (; j < n; ++j) s = s + c[stride * j];
    return s;
}

__device__ inline float ddn_aten_sum_wave(const float *x, int n, int lane) {
    if (n < 8) {
        float r = 0.0f;
        if (lane == 0) r = ddn_aten_sum(x, n);
        return __shfl(r, 0);
    }
    int vs = n / 8, q = vs / 4;
    float p = 0.0f;
    if (lane < 8) {
        float P0 = 0.f, P1 = 0.f, P2 = 0.f, P3 = 0.f;
        for (int i = 0; i < q; ++i) {
            P0 = P0 + x[(4 * i + 0) * 8 + lane];
            P1 = P1 + x[(4 * i + 1) * 8 + lane];
            P2 = P2 + x[(4 * i + 2) * 8 + lane];
            P3 = P3 + x[(4 * i + 3) * 8 + lane];
        }
        for (int i = 4 * q; i < vs; ++i) P0 = P0 + x[i * 8 + lane];
        p = ((P0 + P1) + P2) + P3;
    }
    float acc = 0.0f;
    if (lane == 0)
        for (int i = 8 * vs; i < n; ++i) acc = acc + x[i];
    for (int l = 0; l < 8; ++l) {
        float pl = __shfl(p, l);
        acc = acc + pl;
    }
    return __shfl(acc, 0);
}
