// K5: depth-distribution KL loss  estimate_dp_loss  models/dd_utils.py:6-78 (forward).
// Compiled with -ffp-contract=off.  One 64-lane wave per ray.
//
// Stages (all on the caller's stream, no host sync):
//   1. keep[i]   = blender ? torch.sum(w1[i]) > 1e-10 : 1                      (:16)   wave per row
//   2. rank[i]   = exclusive prefix count of keep (fixed order), R = total     (:22-28) one block
//   3. rowsum[r] = sum_m kl(m) of kept row i (r = rank[i]); bug-for-bug: left_tails_0 is read at
//                  row r, not row i, because the reference does not filter it (:22-28, :57)
//   4. loss      = sum_r rowsum[r] / (R*nf)  (kl_div reduction='mean'); 0 if R == 0 (:19-20)
#include "common.h"

#define DPL_WAVES 4

__global__ __launch_bounds__(256) void dpl_keep_kernel(const float *__restrict__ w1, int n, int nf, int blender,
                                                       int *__restrict__ keep) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int row = blockIdx.x * DPL_WAVES + wave;
    const bool live = row < n;
    float *buf = smem + (size_t)wave * nf;
    if (live)
        for (int j = lane; j < nf; j += 64) buf[j] = w1[(size_t)row * nf + j];
    __syncthreads();
    float s = ddn_aten_sum_wave(buf, nf, lane);
    if (live && lane == 0) keep[row] = blender ? (s > 1e-10f ? 1 : 0) : 1;
}

// exclusive scan of keep[0..n) by ONE 1024-thread block (n <= a few 10^4 rows per chunk)
__global__ __launch_bounds__(1024) void dpl_scan_kernel(const int *__restrict__ keep, int n, int *__restrict__ rank,
                                                        int *__restrict__ total) {
    __shared__ int part[1024];
    const int tid = threadIdx.x;
    const int per = (n + 1023) / 1024;
    const int lo = tid * per, hi = min(n, lo + per);
    int s = 0;
    for (int i = lo; i < hi; ++i) s += keep[i];
    part[tid] = s;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {  // Hillis-Steele inclusive scan
        int v = tid >= off ? part[tid - off] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    int base = tid > 0 ? part[tid - 1] : 0;
    for (int i = lo; i < hi; ++i) {
        rank[i] = base;
        base += keep[i];
    }
    if (tid == 1023) *total = part[1023];
}

__global__ __launch_bounds__(256) void dpl_rows_kernel(
    const float *__restrict__ t1, const float *__restrict__ t0, const float *__restrict__ w1,
    const float *__restrict__ w0, const float *__restrict__ mus0, const float *__restrict__ sig0,
    const float *__restrict__ left0, const float *__restrict__ part0, int n, int nc, int nf,
    const int *__restrict__ keep, const int *__restrict__ rank, float *__restrict__ rowsum) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int row = blockIdx.x * DPL_WAVES + wave;
    const bool live = row < n && keep[row < n ? row : 0];
    const size_t i = live ? row : 0;
    const int r = live ? rank[i] : 0;
    float *p0 = smem + (size_t)wave * (2 * nc + 3 * nf + 2);  // [nc]
    float *cdf = p0 + nc;                                         // [nc+1]
    float *p1 = cdf + nc + 1;                                     // [nf]
    float *est = p1 + nf;                                         // [nf+1]
    float *e1 = est + nf + 1;                                     // [nf]
    const float eps = 1e-12f;
    const float *T1 = t1 + i * (nf + 1), *T0 = t0 + i * (nc + 1);
    const float *lt = left0 + (size_t)r * nc;  // misaligned on purpose (see header)
    const float *pt = part0 + i * nc, *mu = mus0 + i * nc, *sg = sig0 + i * nc;

    if (live) {
        for (int j = lane; j < nc; j += 64) p0[j] = w0[i * nc + j] + eps;
        for (int j = lane; j < nf; j += 64) p1[j] = w1[i * nf + j] + eps;
    }
    __syncthreads();
    float s0 = ddn_aten_sum_wave(p0, nc, lane);                       // :31
    float s1 = ddn_aten_sum_wave(p1, nf, lane);                       // :32
    if (live) {
        for (int j = lane; j < nc; j += 64) p0[j] = p0[j] / s0;
        for (int j = lane; j < nf; j += 64) p1[j] = p1[j] / s1;
    }
    __syncthreads();
    if (live && lane == 0) {  // torch.cumsum order (double running sum), clamp at 1      :38-41
        double a = 0.0;
        cdf[0] = 0.0f;
        for (int j = 0; j < nc - 1; ++j) {
            a += (double)p0[j];
            cdf[j + 1] = fminf(1.0f, (float)a);
        }
        cdf[nc] = 1.0f;
    }
    __syncthreads();
    if (live) {
        for (int m = lane; m <= nf; m += 64) {
            const float tm = T1[m];
            int lo = 0, hi = nc + 1;  // k = last j with T0[j] < tm  (mask = t1 > t0, strict)     :43
            while (lo < hi) {
                int mid = (lo + hi) >> 1;
                if (tm > T0[mid]) lo = mid + 1;
                else hi = mid;
            }
            int k = lo > 0 ? lo - 1 : 0, ki = k;
            while (ki > 0 && cdf[ki - 1] == cdf[k]) --ki;              // torch.max: first index of the max   :48
            if (ki > nc - 1) ki = nc - 1;
            float seclen = T0[ki + 1] - T0[ki];
            float mr = T0[ki] + mu[ki] * seclen;                       // :35
            float sr = sg[ki] * seclen;                                // :36
            float x = (tm - mr) / sr;                                  // :60
            float e = cdf[k] + ((ddn_norm_cdf(x) - lt[ki]) / pt[ki]) * p0[ki];   // :62-64
            if (e > 1.0f) e = 1.0f;                                    // :66
            est[m] = e;
        }
    }
    __syncthreads();
    // estimated_pdf_1 = clamp(diff, 0) + eps, renormalised                                      :68-72
    if (live)
        for (int m = lane; m < nf; m += 64) {
            float dlt = est[m + 1] - est[m];
            if (dlt < 0.0f) dlt = 0.0f;                                // :70
            e1[m] = dlt + eps;
        }
    __syncthreads();
    float se = ddn_aten_sum_wave(e1, nf, lane);                       // :72
    float acc = 0.0f;
    if (live)
        for (int m = lane; m < nf; m += 64) {
            float q = e1[m] / se, p = p1[m];
            float xlogy = (p == 0.0f) ? 0.0f : p * logf(p);           // kl_div(log q, p) = xlogy(p,p) - p*log q   :76
            acc += xlogy - p * logf(q);
        }
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o);
    if (live && lane == 0) rowsum[r] = acc;
}

__global__ void dpl_finish_kernel(const float *__restrict__ rowsum, const int *__restrict__ total, int nf,
                                  float *__restrict__ loss) {
    const int R = *total;
    double s = 0.0;
    for (int r = threadIdx.x; r < R; r += 64) s += (double)rowsum[r];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o);
    if (threadIdx.x == 0) *loss = R > 0 ? (float)(s / ((double)R * nf)) : 0.0f;
}

DDN_EXPORT size_t ddnerf_dp_loss_workspace_bytes(int n) { return n > 0 ? (size_t)n * 12 + 16 : 0; }

DDN_EXPORT int ddnerf_dp_loss_forward(const float *t1, const float *t0, const float *w1, const float *w0,
                                      const float *mus0, const float *sig0, const float *left0, const float *part0,
                                      int n, int nc, int nf, int blender, float *loss, void *workspace,
                                      ddnerf_stream_t stream) {
    DDN_REQUIRE(t1 && t0 && w1 && w0 && mus0 && sig0 && left0 && part0 && loss && workspace, DDNERF_E_ARG);
    DDN_REQUIRE(n > 0 && nc > 0 && nf > 0, DDNERF_E_ARG);
    int *keep = (int *)workspace, *rank = keep + n;
    float *rowsum = (float *)(rank + n);
    int *total = (int *)(rowsum + n);
    hipStream_t st = (hipStream_t)stream;
    dim3 grid((n + DPL_WAVES - 1) / DPL_WAVES);
    hipLaunchKernelGGL(dpl_keep_kernel, grid, dim3(256), (size_t)DPL_WAVES * nf * sizeof(float), st, w1, n, nf, blender,
                       keep);
    hipLaunchKernelGGL(dpl_scan_kernel, dim3(1), dim3(1024), 0, st, keep, n, rank, total);
    size_t lds = (size_t)DPL_WAVES * (2 * nc + 3 * nf + 2) * sizeof(float);
    DDN_REQUIRE(lds <= 64 * 1024, DDNERF_E_RANGE);
    hipLaunchKernelGGL(dpl_rows_kernel, grid, dim3(256), lds, st, t1, t0, w1, w0, mus0, sig0, left0, part0, n, nc, nf,
                       keep, rank, rowsum);
    hipLaunchKernelGGL(dpl_finish_kernel, dim3(1), dim3(64), 0, st, rowsum, total, nf, loss);
    return ddn_launch_status();
}
