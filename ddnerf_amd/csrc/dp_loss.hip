// K5: depth-distribution KL loss  estimate_dp_loss  models/dd_utils.py:6-78 (forward).
// Compiled with -ffp-contract=off.  One 64-lane wave per ray.
//
// Stages (all on the caller's stream, no host sync):
//   1. keep[i]   = blender ? torch.sum(w1[i]) > 1e-10 : 1                      (:16)   wave per row
//   2. rank[i]   = exclusive prefix count of keep, R = number of kept rows      (:22-28) every block of step 3 counts the
//                  flags in front of its own rows itself (integer sums), the finish kernel counts them all: no scan launch
//   3. rowsum[r] = sum_m kl(m) of kept row i (r = rank[i]); bug-for-bug: left_tails_0 is read at
//                  row r, not row i, because the reference does not filter it (:22-28, :57)
//   4. loss      = sum_r rowsum[r] / (R*nf)  (kl_div reduction='mean'); 0 if R == 0 (:19-20)
#include "common.h"

#define DPL_WAVES 4

// (sum of v[0 .. first), sum of v[0 .. n)) by the whole 256-thread block, in every thread: what a block needs of the
// exclusive scan of the keep flags (the rank of its first row, the number of kept rows) without a scan kernel in between.
// Integer sums: any order gives the same result.  Cost: every block reads all n flags, n^2 / 4 reads in total -- 4 M at the
// benchmark's 4096 rays, 67 M L2 reads (~25 us) at the shipped configs' 16384-ray chunks (< 1 % of such a chunk); a chunk size far
// beyond that (65536 rays: 1e9 reads) should get the single-block scan launch back instead.
__device__ __forceinline__ int2 dpl_block_counts(const int *__restrict__ v, int first, int n) {
    __shared__ int part[2][4];
    int before = 0, all = 0;
    for (int i = threadIdx.x; i < n; i += 256) {
        const int x = v[i];
        all += x;
        before += i < first ? x : 0;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        before += __shfl_xor(before, o);
        all += __shfl_xor(all, o);
    }
    if ((threadIdx.x & 63) == 0) {
        part[0][threadIdx.x >> 6] = before;
        part[1][threadIdx.x >> 6] = all;
    }
    __syncthreads();
    return make_int2(part[0][0] + part[0][1] + part[0][2] + part[0][3], part[1][0] + part[1][1] + part[1][2] + part[1][3]);
}
// rank of row `row` (kept rows before it) from the block's first-row rank
__device__ __forceinline__ int dpl_rank_of(const int *__restrict__ keep, int first_rank, int first, int row) {
    int r = first_rank;
    for (int i = first; i < row; ++i) r += keep[i];
    return r;
}

__global__ __launch_bounds__(256) void dpl_keep_kernel(const float *__restrict__ w1, int n, int nf, int blender,
                                                       int *__restrict__ keep) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int row = blockIdx.x * DPL_WAVES + wave;
    const bool live = row < n;
    float *buf = smem + (size_t)wave * nf;
    if (live)
        for (int j = lane; j < nf; j += 64) buf[j] = w1[(size_t)row * nf + j];
    ddn_wave_sync();
    float s = ddn_aten_sum_wave(buf, nf, lane);
    if (live && lane == 0) keep[row] = blender ? (s > 1e-10f ? 1 : 0) : 1;
}

__global__ __launch_bounds__(256) void dpl_rows_kernel(
    const float *__restrict__ t1, const float *__restrict__ t0, const float *__restrict__ w1,
    const float *__restrict__ w0, const float *__restrict__ mus0, const float *__restrict__ sig0,
    const float *__restrict__ left0, const float *__restrict__ part0, int n, int nc, int nf,
    const int *__restrict__ keep, float *__restrict__ rowsum) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int row = blockIdx.x * DPL_WAVES + wave;
    const int2 cnt = dpl_block_counts(keep, blockIdx.x * DPL_WAVES, n);
    const bool live = row < n && keep[row < n ? row : 0];
    const size_t i = live ? row : 0;
    const int r = live ? dpl_rank_of(keep, cnt.x, blockIdx.x * DPL_WAVES, row) : 0;
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
    ddn_wave_sync();
    float s0 = ddn_aten_sum_wave(p0, nc, lane);                       // :31
    float s1 = ddn_aten_sum_wave(p1, nf, lane);                       // :32
    if (live) {
        for (int j = lane; j < nc; j += 64) p0[j] = p0[j] / s0;
        for (int j = lane; j < nf; j += 64) p1[j] = p1[j] / s1;
    }
    ddn_wave_sync();
    if (live && lane == 0) ddn_chain_cdf(p0, cdf, nc);  // torch.cumsum order (double running sum), clamp at 1      :38-41
    ddn_wave_sync();
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
    ddn_wave_sync();
    // estimated_pdf_1 = clamp(diff, 0) + eps, renormalised                                      :68-72
    if (live)
        for (int m = lane; m < nf; m += 64) {
            float dlt = est[m + 1] - est[m];
            if (dlt < 0.0f) dlt = 0.0f;                                // :70
            e1[m] = dlt + eps;
        }
    ddn_wave_sync();
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

// mean over the kept rows in a fixed order (thread t owns rows t, t+1024, ...; then an LDS tree), and -- when the caller
// passes the DD head's scalars -- the level-1 `dp_loss` record of the reference, dp * nf + mus_reg + sig_reg
// (models/models.py:287-289), in the same launch.
__global__ __launch_bounds__(1024) void dpl_finish_kernel(const float *__restrict__ rowsum, const int *__restrict__ keep, int n, int nf,
                                                         float *__restrict__ loss, const float *__restrict__ reg_scal,
                                                         float *__restrict__ loss_total) {
    __shared__ double part[1024];
    __shared__ int cnt[1024];
    const int tid = threadIdx.x;
    int c = 0;
    for (int i = tid; i < n; i += 1024) c += keep[i];
    cnt[tid] = c;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
        if (tid < o) cnt[tid] += cnt[tid + o];
        __syncthreads();
    }
    const int R = cnt[0];  // kept rows
    double s = 0.0;
    for (int r = tid; r < R; r += 1024) s += (double)rowsum[r];
    part[tid] = s;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
        if (tid < o) part[tid] += part[tid + o];
        __syncthreads();
    }
    if (tid == 0) {
        const float l = R > 0 ? (float)(part[0] / ((double)R * nf)) : 0.0f;
        *loss = l;
        if (loss_total) *loss_total = (l * (float)nf + reg_scal[2]) + reg_scal[3];
    }
}

DDN_EXPORT size_t ddnerf_dp_loss_workspace_bytes(int n) { return n > 0 ? (size_t)n * 12 + 16 : 0; }

DDN_EXPORT int ddnerf_dp_loss_forward(const float *t1, const float *t0, const float *w1, const float *w0,
                                      const float *mus0, const float *sig0, const float *left0, const float *part0,
                                      int n, int nc, int nf, int blender, float *loss, const float *reg_scal,
                                      float *loss_total, void *workspace, ddnerf_stream_t stream) {
    DDN_REQUIRE(t1 && t0 && w1 && w0 && mus0 && sig0 && left0 && part0 && loss && workspace, DDNERF_E_ARG);
    DDN_REQUIRE((reg_scal == nullptr) == (loss_total == nullptr), DDNERF_E_ARG);
    DDN_REQUIRE(n > 0 && nc > 0 && nf > 0, DDNERF_E_ARG);
    int *keep = (int *)workspace;
    float *rowsum = (float *)(keep + 2 * n);  // (workspace layout of ddnerf_dp_loss_workspace_bytes: n flags, n spare, n row sums)
    hipStream_t st = (hipStream_t)stream;
    dim3 grid((n + DPL_WAVES - 1) / DPL_WAVES);
    hipLaunchKernelGGL(dpl_keep_kernel, grid, dim3(256), (size_t)DPL_WAVES * nf * sizeof(float), st, w1, n, nf, blender,
                       keep);
    size_t lds = (size_t)DPL_WAVES * (2 * nc + 3 * nf + 2) * sizeof(float);
    DDN_REQUIRE(lds <= 64 * 1024, DDNERF_E_RANGE);
    hipLaunchKernelGGL(dpl_rows_kernel, grid, dim3(256), lds, st, t1, t0, w1, w0, mus0, sig0, left0, part0, n, nc, nf,
                       keep, rowsum);
    hipLaunchKernelGGL(dpl_finish_kernel, dim3(1), dim3(1024), 0, st, rowsum, keep, n, nf, loss, reg_scal, loss_total);
    return ddn_launch_status();
}

// The same with the row filter already in the workspace (ddnerf_composite_forward_keep wrote keep[0..n) while it composited the
// fine pass: torch.sum(w1) is that kernel's wsum): two launches instead of three.
DDN_EXPORT int ddnerf_dp_loss_forward_kept(const float *t1, const float *t0, const float *w1, const float *w0,
                                           const float *mus0, const float *sig0, const float *left0, const float *part0,
                                           int n, int nc, int nf, float *loss, const float *reg_scal, float *loss_total,
                                           void *workspace, ddnerf_stream_t stream) {
    DDN_REQUIRE(t1 && t0 && w1 && w0 && mus0 && sig0 && left0 && part0 && loss && workspace, DDNERF_E_ARG);
    DDN_REQUIRE((reg_scal == nullptr) == (loss_total == nullptr), DDNERF_E_ARG);
    DDN_REQUIRE(n > 0 && nc > 0 && nf > 0, DDNERF_E_ARG);
    int *keep = (int *)workspace;
    float *rowsum = (float *)(keep + 2 * n);
    hipStream_t st = (hipStream_t)stream;
    dim3 grid((n + DPL_WAVES - 1) / DPL_WAVES);
    size_t lds = (size_t)DPL_WAVES * (2 * nc + 3 * nf + 2) * sizeof(float);
    DDN_REQUIRE(lds <= 64 * 1024, DDNERF_E_RANGE);
    hipLaunchKernelGGL(dpl_rows_kernel, grid, dim3(256), lds, st, t1, t0, w1, w0, mus0, sig0, left0, part0, n, nc, nf,
                       keep, rowsum);
    hipLaunchKernelGGL(dpl_finish_kernel, dim3(1), dim3(1024), 0, st, rowsum, keep, n, nf, loss, reg_scal, loss_total);
    return ddn_launch_status();
}

// ---------------------------------------------------------------------------------------------------
// K5b  backward of estimate_dp_loss w.r.t. (w0, mus0, sig0) -- the only differentiable inputs at the call site
// (models/models.py:287-288: every other argument is .detach()ed).  The discrete choices (row filter, bin
// indices k / ki, the two clamps) are constants of the backward pass, exactly as in autograd.
// Recomputes the forward quantities of the row, then walks the chain backwards.  Scatter-adds into bins are
// done bin-parallel (lane j sums the samples m whose bin is j, in ascending m: k_m is monotone because t1 is
// sorted) so the result is order-fixed -- no atomics.
// Rows that were filtered out receive zero gradient.
// ---------------------------------------------------------------------------------------------------
#define DPLB_WORDS(nc, nf) (4 * (nc) + 2 + 11 * ((nf) + 1))

__global__ __launch_bounds__(256) void dpl_rows_bwd_kernel(
    const float *__restrict__ t1, const float *__restrict__ t0, const float *__restrict__ w1,
    const float *__restrict__ w0, const float *__restrict__ mus0, const float *__restrict__ sig0,
    const float *__restrict__ left0, const float *__restrict__ part0, int n, int nc, int nf,
    const int *__restrict__ keep,
    const float *__restrict__ g_loss, float *__restrict__ g_w0, float *__restrict__ g_mus, float *__restrict__ g_sig) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int row = blockIdx.x * DPL_WAVES + wave;
    const bool inrange = row < n;
    const int2 cnt = dpl_block_counts(keep, blockIdx.x * DPL_WAVES, n);  // (rank of the block's first row, kept rows)
    const bool live = inrange && keep[inrange ? row : 0];
    const size_t i = inrange ? row : 0;
    const int r = live ? dpl_rank_of(keep, cnt.x, blockIdx.x * DPL_WAVES, row) : 0;
    const int nf1 = nf + 1;
    float *p0 = smem + (size_t)wave * DPLB_WORDS(nc, nf);  // [nc]
    float *cdf = p0 + nc;                                  // [nc+1]
    float *gp0 = cdf + nc + 1;                             // [nc]    d/d p0
    float *gcdf = gp0 + nc;                                // [nc+1]  pass flag of min(1,.), then d/d cdf
    float *p1 = gcdf + nc + 1;                             // [nf+1]
    float *est = p1 + nf1;                                 // [nf+1]  e_m after the >1 clamp
    float *ge = est + nf1;                                 // [nf+1]  d/d e_m
    float *cA = ge + nf1;                                  // [nf+1]  d e_m / d p0[ki]
    float *cMu = cA + nf1;                                 // [nf+1]  d e_m / d mu[ki]
    float *cSg = cMu + nf1;                                // [nf+1]  d e_m / d sg[ki]
    float *dq = cSg + nf1;                                 // [nf+1]  d_m + eps, then d L / d d_m
    int *kk = (int *)(dq + nf1);                           // [nf+1]  k_m
    int *kki = kk + nf1;                                   // [nf+1]  ki_m
    int *eclamp = kki + nf1;                               // [nf+1]  e_m was clamped to 1
    int *dpos = eclamp + nf1;                              // [nf+1]  d_m was not clamped to 0
    const float eps = 1e-12f;
    const float *T1 = t1 + i * nf1, *T0 = t0 + i * (nc + 1);
    const float *lt = left0 + (size_t)r * nc;  // misaligned on purpose, like the forward
    const float *pt = part0 + i * nc, *mu = mus0 + i * nc, *sg = sig0 + i * nc;

    if (inrange && !live)  // filtered row: zero gradient
        for (int j = lane; j < nc; j += 64) {
            g_w0[i * nc + j] = 0.0f;
            g_mus[i * nc + j] = 0.0f;
            g_sig[i * nc + j] = 0.0f;
        }
    if (live) {
        for (int j = lane; j < nc; j += 64) p0[j] = w0[i * nc + j] + eps;
        for (int j = lane; j < nf; j += 64) p1[j] = w1[i * nf + j] + eps;
    }
    ddn_wave_sync();
    const float s0 = ddn_aten_sum_wave(p0, nc, lane);
    const float s1 = ddn_aten_sum_wave(p1, nf, lane);
    if (live) {
        for (int j = lane; j < nc; j += 64) p0[j] = p0[j] / s0;
        for (int j = lane; j < nf; j += 64) p1[j] = p1[j] / s1;
    }
    ddn_wave_sync();
    if (live && lane == 0) {
        double a = 0.0;
        cdf[0] = 0.0f;
        gcdf[0] = 0.0f;
        for (int j = 0; j < nc - 1; ++j) {
            a += (double)p0[j];
            float c = (float)a;
            cdf[j + 1] = fminf(1.0f, c);
            gcdf[j + 1] = c < 1.0f ? 1.0f : 0.0f;
        }
        cdf[nc] = 1.0f;
        gcdf[nc] = 0.0f;
    }
    ddn_wave_sync();
    if (live)
        for (int m = lane; m <= nf; m += 64) {
            const float tm = T1[m];
            int lo = 0, hi = nc + 1;
            while (lo < hi) {
                int mid = (lo + hi) >> 1;
                if (tm > T0[mid]) lo = mid + 1;
                else hi = mid;
            }
            int k = lo > 0 ? lo - 1 : 0, ki = k;
            while (ki > 0 && cdf[ki - 1] == cdf[k]) --ki;
            if (ki > nc - 1) ki = nc - 1;
            float seclen = T0[ki + 1] - T0[ki];
            float mr = T0[ki] + mu[ki] * seclen, sr = sg[ki] * seclen;
            float x = (tm - mr) / sr;
            float a = (ddn_norm_cdf(x) - lt[ki]) / pt[ki];
            float e = cdf[k] + a * p0[ki];
            int cl = e > 1.0f;
            est[m] = cl ? 1.0f : e;
            eclamp[m] = cl;
            kk[m] = k;
            kki[m] = ki;
            float dedx = (p0[ki] / pt[ki]) * (0.3989422804f * expf(-0.5f * x * x));  // d e / d x
            cA[m] = a;
            cMu[m] = dedx * (-1.0f / sr) * seclen;   // x = (t - mr)/sr, mr = T0 + mu*seclen
            cSg[m] = dedx * (-x / sr) * seclen;      // sr = sg*seclen
        }
    ddn_wave_sync();
    if (live)
        for (int m = lane; m < nf; m += 64) {
            float dlt = est[m + 1] - est[m];
            int pos = !(dlt < 0.0f);
            dq[m] = (pos ? dlt : 0.0f) + eps;
            dpos[m] = pos;
        }
    ddn_wave_sync();
    const float se = ddn_aten_sum_wave(dq, nf, lane);
    float p1sum = 0.0f;
    if (live)
        for (int m = lane; m < nf; m += 64) p1sum += p1[m];
    for (int o = 32; o > 0; o >>= 1) p1sum += __shfl_xor(p1sum, o);
    const float gscale = live ? g_loss[0] / ((float)cnt.y * (float)nf) : 0.0f;
    ddn_wave_sync();
    if (live)
        for (int m = lane; m < nf; m += 64) {  // l = sum_m xlogy(p,p) - p log q,  q = (d+eps)/se
            float q = dq[m] / se;
            dq[m] = dpos[m] ? gscale * (p1sum - p1[m] / q) / se : 0.0f;
        }
    ddn_wave_sync();
    if (live)
        for (int m = lane; m <= nf; m += 64) {
            float g = 0.0f;
            if (m > 0) g += dq[m - 1];
            if (m < nf) g -= dq[m];
            ge[m] = eclamp[m] ? 0.0f : g;
        }
    ddn_wave_sync();
    if (live)
        for (int j = lane; j <= nc; j += 64) {  // d/d cdf[j] = sum over the (contiguous) m with k_m == j
            int lo = 0, hi = nf1;
            while (lo < hi) {
                int mid = (lo + hi) >> 1;
                if (kk[mid] < j) lo = mid + 1;
                else hi = mid;
            }
            float acc = 0.0f;
            for (int m = lo; m <= nf && kk[m] == j; ++m) acc += ge[m];
            gcdf[j] = acc * gcdf[j];  // zero for cdf[0], cdf[nc] and the clamped prefixes
        }
    ddn_wave_sync();
    if (live && lane == 0) {  // cdf[k] = sum_{j<k} p0[j]  ->  d/d p0[j] += sum_{k>j} gcdf[k]
        float run = 0.0f;
        for (int j = nc - 1; j >= 0; --j) {
            run += gcdf[j + 1];
            gp0[j] = run;
        }
    }
    ddn_wave_sync();
    if (live)
        for (int j = lane; j < nc; j += 64) {  // per-bin sums over the (contiguous) m with ki_m == j
            int lo = 0, hi = nf1;
            while (lo < hi) {
                int mid = (lo + hi) >> 1;
                if (kki[mid] < j) lo = mid + 1;
                else hi = mid;
            }
            float a = 0.0f, b = 0.0f, c = 0.0f;
            for (int m = lo; m <= nf && kki[m] == j; ++m) {
                a += ge[m] * cA[m];
                b += ge[m] * cMu[m];
                c += ge[m] * cSg[m];
            }
            gp0[j] += a;
            g_mus[i * nc + j] = b;
            g_sig[i * nc + j] = c;
        }
    ddn_wave_sync();
    float dot = 0.0f;  // p0 = (w0+eps)/s0  ->  g_w0_j = (gp0_j - sum_i gp0_i p0_i) / s0
    if (live)
        for (int j = lane; j < nc; j += 64) dot += gp0[j] * p0[j];
    for (int o = 32; o > 0; o >>= 1) dot += __shfl_xor(dot, o);
    if (live)
        for (int j = lane; j < nc; j += 64) g_w0[i * nc + j] = (gp0[j] - dot) / s0;
}

DDN_EXPORT int ddnerf_dp_loss_backward(const float *t1, const float *t0, const float *w1, const float *w0,
                                       const float *mus0, const float *sig0, const float *left0, const float *part0,
                                       int n, int nc, int nf, int blender, const float *g_loss, float *g_w0,
                                       float *g_mus, float *g_sig, void *workspace, ddnerf_stream_t stream) {
    DDN_REQUIRE(t1 && t0 && w1 && w0 && mus0 && sig0 && left0 && part0 && g_loss && g_w0 && g_mus && g_sig && workspace,
                DDNERF_E_ARG);
    DDN_REQUIRE(n > 0 && nc > 0 && nf > 0, DDNERF_E_ARG);
    int *keep = (int *)workspace;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid((n + DPL_WAVES - 1) / DPL_WAVES);
    // the keep flags are recomputed (cheap) so the backward does not depend on the forward's workspace
    hipLaunchKernelGGL(dpl_keep_kernel, grid, dim3(256), (size_t)DPL_WAVES * nf * sizeof(float), st, w1, n, nf, blender,
                       keep);
    size_t lds = (size_t)DPL_WAVES * DPLB_WORDS(nc, nf) * 4;
    DDN_REQUIRE(lds <= 64 * 1024, DDNERF_E_RANGE);
    hipLaunchKernelGGL(dpl_rows_bwd_kernel, grid, dim3(256), lds, st, t1, t0, w1, w0, mus0, sig0, left0, part0, n, nc, nf,
                       keep, g_loss, g_w0, g_mus, g_sig);
    return ddn_launch_status();
}
