/*
 * ddnerf_oracle_mlp.c -- CPU restatement of the 8x256 MLP backbones.
 *
 * TEST INFRASTRUCTURE ONLY (see ddnerf_oracle.c).  Restates
 *   MipNeRFModel.forward        models/base_architectures.py:40-61   (depth_head = 0, 4 outputs)
 *   DepthMipNeRFModel.forward   models/base_architectures.py:103-126 (depth_head = 1, 6 outputs)
 * as plain fp32 loops (axpy form over a transposed weight copy so the compiler can vectorise
 * without reassociating; FMA allowed here -- the reference's MKL GEMM order is not pinned either).
 *
 * params: 2*L pointers in reference registration order
 *   layers_xyz.0..7, fc_feat, fc_alpha, layers_dir.0, fc_rgb [, fc_mu_sigma]  -> (weight[out][in], bias[out])
 * x: [M, ldx] rows = [ipe 0:96 | dir 96:123 | pad];  out: [M, 4|6] = (rgb3, alpha[, mu, sigma])
 */
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define DDO_API __attribute__((visibility("default")))
#define RB 16 /* rows per block */
#define CLONES __attribute__((target_clones("avx512f", "avx2", "default")))

/* y[RB][nout] = b + x[RB][nin(ldx)] * Wt[nin][nout] */
CLONES static void dense_block(const float *x, int ldx, int nin, const float *Wt, const float *b, int nout, float *y,
                               int ldy, int rows, int relu) {
    int r, k, n;
    for (r = 0; r < rows; ++r) {
        float *yr = y + (size_t)r * ldy;
        const float *xr = x + (size_t)r * ldx;
        for (n = 0; n < nout; ++n) yr[n] = b[n];
        for (k = 0; k < nin; ++k) {
            const float xv = xr[k];
            const float *w = Wt + (size_t)k * nout;
            for (n = 0; n < nout; ++n) yr[n] += xv * w[n];
        }
        if (relu)
            for (n = 0; n < nout; ++n) yr[n] = yr[n] > 0.0f ? yr[n] : 0.0f;
    }
}

static float *transpose(const float *W, int nout, int nin) {
    float *t = (float *)malloc(sizeof(float) * (size_t)nout * nin);
    int o, i;
    for (o = 0; o < nout; ++o)
        for (i = 0; i < nin; ++i) t[(size_t)i * nout + o] = W[(size_t)o * nin + i];
    return t;
}

DDO_API void ddo_mlp_forward(const float *x, int ldx, const float *const *params, float *out, long M, int depth_head) {
    const int nl = depth_head ? 13 : 12;
    static const int nout_[13] = {256, 256, 256, 256, 256, 256, 256, 256, 256, 1, 128, 3, 2};
    static const int nin_[13] = {96, 256, 256, 256, 256, 352, 256, 256, 256, 256, 283, 128, 128};
    float *Wt[13];
    const int no = depth_head ? 6 : 4;
    long m0;
    int l;
    for (l = 0; l < nl; ++l) Wt[l] = transpose(params[2 * l], nout_[l], nin_[l]);
#pragma omp parallel
    {
        float *h0 = (float *)malloc(sizeof(float) * RB * 352 * 4), *h1 = h0 + RB * 352, *ft = h1 + RB * 352,
              *dr = ft + RB * 352;
#pragma omp for schedule(static)
        for (m0 = 0; m0 < M; m0 += RB) {
            int rows = (int)(M - m0 < RB ? M - m0 : RB), r, i;
            const float *xb = x + (size_t)m0 * ldx;
            float *a = h0, *b = h1, *t;
            dense_block(xb, ldx, 96, Wt[0], params[1], 256, a, 352, rows, 1);
            for (i = 1; i < 8; ++i) {
                if (i == 5) { /* cat(xyz, x) :46 */
                    for (r = 0; r < rows; ++r) {
                        memmove(a + (size_t)r * 352 + 96, a + (size_t)r * 352, sizeof(float) * 256);
                        memcpy(a + (size_t)r * 352, xb + (size_t)r * ldx, sizeof(float) * 96);
                    }
                    dense_block(a, 352, 352, Wt[5], params[11], 256, b, 352, rows, 1);
                } else {
                    dense_block(a, 352, 256, Wt[i], params[2 * i + 1], 256, b, 352, rows, 1);
                }
                t = a; a = b; b = t;
            }
            dense_block(a, 352, 256, Wt[8], params[17], 256, ft, 352, rows, 0);  /* fc_feat :50 */
            for (r = 0; r < rows; ++r) {                                          /* fc_alpha :51 */
                float alpha;
                dense_block(ft + (size_t)r * 352, 352, 256, Wt[9], params[19], 1, &alpha, 1, 1, 0);
                out[(size_t)(m0 + r) * no + 3] = alpha;
                memcpy(ft + (size_t)r * 352 + 256, xb + (size_t)r * ldx + 96, sizeof(float) * 27); /* cat(feat, dirs) :53 */
            }
            dense_block(ft, 352, 283, Wt[10], params[21], 128, dr, 352, rows, 1);
            for (r = 0; r < rows; ++r) {
                dense_block(dr + (size_t)r * 352, 352, 128, Wt[11], params[23], 3, out + (size_t)(m0 + r) * no, 3, 1, 0);
                if (depth_head)
                    dense_block(dr + (size_t)r * 352, 352, 128, Wt[12], params[25], 2, out + (size_t)(m0 + r) * no + 4, 2,
                                1, 0);
            }
        }
        free(h0);
    }
    for (l = 0; l < nl; ++l) free(Wt[l]);
}
