// The loss of one training iteration as ONE launch and its gradient as one:  train_model.py:156-172
//     loss = c0 * mse(rgb_coarse, target) + c1 * mse(rgb_fine, target) + c_dp * mean(dp_loss)
// (torch.nn.functional.mse_loss = the mean of the squared differences; `dp_loss` holds one value per ray chunk).  The reference -- and the
// plain torch code this replaces in train_step.py -- assembles it from ~12 elementwise / reduction launches and autograd derives ~14 more;
// each is a few microseconds of kernel and a launch gap on the step's one stream: ~0.2 ms of a 7.3-ms x3 step (tools/train_small_ops.py).
// One 1024-thread workgroup, sums in double in a fixed order (thread t owns elements t, t + 1024, ...; then an LDS tree): reproducible.
#include "common.h"

__global__ __launch_bounds__(1024) void train_loss_fwd_kernel(const float *__restrict__ rgb0, const float *__restrict__ rgb1,
                                                              const float *__restrict__ target, long count, const float *__restrict__ dp,
                                                              int n_dp, float c0, float c1, float c_dp, float *__restrict__ out) {
    __shared__ double part[3][1024];
    const int tid = threadIdx.x;
    double s0 = 0.0, s1 = 0.0, sd = 0.0;
    for (long i = tid; i < count; i += 1024) {
        const float t = target[i];
        const float d0 = rgb0[i] - t;
        s0 += (double)(d0 * d0);
        if (rgb1) {
            const float d1 = rgb1[i] - t;
            s1 += (double)(d1 * d1);
        }
    }
    for (int i = tid; i < n_dp; i += 1024) sd += (double)dp[i];
    part[0][tid] = s0;
    part[1][tid] = s1;
    part[2][tid] = sd;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
        if (tid < o) {
            part[0][tid] += part[0][tid + o];
            part[1][tid] += part[1][tid + o];
            part[2][tid] += part[2][tid + o];
        }
        __syncthreads();
    }
    if (tid == 0) {
        const float m0 = (float)(part[0][0] / (double)count), m1 = rgb1 ? (float)(part[1][0] / (double)count) : 0.0f;
        const float md = n_dp > 0 ? (float)(part[2][0] / (double)n_dp) : 0.0f;
        float loss = c0 * m0;                      // train_model.py:159-161: sum over the levels, in order
        if (rgb1) loss = loss + c1 * m1;
        if (n_dp > 0) loss = loss + c_dp * md;     // :163-167
        out[0] = loss;
        out[1] = m0;
        out[2] = m1;
        out[3] = md;
    }
}

// g_rgb_l = g * c_l * 2 (rgb_l - target) / count;  g_dp[j] = g * c_dp / n_dp        (g = the upstream gradient of the scalar loss)
__global__ void train_loss_bwd_kernel(const float *__restrict__ rgb0, const float *__restrict__ rgb1, const float *__restrict__ target,
                                      long count, int n_dp, float c0, float c1, float c_dp, const float *__restrict__ g,
                                      float *__restrict__ g_rgb0, float *__restrict__ g_rgb1, float *__restrict__ g_dp) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const float gg = g ? g[0] : 1.0f;
    if (i < count) {
        const float t = target[i];
        const float k = 2.0f / (float)count;
        g_rgb0[i] = (gg * c0) * (k * (rgb0[i] - t));
        if (g_rgb1) g_rgb1[i] = (gg * c1) * (k * (rgb1[i] - t));
    }
    if (g_dp && i < n_dp) g_dp[i] = (gg * c_dp) / (float)n_dp;
}

DDN_EXPORT int ddnerf_train_loss_forward(const float *rgb0, const float *rgb1, const float *target, long count, const float *dp, int n_dp,
                                         float c0, float c1, float c_dp, float *out, ddnerf_stream_t stream) {
    DDN_REQUIRE(rgb0 && target && out, DDNERF_E_ARG);
    DDN_REQUIRE(count > 0 && n_dp >= 0 && (n_dp == 0 || dp), DDNERF_E_ARG);
    hipLaunchKernelGGL(train_loss_fwd_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, rgb0, rgb1, target, count, dp, n_dp, c0, c1, c_dp, out);
    return ddn_launch_status();
}

DDN_EXPORT int ddnerf_train_loss_backward(const float *rgb0, const float *rgb1, const float *target, long count, int n_dp, float c0, float c1,
                                          float c_dp, const float *g, float *g_rgb0, float *g_rgb1, float *g_dp, ddnerf_stream_t stream) {
    DDN_REQUIRE(rgb0 && target && g_rgb0, DDNERF_E_ARG);
    DDN_REQUIRE((rgb1 == nullptr) == (g_rgb1 == nullptr), DDNERF_E_ARG);
    DDN_REQUIRE(count > 0 && n_dp >= 0 && (n_dp == 0 || g_dp), DDNERF_E_ARG);
    const long total = count > n_dp ? count : n_dp;
    hipLaunchKernelGGL(train_loss_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, rgb0, rgb1, target, count,
                       n_dp, c0, c1, c_dp, g, g_rgb0, g_rgb1, n_dp ? g_dp : nullptr);
    return ddn_launch_status();
}
