// Order-fixed reduction of the split-K partial slabs of the weight-gradient kernels (mlp_f32_wgrad.hip, mlp_x3_wgrad.hip).
#pragma once
#include "common.h"

// dst[r*dst_ld + dst_col0 + c] = sum over slabs (ascending) of slab[(src_row0 + r)*slab_ld + src_col0 + c]
static __global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float *__restrict__ slabs, int nslabs, size_t slab_stride,
                                                           int slab_ld, int src_row0, int src_col0, int n_rows, int n_cols,
                                                           float *__restrict__ dst, int dst_ld, int dst_col0) {
    // 64 output elements per block; the slabs of an element are split over 4 thread groups (k mod 4), each keeping 4
    // loads in flight; partial sums are combined in a fixed order (reproducible, no atomics)
    __shared__ float part[4][64];
    const int e = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int idx = blockIdx.x * 64 + e;
    const bool live = idx < n_rows * n_cols;
    const int r = live ? idx / n_cols : 0, c = live ? idx % n_cols : 0;
    const float *p = slabs + (size_t)(src_row0 + r) * slab_ld + src_col0 + c;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int k = g;
    for (; k + 12 < nslabs; k += 16) {
        s0 += p[(size_t)k * slab_stride];
        s1 += p[(size_t)(k + 4) * slab_stride];
        s2 += p[(size_t)(k + 8) * slab_stride];
        s3 += p[(size_t)(k + 12) * slab_stride];
    }
    for (; k < nslabs; k += 4) s0 += p[(size_t)k * slab_stride];
    part[g][e] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (g == 0 && live) dst[(size_t)r * dst_ld + dst_col0 + c] = (part[0][e] + part[1][e]) + (part[2][e] + part[3][e]);
}


// The weight slabs and the bias slabs of one job in ONE launch: blocks [0, nb_w) reduce the weights, the rest the bias sums
// (same arithmetic and order as two wgrad_reduce_kernel launches).
struct WgradReduceJob {
    const float *slabs;
    size_t slab_stride;
    int slab_ld, n_rows, n_cols, dst_ld, dst_col0;
    float *dst;
};
static __global__ __launch_bounds__(256) void wgrad_reduce_pair_kernel(WgradReduceJob w, WgradReduceJob bs, int nb_w, int nslabs) {
    __shared__ float part[4][64];
    const bool second = (int)blockIdx.x >= nb_w;
    const WgradReduceJob &j = second ? bs : w;
    const int e = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int idx = ((int)blockIdx.x - (second ? nb_w : 0)) * 64 + e;
    const bool live = idx < j.n_rows * j.n_cols;
    const int r = live ? idx / j.n_cols : 0, c = live ? idx % j.n_cols : 0;
    const float *p = j.slabs + (size_t)r * j.slab_ld + c;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int k = g;
    for (; k + 12 < nslabs; k += 16) {
        s0 += p[(size_t)k * j.slab_stride];
        s1 += p[(size_t)(k + 4) * j.slab_stride];
        s2 += p[(size_t)(k + 8) * j.slab_stride];
        s3 += p[(size_t)(k + 12) * j.slab_stride];
    }
    for (; k < nslabs; k += 4) s0 += p[(size_t)k * j.slab_stride];
    part[g][e] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (g == 0 && live) j.dst[(size_t)r * j.dst_ld + j.dst_col0 + c] = (part[0][e] + part[1][e]) + (part[2][e] + part[3][e]);
}
