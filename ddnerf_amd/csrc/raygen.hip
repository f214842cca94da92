// Ray generation on device: pinhole bundles with per-pixel radii and the NDC warp.
// Compiled with -ffp-contract=off (reference operation order).
#include "common.h"

struct Cam {
    float r[9];  // cam2world[:3,:3] row-major
    float t[3];  // cam2world[:3,3]
};

// get_ray_bundle   general_utils/nerf_helpers.py:67-125.  One thread per pixel (row j, column i).
__global__ void ray_bundle_kernel(int H, int W, float focal, Cam cam, float *__restrict__ origins,
                                  float *__restrict__ directions, float *__restrict__ radii) {
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= H * W) return;
    int j = idx / W, i = idx % W;
    float d0 = ((float)i - (float)W * 0.5f) / focal;          // :103
    float d1 = -((float)j - (float)H * 0.5f) / focal;         // :104
    float d2 = -1.0f;                                         // :105
    for (int a = 0; a < 3; ++a) {
        float v = (d0 * cam.r[3 * a] + d1 * cam.r[3 * a + 1]) + d2 * cam.r[3 * a + 2];   // :109-111
        if (v == 0.0f) v += 1e-5f;                            // :115
        directions[3 * (size_t)idx + a] = v;
        float o = cam.t[a];
        if (o == 0.0f) o += 1e-5f;                            // :114
        origins[3 * (size_t)idx + a] = o;
    }
    // dx = |directions_cam[j] - directions_cam[j+1]| (only the y component differs); the last row repeats row H-3
    int jj = j < H - 1 ? j : H - 3;                           // :117-119
    if (jj < 0) jj = 0;
    float y0 = -((float)jj - (float)H * 0.5f) / focal, y1 = -((float)(jj + 1) - (float)H * 0.5f) / focal;
    float dy = y0 - y1;
    float dx = sqrtf((0.0f + dy * dy) + 0.0f);
    radii[idx] = dx * 2.0f / 3.46410155f;                     // :123  2/sqrt(12)
}

DDN_EXPORT int ddnerf_ray_bundle(int H, int W, float focal, const float *cam2world_host, float *origins,
                                 float *directions, float *radii, ddnerf_stream_t stream) {
    DDN_REQUIRE(cam2world_host && origins && directions && radii, DDNERF_E_ARG);
    DDN_REQUIRE(H > 2 && W > 0, DDNERF_E_ARG);
    Cam cam;
    for (int a = 0; a < 3; ++a) {
        for (int b = 0; b < 3; ++b) cam.r[3 * a + b] = cam2world_host[4 * a + b];
        cam.t[a] = cam2world_host[4 * a + 3];
    }
    hipLaunchKernelGGL(ray_bundle_kernel, dim3((H * W + 255) / 256), dim3(256), 0, (hipStream_t)stream, H, W, focal, cam,
                       origins, directions, radii);
    return ddn_launch_status();
}

// ndc_mipnerf_rays   data_utils/dataset_helpers.py:3-42.  Pass 1: warp origins/directions; pass 2: radii from the
// distances between neighbouring NDC origins.
__global__ void ndc_warp_kernel(int H, int W, float focal, float near_, const float *__restrict__ ro,
                                const float *__restrict__ rd, float *__restrict__ o_ndc, float *__restrict__ d_ndc) {
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= H * W) return;
    const float *o = ro + 3 * (size_t)idx, *d = rd + 3 * (size_t)idx;
    float t = -(near_ + o[2]) / d[2];                                           // :8
    float ox = o[0] + t * d[0], oy = o[1] + t * d[1], oz = o[2] + t * d[2];     // :9
    float cw = -1.0f / ((float)W / (2.0f * focal)), ch = -1.0f / ((float)H / (2.0f * focal));
    o_ndc[3 * (size_t)idx + 0] = cw * ox / oz;                                  // :12
    o_ndc[3 * (size_t)idx + 1] = ch * oy / oz;                                  // :13
    o_ndc[3 * (size_t)idx + 2] = 1.0f + 2.0f * near_ / oz;                      // :14
    d_ndc[3 * (size_t)idx + 0] = cw * (d[0] / d[2] - ox / oz);                  // :16-20
    d_ndc[3 * (size_t)idx + 1] = ch * (d[1] / d[2] - oy / oz);                  // :21-25
    d_ndc[3 * (size_t)idx + 2] = -2.0f * near_ / oz;                            // :26
}

__device__ __forceinline__ float dist3(const float *a, const float *b) {
    float x = a[0] - b[0], y = a[1] - b[1], z = a[2] - b[2];
    return sqrtf((x * x + y * y) + z * z);
}

__global__ void ndc_radii_kernel(int H, int W, const float *__restrict__ o_ndc, float *__restrict__ radii) {
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= H * W) return;
    int j = idx / W, i = idx % W;
    int jj = j < H - 1 ? j : H - 3, ii = i < W - 1 ? i : W - 3;                 // :34-38 (last row/column repeat)
    const float *base = o_ndc;
    float dx = dist3(base + 3 * ((size_t)jj * W + i), base + 3 * ((size_t)(jj + 1) * W + i));
    float dy = dist3(base + 3 * ((size_t)j * W + ii), base + 3 * ((size_t)j * W + ii + 1));
    radii[idx] = (0.5f * (dx + dy)) * 2.0f / 3.46410155f;                       // :41
}

DDN_EXPORT int ddnerf_ndc_rays(int H, int W, float focal, float near_, const float *origins, const float *directions,
                               float *origins_ndc, float *directions_ndc, float *radii, ddnerf_stream_t stream) {
    DDN_REQUIRE(origins && directions && origins_ndc && directions_ndc && radii, DDNERF_E_ARG);
    DDN_REQUIRE(H > 2 && W > 2, DDNERF_E_ARG);
    dim3 grid((H * W + 255) / 256);
    hipLaunchKernelGGL(ndc_warp_kernel, grid, dim3(256), 0, (hipStream_t)stream, H, W, focal, near_, origins, directions,
                       origins_ndc, directions_ndc);
    hipLaunchKernelGGL(ndc_radii_kernel, grid, dim3(256), 0, (hipStream_t)stream, H, W, origins_ndc, radii);
    return ddn_launch_status();
}

// switch_t_ndc_to_regular   data_utils/dataset_helpers.py:45-49 (train_model.py:227-228: the depth maps of an NDC validation
// pass back in camera-space units).  ro / rd are the REGULAR (un-warped) bundle of the same view; only their z columns are read.
__global__ void ndc_depth_to_regular_kernel(long n, const float *__restrict__ ndc_depth, const float *__restrict__ ro,
                                            const float *__restrict__ rd, float *__restrict__ out) {
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    const float t = ndc_depth[idx], oz = ro[3 * idx + 2], dz = rd[3 * idx + 2];
    out[idx] = (t * oz) / (dz - t * dz) + 1.0f;
}

DDN_EXPORT int ddnerf_ndc_depth_to_regular(long n, const float *ndc_depth, const float *origins, const float *directions, float *depth,
                                           ddnerf_stream_t stream) {
    DDN_REQUIRE(ndc_depth && origins && directions && depth, DDNERF_E_ARG);
    DDN_REQUIRE(n > 0, DDNERF_E_ARG);
    hipLaunchKernelGGL(ndc_depth_to_regular_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, n, ndc_depth,
                       origins, directions, depth);
    return ddn_launch_status();
}
