// FETCH_SIZE calibration: the same 268 MB [M,128] fp32 buffer read once with (a) the MLP kernels' per-lane-row
// pattern (lane = sample, 16 B pieces of its own 512-B row) and (b) a plain coalesced float4 stream.
#include <hip/hip_runtime.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
extern "C" __global__ __launch_bounds__(256) void calib_rows(const float *__restrict__ feat, float *__restrict__ out, long M) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 31, h = lane >> 5;
    const long m = (long)blockIdx.x * 128 + wave * 32 + j;
    if (m >= M) return;
    const float *frow = feat + (size_t)m * 128;
    float s = 0.f;
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 v = *(const f32x4 *)(frow + 32 * b + 8 * g + 4 * h);
            s += v.x + v.y + v.z + v.w;
        }
    out[(size_t)m * 2 + h] = s;
}
extern "C" __global__ __launch_bounds__(256) void calib_stream(const float *__restrict__ feat, float *__restrict__ out, long n4) {
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    float s = 0.f;
    for (; i < n4; i += (long)gridDim.x * 256) {
        const f32x4 v = *(const f32x4 *)(feat + 4 * i);
        s += v.x + v.y + v.z + v.w;
    }
    out[(long)blockIdx.x * 256 + threadIdx.x] = s;
}
// (c) the persistent bf16 MLP kernel's pattern: bf16 [M,128] rows (256 B); lane (j = l & 15, g = l >> 4) reads 16 B at byte
// 64 q + 16 g of row 16 c + j, for q = 0..3 and column blocks c = 0..3 of the wave's 64 samples: every instruction fetches 64-byte
// quarter rows
extern "C" __global__ __launch_bounds__(256) void calib_bf16rows(const unsigned short *__restrict__ feat, float *__restrict__ out, long M) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 15, g = lane >> 4;
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const long m = (long)blockIdx.x * 256 + wave * 64 + c * 16 + j;
        if (m >= M) continue;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 v = *(const f32x4 *)((const char *)feat + (size_t)m * 256 + 64 * q + 16 * g);
            s += v.x + v.y + v.z + v.w;
        }
    }
    out[(size_t)blockIdx.x * 256 + tid] = s;
}
extern "C" int run_bf16rows(const void *feat, float *out, long M, void *st) {
    hipLaunchKernelGGL(calib_bf16rows, dim3((M + 255) / 256), dim3(256), 0, (hipStream_t)st, (const unsigned short *)feat, out, M);
    return (int)hipGetLastError();
}
extern "C" int run_rows(const float *feat, float *out, long M, void *st) {
    hipLaunchKernelGGL(calib_rows, dim3((M + 127) / 128), dim3(256), 0, (hipStream_t)st, feat, out, M);
    return (int)hipGetLastError();
}
extern "C" int run_stream(const float *feat, float *out, long n4, void *st) {
    hipLaunchKernelGGL(calib_stream, dim3(4096), dim3(256), 0, (hipStream_t)st, feat, out, n4);
    return (int)hipGetLastError();
}
