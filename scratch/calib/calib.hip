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
extern "C" int run_rows(const float *feat, float *out, long M, void *st) {
    hipLaunchKernelGGL(calib_rows, dim3((M + 127) / 128), dim3(256), 0, (hipStream_t)st, feat, out, M);
    return (int)hipGetLastError();
}
extern "C" int run_stream(const float *feat, float *out, long n4, void *st) {
    hipLaunchKernelGGL(calib_stream, dim3(4096), dim3(256), 0, (hipStream_t)st, feat, out, n4);
    return (int)hipGetLastError();
}
