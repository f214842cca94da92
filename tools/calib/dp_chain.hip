// Microbenchmark: cycles per step of a DEPENDENT chain of double-precision / single-precision vector instructions on gfx950, one lane active
// (exec = 1) or all 64, one wave per SIMD or four (the wave-per-ray kernels' situation).  hipcc --offload-arch=gfx950 -O3 -o tools/calib/dp_chain tools/calib/dp_chain.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int OP>
__global__ void chain(double *out, unsigned long long *cyc, double seed, int active_lanes) {
    double p = seed + threadIdx.x * 1e-9;
    const double x = 0.999999;
    float pf = (float)p;
    const float xf = 0.999999f;
    const bool on = (threadIdx.x & 63) < active_lanes;
    unsigned long long t0 = 0, t1 = 0;
    if (on) {
        t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
        for (int i = 0; i < 64; ++i) {
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                if constexpr (OP == 0) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(p) : "v"(x));
                if constexpr (OP == 1) asm volatile("v_add_f64 %0, %0, %1" : "+v"(p) : "v"(x));
                if constexpr (OP == 2) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(p) : "v"(x));
                if constexpr (OP == 3) asm volatile("v_add_f32 %0, %0, %1" : "+v"(pf) : "v"(xf));
                if constexpr (OP == 4) asm volatile("v_mul_f64 %0, %0, %2\n\tv_cvt_f32_f64 %1, %0" : "+v"(p), "=v"(pf) : "v"(x));
                if constexpr (OP == 5) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(p) : "v"(pf));   // (independent conversions: issue rate)
            }
        }
        asm volatile("s_nop 0" ::"v"(p), "v"(pf));
        t1 = __builtin_amdgcn_s_memtime();
    }
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    out[blockIdx.x * blockDim.x + threadIdx.x] = p + pf;
}

template <int OP>
static void run(const char *name, int waves_per_block, int lanes) {
    const int blocks = 256;
    double *out;
    unsigned long long *cyc;
    hipMalloc(&out, sizeof(double) * blocks * 64 * waves_per_block);
    hipMalloc(&cyc, sizeof(unsigned long long) * blocks);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(chain<OP>, dim3(blocks), dim3(64 * waves_per_block), 0, 0, out, cyc, 1.0, lanes);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks);
    hipMemcpy(h.data(), cyc, sizeof(unsigned long long) * blocks, hipMemcpyDeviceToHost);
    double s = 0;
    for (auto v : h) s += v;
    printf("%-44s %2d waves per CU, %2d lanes active: %7.1f cycles per dependent step\n", name, waves_per_block, lanes, s / blocks / 1024.0);
    hipFree(out);
    hipFree(cyc);
}

int main() {
    for (int w : {4, 16})
        for (int lanes : {1, 64}) {
            run<0>("v_mul_f64 (dependent)", w, lanes);
            run<1>("v_add_f64 (dependent)", w, lanes);
            run<2>("v_fma_f64 (dependent)", w, lanes);
            run<3>("v_add_f32 (dependent)", w, lanes);
            run<4>("v_mul_f64 + v_cvt_f32_f64 of the result", w, lanes);
            run<5>("v_cvt_f64_f32 (independent)", w, lanes);
        }
    return 0;
}
