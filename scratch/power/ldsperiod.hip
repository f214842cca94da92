// Is there a periodic stall in a stream of ds_read_b128 + MFMA (the MLP kernels' A-fragment ring)?  One workgroup per CU, four waves,
// per k-step 4 MFMAs + 1 ds_read_b128 (ring of 4, counted waits); wave 0 of workgroup 0 stamps s_memtime every STEP k-steps.
// hipcc --offload-arch=gfx950 -O3 ldsperiod.hip -o ldsperiod
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define NST 256

template <int STEP, int READS_PER_KSTEP>
__global__ __launch_bounds__(256, 1) void k(const bf16x8 *__restrict__ src, float *__restrict__ out, unsigned long long *stamps) {
    __shared__ bf16x8 lds[64 * 64];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 64 * 64; i += 256) lds[i] = src[i & 127];
    __syncthreads();
    bf16x8 b = src[lane];
    f32x4 acc[4] = {};
    unsigned long long st[NST];
    bf16x8 ring[4] = {lds[lane], lds[64 + lane], lds[128 + lane], lds[192 + lane]};
#pragma unroll
    for (int s = 0; s < NST; ++s) {
#pragma unroll
        for (int u = 0; u < STEP; ++u) {
            const int n = s * STEP + u;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[c]) : "v"(ring[n & 3]), "v"(b));
                if (c == 0 && READS_PER_KSTEP >= 1) ring[(n + 3) & 3] = lds[((n + 3) & 63) * 64 + lane];
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        st[s] = __builtin_amdgcn_s_memtime();
    }
    asm volatile("s_nop 7\n\ts_nop 7");
    out[blockIdx.x * 256 + tid] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
    if (blockIdx.x == 0 && tid == 0)
        for (int s = 0; s < NST; ++s) stamps[s] = st[s];
}

template <int STEP, int R>
static void run(const bf16x8 *src, float *out, unsigned long long *stamps, int ncu) {
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL((k<STEP, R>), dim3(ncu), dim3(256), 0, 0, src, out, stamps);
        hipDeviceSynchronize();
    }
    std::vector<unsigned long long> h(NST);
    hipMemcpy(h.data(), stamps, NST * 8, hipMemcpyDeviceToHost);
    printf("STEP %d k-steps per stamp (%d LDS reads, ideal %d cycles), reads per k-step %d:\n", STEP, STEP * R, STEP * 64, R);
    for (int s = 1; s < NST; ++s) printf("%llu%s", h[s] - h[s - 1], (s % 32 == 0) ? "\n" : " ");
    printf("\n");
}

int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int ncu = p.multiProcessorCount;
    std::vector<unsigned short> h(128 * 8, 0x3f80);
    bf16x8 *src;
    float *out;
    unsigned long long *stamps;
    hipMalloc(&src, h.size() * 2);
    hipMalloc(&out, (size_t)ncu * 256 * 4);
    hipMalloc(&stamps, NST * 8);
    hipMemcpy(src, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    run<4, 1>(src, out, stamps, ncu);
    run<4, 0>(src, out, stamps, ncu);
    return 0;
}
