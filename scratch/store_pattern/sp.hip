// store-pattern bandwidth: the record stores of the x3 training kernels without any compute
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define ROWS 2560
#define BLOCK_BYTES (ROWS * 64)
template <int MODE>  // 0: dword nt, blocked layout (4 runs of 64 B per instruction); 1: dwordx4 nt quad-row addresses; 2: dword plain
__global__ __launch_bounds__(256, 1) void k(char *rec, long ntiles) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 15, g = lane >> 4;
    for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        char *base = rec + (size_t)(tile * 8 + wave * 2) * BLOCK_BYTES;
        for (int b = 0; b < 152; ++b)
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                char *row = base + (size_t)c * BLOCK_BYTES + (size_t)b * 16 * 64;
                if (MODE == 1) {
                    __builtin_nontemporal_store(u32x4{(unsigned)b, 1u, 2u, 3u}, (u32x4 *)(row + 256 * g + 16 * j));
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        unsigned *p = (unsigned *)(row + e * 64 + 256 * g + 4 * j);
                        if (MODE == 0) __builtin_nontemporal_store((unsigned)(b + e), p);
                        else *p = (unsigned)(b + e);
                    }
                }
            }
    }
}
int main() {
    const long M = 524288, ntiles = M / 128;
    char *rec;
    hipMalloc(&rec, (size_t)ROWS * M * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 3; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            for (int i = 0; i < 5; ++i) {
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(256), 0, 0, rec, ntiles);
                if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(256), 0, 0, rec, ntiles);
                if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(256), 0, 0, rec, ntiles);
            }
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
            const double bytes = (double)ntiles * 4 * 152 * 2 * 4 * 256;
            if (rep) printf("mode %d: %.3f ms  %.2f TB/s (%.2f GB)\n", mode, ms, bytes / ms / 1e9, bytes / 1e9);
        }
    }
    return 0;
}
