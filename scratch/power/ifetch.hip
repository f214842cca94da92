// Does a straight-line kernel pay for crossing 4-KiB pages of CODE, and does that depend on whether the code fits the instruction
// cache (64 KB per two CUs)?  A dense mix -- per MFMA two 8-byte VALU instructions: 24 bytes of code per 16 cycles, 1.5 bytes per cycle
// and wave (the MLP kernels: ~1.1) -- as ONE loop body of BODY_KB kilobytes, run for the same number of MFMAs whatever the size.
// hipcc --offload-arch=gfx950 -O3 ifetch.hip -o ifetch
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// MIX 4: MIX 1 + s_barrier every 128 MFMAs.  MIX 5: MIX 4 + a clock stamp (s_memtime, s_waitcnt, 8-byte global store) every 32 MFMAs.
// MIX 0: MFMA + 2 VALU.  MIX 1: + one ds_read_b128 per four MFMAs, consumed three reads later behind a counted s_waitcnt (the MLP
// kernels' A-fragment ring).  MIX 2: + one 16-byte global load per 16 MFMAs from a 1.4 MB buffer (data-side translations and L2 traffic
// beside the instruction fetch).  MIX 3: both.
template <int BODY_KB, int MIX>
__global__ __launch_bounds__(256, 1) void k(const bf16x8 *__restrict__ src, float *__restrict__ out, int total_mfma, unsigned long long *stamps) {
    constexpr int PER = BODY_KB * 1024 / 24;  // MFMAs per loop body (the extra instructions of MIX > 0 make the body a little longer)
    __shared__ bf16x8 lds[64 * 16];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 64 * 16; i += 256) lds[i] = src[i & 127];
    __syncthreads();
    bf16x8 ring[4] = {lds[lane], lds[64 + lane], lds[128 + lane], lds[192 + lane]};
    f32x4 gsum = {};
    const f32x4 *gsrc = (const f32x4 *)src + lane + 64 * (tid >> 6);
    bf16x8 a = src[lane], b = src[64 + lane];
    f32x4 acc[8] = {};
    unsigned d[8] = {1u * tid, 3u * tid, 5u * tid, 7u * tid, 9u * tid, 11u * tid, 13u * tid, 15u * tid};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < total_mfma / PER; ++it) {
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            if constexpr ((MIX & 1) != 0 || MIX >= 4) {
                asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[i & 7]) : "v"(ring[(i >> 2) & 3]), "v"(b));
                if ((i & 3) == 0) ring[((i >> 2) + 3) & 3] = lds[((i >> 2) & 15) * 64 + lane];
            } else {
                asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[i & 7]) : "v"(a), "v"(b));
            }
            asm volatile("v_pk_max_i16 %0, %0, 0\n\tv_pk_max_i16 %1, %1, 0" : "+v"(d[i & 7]), "+v"(d[(i + 3) & 7]));
            if constexpr (MIX >= 4) {
                if ((i & 127) == 127) __builtin_amdgcn_s_barrier();
                if (MIX == 5 && (i & 31) == 31) {
                    const unsigned long long t = __builtin_amdgcn_s_memtime();
                    if (tid == 0) stamps[1024 + (i >> 5) % 64] = t;
                }
            }
            if constexpr (MIX < 4 && (MIX & 2) != 0)
                if ((i & 15) == 0) gsum += __builtin_nontemporal_load(gsrc + ((i >> 4) * 2053 % 5000) * 16);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    asm volatile("s_nop 7\n\ts_nop 7");
    f32x4 s = {};
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i];
    unsigned dd = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) dd += d[i];
    out[(size_t)blockIdx.x * 256 + tid] = s[0] + s[1] + s[2] + s[3] + (float)dd + gsum[0] + gsum[3];
    if (tid == 0) stamps[blockIdx.x] = t1 - t0;
}

template <int BODY_KB, int MIX>
static void run(const bf16x8 *src, float *out, unsigned long long *stamps, int ncu) {
    constexpr int PER = BODY_KB * 1024 / 24;
    const int total = 2000000 / PER * PER;   // ~2 M MFMAs per wave: 32 M cycles
    for (int rep = 0; rep < 4; ++rep) {
        hipLaunchKernelGGL((k<BODY_KB, MIX>), dim3(ncu), dim3(256), 0, 0, src, out, total, stamps);
        hipDeviceSynchronize();
    }
    std::vector<unsigned long long> h(ncu);
    hipMemcpy(h.data(), stamps, ncu * 8, hipMemcpyDeviceToHost);
    double cyc = 0;
    for (int i = 0; i < ncu; ++i) cyc += (double)h[i];
    cyc /= ncu;
    printf("mix %d  loop body %4d KB (%3d pages): %.2f cycles per MFMA (ideal 16), matrix pipe %.1f %% busy; extra cycles per 4 KiB of code %.0f\n", MIX, BODY_KB, BODY_KB / 4,
           cyc / total, 100.0 * 16.0 * total / cyc, (cyc / total - 16.0) * (4096.0 / 24.0));
}

int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int ncu = p.multiProcessorCount;
    std::vector<unsigned short> h(1500000 / 2, 0x3f80);
    bf16x8 *src;
    float *out;
    unsigned long long *stamps;
    hipMalloc(&src, h.size() * 2);
    hipMalloc(&out, (size_t)ncu * 256 * 4);
    hipMalloc(&stamps, (ncu + 2048) * 8);
    hipMemcpy(src, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    run<24, 1>(src, out, stamps, ncu);
    run<192, 1>(src, out, stamps, ncu);
    run<24, 4>(src, out, stamps, ncu);
    run<192, 4>(src, out, stamps, ncu);
    run<24, 5>(src, out, stamps, ncu);
    run<192, 5>(src, out, stamps, ncu);
    return 0;
}
