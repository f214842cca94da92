// DIAGNOSTIC library only (libddnerf_diag.so): bare bf16 MFMA loops at the bf16 MLP kernels' per-wave tile, one 256-thread workgroup
// per CU, for the "same-box ceiling" that bench.py prints beside the bf16 kernel's roofline fraction.  The chip is power-limited on
// these loops: what a loop sustains is set by the clock the part holds, and parts differ by several per cent, so
// a ceiling is only comparable with a kernel time when both were measured on the same device in the same run.
//   mode 0: v_mfma_f32_16x16x32_bf16 back to back, operands in registers
//   mode 1: + the A fragment of every k-step (four MFMAs) read from LDS by one ds_read_b128, three k-steps ahead (the kernels' ratio)
//   mode 2: + the kernels' weight staging: one 1-KiB LDS-DMA piece per wave per 24 MFMAs from a 1.4 MB image (L2 hits) into four
//           36-KiB slots, one s_barrier behind a counted s_waitcnt vmcnt (three barriers of lead) per 96 MFMAs (two-group kernel: one piece per 24.8, one
//           barrier per 117)
//   mode 3 / 4: modes 1 / 2 with HALF the fragment reads (one ds_read_b128 per EIGHT MFMAs): what a body whose two sample groups walk a
//           weight chunk in lockstep -- every A fragment feeding both groups' MFMAs -- would be up against
// Operands: uniform(-1, 1) bf16 (the caller fills `src`).  Every workgroup reports d(s_memtime), d(s_memrealtime).
#include "mlp_bf16_common.h"

typedef __bf16 cbf16x8 __attribute__((ext_vector_type(8)));
typedef float cf32x4 __attribute__((ext_vector_type(4)));

#define CEIL_SLOT (36 * 1024)
#define CEIL_IMG (1408 * 1024)   // bytes of `src` a mode-2 loop streams (a multiple of 4 KiB; the MLP's packed image is 1.36 MiB)

template <int MODE, bool HALF = false>
__global__ __launch_bounds__(256, 1) void mfma_ceiling_kernel(const char *__restrict__ src, float *__restrict__ out, int iters,
                                                              unsigned long long *__restrict__ stamps) {
    __shared__ __attribute__((aligned(16))) char lds[4 * CEIL_SLOT];
    const int tid = threadIdx.x, lane = tid & 63;
    const unsigned wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid * 16; i < 4 * CEIL_SLOT; i += 256 * 16) *(uint4 *)(lds + i) = *(const uint4 *)(src + (i % CEIL_IMG));
    __syncthreads();
    cbf16x8 b[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) b[i] = *(const cbf16x8 *)(src + 65536 + 1024 * i + 16 * lane + 8192 * wave);
    cbf16x8 a[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) a[i] = *(const cbf16x8 *)(lds + 1024 * i + 16 * lane);
    cf32x4 acc[8] = {};
    const unsigned lds0 = __builtin_amdgcn_readfirstlane(lds_addr_of(lds));
    const unsigned lane16 = 16u * lane;
    unsigned src_k = 4 * wave, dst_k = 4 * wave;   // this wave's next 4-KiB run of the image / of LDS (the four waves interleave, as the kernels' waves share a chunk)
    const char *p_it = src;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 24; ++s) {   // 24 k-steps x 4 MFMAs = 96 MFMAs per iteration
            const cbf16x8 av = HALF ? a[(s >> 1) & 3] : a[s & 3];
            // (a scalar base per iteration + an immediate per k-step: no address arithmetic beside the MFMAs)
            if constexpr (MODE >= 1 && !HALF) a[(s + 3) & 3] = *(const cbf16x8 *)(lds + (it & 3) * (24 * 1024) + s * 1024 + 16 * lane);
            if constexpr (MODE >= 1 && HALF) {
                if ((s & 1) == 0) a[((s >> 1) + 3) & 3] = *(const cbf16x8 *)(lds + (it & 3) * (24 * 1024) + s * 1024 + 16 * lane);
            }
            if constexpr (MODE == 2) {
                if (s % 6 == 0) {   // one piece per 24 MFMAs; the four pieces of an iteration share ONE (M0, source base) pair and differ in
                                    // the immediate offset, which applies to both addresses -- the kernels' form (a piece = one instruction)
                    if (s == 0) {
                        const unsigned dst = lds0 + dst_k * 1024u;
                        p_it = src + (size_t)src_k * 1024u;
                        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" : : "s"(dst) : "memory");
                        src_k += 16;
                        if (src_k >= CEIL_IMG / 1024u) src_k -= CEIL_IMG / 1024u;
                        dst_k += 16;
                        if (dst_k >= 144u) dst_k -= 144u;
                    }
                    asm volatile("global_load_lds_dwordx4 %0, %1 offset:%2" : : "v"(lane16), "s"(p_it), "n"((s / 6) * 1024) : "memory");
                }
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[4 * (s & 1) + c]) : "v"(av), "v"(b[(c + s) & 7]));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (MODE == 2) {
            // (the pieces issued three iterations ago have landed, for every wave: the kernels give a chunk two to three periods between
            // its issue and the barrier that certifies it -- a piece takes 2.5 - 3 k cycles to land when every CU streams)
            asm volatile("s_waitcnt vmcnt(12)\n\ts_barrier" ::: "memory");
        }
    }
    asm volatile("s_waitcnt vmcnt(0)\n\ts_nop 7\n\ts_nop 7" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    cf32x4 s = {};
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i];
    out[(size_t)blockIdx.x * 256 + tid] = s[0] + s[1] + s[2] + s[3];
    if (tid == 0) {
        stamps[2 * blockIdx.x] = t1 - t0;
        stamps[2 * blockIdx.x + 1] = r1 - r0;
    }
}

// src: >= CEIL_IMG + 128 KiB bytes of bf16 operands; out: 256 floats per CU; stamps: 2 values per CU.  Launches one workgroup per CU;
// a wave issues iters x 96 MFMAs (16 x 16 x 32 x 2 FLOP each, four waves per workgroup).
DDN_EXPORT int ddnerf_debug_mfma_ceiling(int mode, const void *src, float *out, int iters, unsigned long long *stamps, ddnerf_stream_t stream) {
    DDN_REQUIRE(src && out && stamps && iters > 0, DDNERF_E_ARG);
    DDN_REQUIRE(mode >= 0 && mode <= 4, DDNERF_E_RANGE);
    const dim3 grid((unsigned)ddn_cu_count());
    if (mode == 0) hipLaunchKernelGGL(mfma_ceiling_kernel<0>, grid, dim3(256), 0, (hipStream_t)stream, (const char *)src, out, iters, stamps);
    if (mode == 1) hipLaunchKernelGGL(mfma_ceiling_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, (const char *)src, out, iters, stamps);
    if (mode == 2) hipLaunchKernelGGL(mfma_ceiling_kernel<2>, grid, dim3(256), 0, (hipStream_t)stream, (const char *)src, out, iters, stamps);
    if (mode == 3) hipLaunchKernelGGL((mfma_ceiling_kernel<1, true>), grid, dim3(256), 0, (hipStream_t)stream, (const char *)src, out, iters, stamps);
    if (mode == 4) hipLaunchKernelGGL((mfma_ceiling_kernel<2, true>), grid, dim3(256), 0, (hipStream_t)stream, (const char *)src, out, iters, stamps);
    return ddn_launch_status();
}
DDN_EXPORT size_t ddnerf_debug_mfma_ceiling_src_bytes(void) { return (size_t)CEIL_IMG + 128 * 1024; }
