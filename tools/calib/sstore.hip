// Do SCALAR STORES work on gfx950, and what do they cost beside a chain of fp32 MFMAs?  (round 5: 1-bit ReLU masks for the fp32 training
// tier -- a v_cmp writes a wave's 64 mask bits of one accumulator register into an SGPR pair; the only way from there to memory that
// costs no vector-ALU instruction is s_store_dword*.)
//   check<OVERWRITE>: every wave turns `iters` rows of 64 floats into 64-bit masks (v_cmp_lt_f32 s[20:21], 0, v) and stores each by
//                     s_store_dwordx2; OVERWRITE 2: the data registers are overwritten by the very next instruction (is the data read at
//                     issue?), 1: after an s_waitcnt lgkmcnt(0), 0: and an s_nop between the compare and the store.  s_dcache_wb before the end.  A second kernel reads the masks back by
//                     s_load_dwordx2 and applies them with v_cndmask_b32; the host checks both against the floats.
//   side<MODE>:       cycles per v_mfma_f32_32x32x2_f32 of a dependent chain with, behind every 8th MFMA,
//                     0 nothing  1 four v_cmp -> SGPR pairs  2 four v_cmp + two s_store_dwordx4  3 two s_store_dwordx4 only
//                     4 one s_load_dwordx8 (result not waited for)  5 four v_cndmask_b32 with SGPR-pair conditions  6 eight VALU (cmp + cndmask)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int OVERWRITE>
__global__ __launch_bounds__(256) void store_kernel(const float *__restrict__ x, unsigned long long *__restrict__ masks, int iters) {
    const int lane = threadIdx.x & 63;
    const unsigned wave = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    const float *xr = x + (size_t)wave * iters * 64 + lane;
    unsigned long long *mw = masks + (size_t)wave * iters;
    for (int it = 0; it < iters; ++it) {
        const float v = xr[(size_t)it * 64];
        const unsigned off = 8u * (unsigned)it;
        if (OVERWRITE == 2)
            asm volatile("v_cmp_lt_f32 s[20:21], 0, %0\n\ts_store_dwordx2 s[20:21], %1, %2\n\ts_mov_b64 s[20:21], -1\n\tv_cmp_gt_f32 s[20:21], 0, %0"
                         ::"v"(v), "s"(mw), "s"(off) : "s20", "s21", "memory");
        else if (OVERWRITE == 1)
            asm volatile("v_cmp_lt_f32 s[20:21], 0, %0\n\ts_store_dwordx2 s[20:21], %1, %2\n\ts_waitcnt lgkmcnt(0)\n\ts_mov_b64 s[20:21], -1"
                         ::"v"(v), "s"(mw), "s"(off) : "s20", "s21", "memory");
        else
            asm volatile("v_cmp_lt_f32 s[20:21], 0, %0\n\ts_nop 7\n\ts_store_dwordx2 s[20:21], %1, %2\n\ts_waitcnt lgkmcnt(0)\n\ts_mov_b64 s[20:21], -1"
                         ::"v"(v), "s"(mw), "s"(off) : "s20", "s21", "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_dcache_wb" ::: "memory");
}

__global__ __launch_bounds__(256) void apply_kernel(const float *__restrict__ x, const unsigned long long *__restrict__ masks, float *__restrict__ y, int iters) {
    const int lane = threadIdx.x & 63;
    const unsigned wave = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    const float *xr = x + (size_t)wave * iters * 64 + lane;
    float *yr = y + (size_t)wave * iters * 64 + lane;
    const unsigned long long *mw = masks + (size_t)wave * iters;
    for (int it = 0; it < iters; ++it) {
        float v = xr[(size_t)it * 64] + 1.0f;
        const unsigned off = 8u * (unsigned)it;
        asm volatile("s_load_dwordx2 s[20:21], %1, %2\n\ts_waitcnt lgkmcnt(0)\n\tv_cndmask_b32 %0, 0, %0, s[20:21]" : "+v"(v) : "s"(mw), "s"(off) : "s20", "s21", "memory");
        yr[(size_t)it * 64] = v;
    }
}

template <int MODE>
__global__ __launch_bounds__(256, 1) void side_kernel(float *out, unsigned long long *masks, int iters, unsigned long long *stamps) {
    float a = 1.0f + threadIdx.x * 1e-3f, b = 0.5f + threadIdx.x * 1e-4f;
    float x0 = a - 1.05f, x1 = b - 0.51f, x2 = a * b - 0.6f, x3 = a - b, y0 = 1, y1 = 2, y2 = 3, y3 = 4;
    const unsigned wave = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    unsigned long long *mw = masks + (size_t)wave * 64;   // 512 B per wave, rewritten every time
#define INIT(n) asm volatile("v_accvgpr_write_b32 a" #n ", %0" ::"v"(0.0f));
    INIT(0) INIT(1) INIT(2) INIT(3) INIT(4) INIT(5) INIT(6) INIT(7) INIT(8) INIT(9) INIT(10) INIT(11) INIT(12) INIT(13) INIT(14) INIT(15)
    asm volatile("s_mov_b64 s[20:21], -1\n\ts_mov_b64 s[22:23], 0\n\ts_mov_b64 s[24:25], -1\n\ts_mov_b64 s[26:27], 0" ::: "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 64; ++s) {
            asm volatile("v_mfma_f32_32x32x2_f32 a[0:15], %0, %1, a[0:15]" ::"v"(a), "v"(b) : "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15");
            if (s % 8 != 7) continue;
            const unsigned off = 64u * (unsigned)(s / 8);
            if (MODE == 1 || MODE == 2 || MODE == 6)
                asm volatile("v_cmp_lt_f32 s[20:21], 0, %0\n\tv_cmp_lt_f32 s[22:23], 0, %1\n\tv_cmp_lt_f32 s[24:25], 0, %2\n\tv_cmp_lt_f32 s[26:27], 0, %3"
                             ::"v"(x0), "v"(x1), "v"(x2), "v"(x3) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");
            if (MODE == 2 || MODE == 3)
                asm volatile("s_store_dwordx4 s[20:23], %0, %1\n\ts_store_dwordx4 s[24:27], %0, %1 offset:16" ::"s"(mw), "s"(off) : "memory");
            if (MODE == 4) asm volatile("s_load_dwordx8 s[28:35], %0, %1" ::"s"(mw), "s"(off) : "s28", "s29", "s30", "s31", "s32", "s33", "s34", "s35", "memory");
            if (MODE == 5 || MODE == 6)
                asm volatile("v_cndmask_b32 %0, 0, %0, s[20:21]\n\tv_cndmask_b32 %1, 0, %1, s[22:23]\n\tv_cndmask_b32 %2, 0, %2, s[24:25]\n\tv_cndmask_b32 %3, 0, %3, s[26:27]"
                             : "+v"(y0), "+v"(y1), "+v"(y2), "+v"(y3));
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_dcache_wb" ::: "memory");
    float s = y0 + y1 + y2 + y3, v;
#define RD(n) asm volatile("v_accvgpr_read_b32 %0, a" #n : "=v"(v)); s += v;
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7");
    RD(0) RD(5) RD(15)
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) stamps[blockIdx.x] = t1 - t0;
}

template <int OVERWRITE>
static int run_check(int nwg, int iters) {
    const size_t nw = (size_t)nwg * 4, n = nw * iters * 64;
    std::vector<float> hx(n);
    unsigned r = 12345u + OVERWRITE;
    for (size_t i = 0; i < n; ++i) {
        r = r * 1664525u + 1013904223u;
        hx[i] = (r >> 28) == 0 ? 0.0f : ((int)(r >> 8 & 0xffff) - 32768) * 1e-3f;   // some exact zeros (mask bit 0)
    }
    float *x, *y;
    unsigned long long *m;
    CHECK(hipMalloc(&x, n * 4));
    CHECK(hipMalloc(&y, n * 4));
    CHECK(hipMalloc(&m, nw * iters * 8));
    CHECK(hipMemcpy(x, hx.data(), n * 4, hipMemcpyHostToDevice));
    CHECK(hipMemset(m, 0x5a, nw * iters * 8));
    CHECK(hipMemset(y, 0, n * 4));
    hipLaunchKernelGGL(store_kernel<OVERWRITE>, dim3(nwg), dim3(256), 0, 0, x, m, iters);
    hipLaunchKernelGGL(apply_kernel, dim3(nwg), dim3(256), 0, 0, x, m, y, iters);
    CHECK(hipDeviceSynchronize());
    std::vector<unsigned long long> hm(nw * iters);
    std::vector<float> hy(n);
    CHECK(hipMemcpy(hm.data(), m, nw * iters * 8, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(hy.data(), y, n * 4, hipMemcpyDeviceToHost));
    size_t bad_m = 0, bad_y = 0;
    for (size_t w = 0; w < nw * iters; ++w) {
        unsigned long long e = 0;
        for (int l = 0; l < 64; ++l) e |= (unsigned long long)(hx[w * 64 + l] > 0.0f) << l;
        bad_m += hm[w] != e;
        for (int l = 0; l < 64; ++l) bad_y += hy[w * 64 + l] != (hx[w * 64 + l] > 0.0f ? hx[w * 64 + l] + 1.0f : 0.0f);
    }
    printf("check overwrite=%d: %zu masks, %zu wrong in memory, %zu wrong values after s_load + v_cndmask\n", OVERWRITE, nw * iters, bad_m, bad_y);
    hipFree(x); hipFree(y); hipFree(m);
    return 0;
}

template <int MODE>
static int run_side(const char *what) {
    const int nwg = 256, iters = 2000;
    float *out;
    unsigned long long *st, *m;
    CHECK(hipMalloc(&out, nwg * 256 * 4));
    CHECK(hipMalloc(&st, nwg * 8));
    CHECK(hipMalloc(&m, (size_t)nwg * 4 * 64 * 8));
    CHECK(hipMemset(m, 0, (size_t)nwg * 4 * 64 * 8));
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(side_kernel<MODE>, dim3(nwg), dim3(256), 0, 0, out, m, iters, st);
    CHECK(hipDeviceSynchronize());
    std::vector<unsigned long long> h(nwg);
    CHECK(hipMemcpy(h.data(), st, nwg * 8, hipMemcpyDeviceToHost));
    double s = 0;
    for (auto v : h) s += (double)v;
    // s_memtime ticks at 100 MHz on this part; report per-MFMA time relative to mode 0 by the caller
    printf("side %d (%s): %.3f ticks per 64 MFMAs (mean over %d workgroups)\n", MODE, what, s / nwg / iters, nwg);
    hipFree(out); hipFree(st); hipFree(m);
    return 0;
}

int main() {
    if (run_check<0>(64, 500)) return 1;      // s_nop between the compare and the store, wait before the registers change
    if (run_check<1>(64, 500)) return 1;      // no s_nop
    if (run_check<2>(64, 500)) return 1;      // registers overwritten right behind the store
    if (run_check<2>(1024, 200)) return 1;
    run_side<0>("nothing");
    run_side<1>("4 v_cmp -> sgpr pairs per 8 MFMAs");
    run_side<2>("4 v_cmp + 2 s_store_dwordx4");
    run_side<3>("2 s_store_dwordx4");
    run_side<4>("1 s_load_dwordx8");
    run_side<5>("4 v_cndmask with sgpr conditions");
    run_side<6>("4 v_cmp + 4 v_cndmask");
    run_side<0>("nothing");
    return 0;
}
