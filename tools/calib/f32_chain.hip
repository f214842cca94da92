// Microbenchmarks for the fp32 MLP kernel's schedule (one 256-thread workgroup per CU, operands in registers):
//  chain<NACC>: does a chain of DEPENDENT v_mfma_f32_32x32x2_f32 (same accumulator as srcC and vDst) issue back to back?
//  side<MODE>:  what does side work between two MFMAs of such a chain cost?
//     0 nothing   1 one v_accvgpr_read of ANOTHER accumulator tile   2 one VGPR->VGPR v_max_i32   3 read + max (the kernel's ReLU of one element)
//     4 two reads + two max   5 v_accvgpr_read + v_max_i32 + v_accvgpr_write back   6 one ds_read_b128   7 MFMA B operand from an AGPR
//     8 s_mov_b32   9 s_nop 0   10 s_waitcnt lgkmcnt(0)   11 buffer-less global_load_dwordx4   12 ds_write_b32
//     16 global_load_lds_dwordx4 (LDS-DMA)   17 global_load_dword   18 global_load_dwordx4 with scalar base   19 the same behind every 2nd MFMA
//     20 global_store_dword with scalar base   21 with a 64-bit vector address   22 scalar base, behind every 4th MFMA
//     13 one v_max_i32 behind every 8th MFMA only   14 sixteen v_max_i32 behind every 8th MFMA   15 thirty-two behind every 8th
//  two<MODE>:   two INDEPENDENT chains (a[0:15], a[16:31]) alternating; MODE 0 nothing, 2 one v_max_i32 behind each MFMA, 4 four
//  swtch<N>:    chains of 8 MFMAs on alternating accumulators (a slice boundary every 8 MFMAs), N v_max_i32 AT the boundary
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int NACC>
__global__ __launch_bounds__(256, 1) void chain_kernel(float *out, int iters, unsigned long long *stamps) {
    f32x16 acc[NACC] = {};
    float a = 1.0f + threadIdx.x * 1e-3f, b[8];
    for (int i = 0; i < 8; ++i) b[i] = 0.5f + i * 1e-3f + threadIdx.x * 1e-4f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 64; ++s) acc[s % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b[s & 7], acc[s % NACC], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int n = 0; n < NACC; ++n)
        for (int r = 0; r < 16; ++r) s += acc[n][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) stamps[blockIdx.x] = t1 - t0;
}

template <int MODE>
__global__ __launch_bounds__(256, 1) void side_kernel(float *out, int iters, unsigned long long *stamps) {
    __shared__ float lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = i;
    __syncthreads();
    float a = 1.0f + threadIdx.x * 1e-3f, b = 0.5f + threadIdx.x * 1e-4f;
    float x0 = a, x1 = b, y0 = 0, y1 = 0;
    float4 dsv = {0, 0, 0, 0};
    const unsigned laddr = (threadIdx.x & 63) * 16 % 1024;
    // a[0:15] = the chain's accumulator, a[16:31] = another tile (the side work's operand), a32 = B operand of mode 7
    asm volatile("v_accvgpr_write_b32 a32, %0" ::"v"(b));
#define INIT(n) asm volatile("v_accvgpr_write_b32 a" #n ", %0" ::"v"(0.0f));
    INIT(0) INIT(1) INIT(2) INIT(3) INIT(4) INIT(5) INIT(6) INIT(7) INIT(8) INIT(9) INIT(10) INIT(11) INIT(12) INIT(13) INIT(14) INIT(15)
    INIT(16) INIT(17) INIT(18) INIT(19) INIT(20) INIT(21) INIT(22) INIT(23)
    asm volatile("s_mov_b32 m0, %0" ::"s"(2048u));
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 64; ++s) {
            if (MODE == 7) asm volatile("v_mfma_f32_32x32x2_f32 a[0:15], %0, a32, a[0:15]" ::"v"(a) : "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15");
            else asm volatile("v_mfma_f32_32x32x2_f32 a[0:15], %0, %1, a[0:15]" ::"v"(a), "v"(b) : "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15");
            if (MODE == 1) asm volatile("v_accvgpr_read_b32 %0, a16" : "=v"(y0));
            if (MODE == 2) asm volatile("v_max_i32 %0, 0, %1" : "=v"(y0) : "v"(x0));
            if (MODE == 3) asm volatile("v_accvgpr_read_b32 %0, a16\n\tv_max_i32 %0, 0, %0" : "=v"(y0));
            if (MODE == 4) asm volatile("v_accvgpr_read_b32 %0, a16\n\tv_accvgpr_read_b32 %1, a17\n\tv_max_i32 %0, 0, %0\n\tv_max_i32 %1, 0, %1" : "=v"(y0), "=v"(y1));
            if (MODE == 5) asm volatile("v_accvgpr_read_b32 %0, a16\n\tv_max_i32 %0, 0, %0\n\ts_nop 0\n\tv_accvgpr_write_b32 a17, %0" : "=v"(y0)::"a17");
            if (MODE == 6) asm volatile("ds_read_b128 %0, %1" : "=v"(dsv) : "v"(laddr));
            if (MODE == 8) asm volatile("s_mov_b32 s20, 0x12345" ::: "s20");
            if (MODE == 9) asm volatile("s_nop 0");
            if (MODE == 10) asm volatile("s_waitcnt lgkmcnt(0)");
            if (MODE == 11) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dsv) : "v"(out + (threadIdx.x & 63) * 4));
            if (MODE == 16) asm volatile("global_load_lds_dwordx4 %0, %1" ::"v"(laddr), "s"(out) : "memory");
            if (MODE == 17) asm volatile("global_load_dword %0, %1, off" : "=v"(dsv.x) : "v"(out + (threadIdx.x & 63)));
            if (MODE == 18) asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dsv) : "v"(laddr), "s"(out));
            if (MODE == 19 && s % 2 == 0) asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dsv) : "v"(laddr), "s"(out));
            if (MODE == 20) asm volatile("global_store_dword %0, %1, %2" ::"v"(laddr), "v"(x0), "s"(out + 256 * blockIdx.x) : "memory");
            if (MODE == 21) asm volatile("global_store_dword %0, %1, off" ::"v"(out + 64 * blockIdx.x + (threadIdx.x & 63)), "v"(x0) : "memory");
            if (MODE == 22 && s % 4 == 0) asm volatile("global_store_dword %0, %1, %2" ::"v"(laddr), "v"(x0), "s"(out + 256 * blockIdx.x) : "memory");
            if (MODE == 12) asm volatile("ds_write_b32 %0, %1" ::"v"(laddr), "v"(x0));
            if (MODE == 13 && s % 8 == 7) asm volatile("v_max_i32 %0, 0, %1" : "=v"(y0) : "v"(x0));
            if ((MODE == 14 || MODE == 15) && s % 8 == 7) {
#pragma unroll
                for (int k = 0; k < (MODE == 14 ? 16 : 32); ++k) asm volatile("v_max_i32 %0, 0, %1" : "=v"(y0) : "v"(x0));
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    float s = y0 + y1 + x1 + dsv.x + dsv.w, v;
#define RD(n) asm volatile("v_accvgpr_read_b32 %0, a" #n : "=v"(v)); s += v;
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7");
    RD(0) RD(5) RD(15) RD(17)
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) stamps[blockIdx.x] = t1 - t0;
}

#define ACC0 "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15"
#define ACC1 "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31"
template <int MODE, int SW>
__global__ __launch_bounds__(256, 1) void two_kernel(float *out, int iters, unsigned long long *stamps) {
    float a = 1.0f + threadIdx.x * 1e-3f, b = 0.5f + threadIdx.x * 1e-4f;
    float x0 = a, y0 = 0, v;
    for (int i = 0; i < 1; ++i) {
        INIT(0) INIT(1) INIT(2) INIT(3) INIT(4) INIT(5) INIT(6) INIT(7) INIT(8) INIT(9) INIT(10) INIT(11) INIT(12) INIT(13) INIT(14) INIT(15)
        INIT(16) INIT(17) INIT(18) INIT(19) INIT(20) INIT(21) INIT(22) INIT(23) INIT(24) INIT(25) INIT(26) INIT(27) INIT(28) INIT(29) INIT(30) INIT(31)
    }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 64; ++s) {
            const bool first = SW ? ((s / 8) % 2 == 0) : (s % 2 == 0);
            if (first) asm volatile("v_mfma_f32_32x32x2_f32 a[0:15], %0, %1, a[0:15]" ::"v"(a), "v"(b) : ACC0);
            else asm volatile("v_mfma_f32_32x32x2_f32 a[16:31], %0, %1, a[16:31]" ::"v"(a), "v"(b) : ACC1);
            if (!SW) {
#pragma unroll
                for (int k = 0; k < MODE; ++k) asm volatile("v_max_i32 %0, 0, %1" : "=v"(y0) : "v"(x0));
            } else if (s % 8 == 7) {
#pragma unroll
                for (int k = 0; k < MODE; ++k) asm volatile("v_max_i32 %0, 0, %1" : "=v"(y0) : "v"(x0));
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = y0;
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7");
    RD(0) RD(5) RD(15) RD(17)
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) stamps[blockIdx.x] = t1 - t0;
}

// Other MFMA shapes: KIND 0 v_mfma_f32_16x16x4_f32 (8 passes), 1 v_mfma_f32_16x16x32_bf16 (4 passes... see the cycles), 2 v_mfma_f32_32x32x16_bf16;
// NV v_max_i32 behind every MFMA, the MFMAs rotating over NACC independent accumulators.
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8v __attribute__((ext_vector_type(8)));
// mix<NV, NL, NM>: v_mfma_f32_16x16x32_bf16 on four rotating accumulators with NV independent VALU instructions, NL ds_read_b128 and NM
// scalar-base buffer-less global loads (NM = 2: one LDS-DMA piece instead) behind EVERY MFMA -- what one gap of the bf16 kernels can hold
template <int NV, int NL, int NM>
__global__ __launch_bounds__(256, 1) void mix_kernel(float *out, int iters, unsigned long long *stamps) {
    __shared__ float lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = i;
    __syncthreads();
    f32x4v acc4[4] = {};
    float a = 1.0f + threadIdx.x * 1e-3f, b = 0.5f + threadIdx.x * 1e-4f;
    bf16x8v ha, hb;
    for (int i = 0; i < 8; ++i) ha[i] = (__bf16)(a + i), hb[i] = (__bf16)(b + i);
    float x[4] = {a, b, a + b, a - b}, y[4] = {0, 0, 0, 0};
    float4 dsv[2] = {}, gv = {};
    const unsigned laddr = (threadIdx.x & 63) * 16;
    asm volatile("s_mov_b32 m0, %0" ::"s"(16384u));
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 64; ++s) {
            asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc4[s % 4]) : "v"(ha), "v"(hb));
#pragma unroll
            for (int k = 0; k < NV; ++k) asm volatile("v_max_i32 %0, 0, %1" : "=v"(y[k % 4]) : "v"(x[k % 4]));
#pragma unroll
            for (int k = 0; k < NL; ++k) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dsv[k % 2]) : "v"(laddr), "n"(1024 * (k % 2)));
            if (NM == 1) asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(gv) : "v"(laddr), "s"(out + 256 * blockIdx.x));
            if (NM == 2) asm volatile("global_load_lds_dwordx4 %0, %1" ::"v"(laddr), "s"(out + 256 * blockIdx.x) : "memory");
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    float sum = y[0] + y[1] + y[2] + y[3] + dsv[0].x + dsv[1].y + gv.x;
    for (int n = 0; n < 4; ++n) sum += acc4[n][0];
    out[blockIdx.x * 256 + threadIdx.x] = sum;
    if (threadIdx.x == 0) stamps[blockIdx.x] = t1 - t0;
}

template <int KIND, int NACC, int NV>
__global__ __launch_bounds__(256, 1) void shape_kernel(float *out, int iters, unsigned long long *stamps) {
    f32x4v acc4[NACC] = {};
    f32x16 acc16[NACC] = {};
    float a = 1.0f + threadIdx.x * 1e-3f, b = 0.5f + threadIdx.x * 1e-4f;
    bf16x8v ha, hb;
    for (int i = 0; i < 8; ++i) ha[i] = (__bf16)(a + i), hb[i] = (__bf16)(b + i);
    float x0 = a, y0 = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 64; ++s) {
            if (KIND == 0) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc4[s % NACC]) : "v"(a), "v"(b));
            if (KIND == 1) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc4[s % NACC]) : "v"(ha), "v"(hb));
            if (KIND == 2) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc16[s % NACC]) : "v"(ha), "v"(hb));
#pragma unroll
            for (int k = 0; k < NV; ++k) asm volatile("v_max_i32 %0, 0, %1" : "=v"(y0) : "v"(x0));
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = y0;
    for (int n = 0; n < NACC; ++n) s += acc4[n][0] + acc16[n][0];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) stamps[blockIdx.x] = t1 - t0;
}

template <class F>
static int run(const char *name, F kernel, int iters, float *out, unsigned long long *st, int cus) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(kernel, dim3(cus), dim3(256), 0, 0, out, iters, st);
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(kernel, dim3(cus), dim3(256), 0, 0, out, iters, st);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h(cus);
    CHECK(hipMemcpy(h.data(), st, 8 * cus, hipMemcpyDeviceToHost));
    double cyc = 0;
    for (auto v : h) cyc += (double)v;
    cyc /= cus;
    const double mf = (double)iters * 64;
    printf("%-28s %.3f ms, %.2f cycles per MFMA, %.1f TFLOP/s\n", name, ms, cyc / mf, mf * 4 * cus * 2.0 * 32 * 32 * 2 / (ms * 1e-3) / 1e12);
    return 0;
}
#define MIX(V, L, M) run("bf16 16x16x32 + valu " #V " ds_read " #L " vmem " #M, mix_kernel<V, L, M>, iters, out, st, cus)
#define SHAPE(K, N, V) run("shape " #K " acc " #N " valu " #V, shape_kernel<K, N, V>, iters, out, st, cus)

int main() {
    int cus = 256;
    float *out;
    unsigned long long *st;
    CHECK(hipMalloc(&out, 4 * 256 * cus));
    CHECK(hipMalloc(&st, 8 * cus));
    const int iters = 10000;
    run("chain, 1 accumulator", chain_kernel<1>, iters, out, st, cus);
    run("chain, 2 accumulators", chain_kernel<2>, iters, out, st, cus);
    run("side 0 nothing", side_kernel<0>, iters, out, st, cus);
    run("side 1 accvgpr_read", side_kernel<1>, iters, out, st, cus);
    run("side 2 v_max_i32", side_kernel<2>, iters, out, st, cus);
    run("side 3 read+max", side_kernel<3>, iters, out, st, cus);
    run("side 4 2x(read+max)", side_kernel<4>, iters, out, st, cus);
    run("side 5 read+max+write", side_kernel<5>, iters, out, st, cus);
    run("side 6 ds_read_b128", side_kernel<6>, iters, out, st, cus);
    run("side 7 B operand in AGPR", side_kernel<7>, iters, out, st, cus);
    run("side 8 s_mov_b32", side_kernel<8>, iters, out, st, cus);
    run("side 9 s_nop 0", side_kernel<9>, iters, out, st, cus);
    run("side 10 s_waitcnt", side_kernel<10>, iters, out, st, cus);
    run("side 11 global_load_dwordx4", side_kernel<11>, iters, out, st, cus);
    run("side 12 ds_write_b32", side_kernel<12>, iters, out, st, cus);
    run("side 20 global_store_dword saddr", side_kernel<20>, iters, out, st, cus);
    run("side 21 global_store_dword 64-bit", side_kernel<21>, iters, out, st, cus);
    run("side 22 store saddr every 4th", side_kernel<22>, iters, out, st, cus);
    run("side 16 global_load_lds_dwordx4", side_kernel<16>, iters, out, st, cus);
    run("side 17 global_load_dword", side_kernel<17>, iters, out, st, cus);
    run("side 18 global_load_dwordx4 saddr", side_kernel<18>, iters, out, st, cus);
    run("side 19 same, every 2nd MFMA", side_kernel<19>, iters, out, st, cus);
    run("side 13 1 max per 8 MFMAs", side_kernel<13>, iters, out, st, cus);
    run("side 14 16 max per 8 MFMAs", side_kernel<14>, iters, out, st, cus);
    run("side 15 32 max per 8 MFMAs", side_kernel<15>, iters, out, st, cus);
    run("two chains, nothing", two_kernel<0, 0>, iters, out, st, cus);
    run("two chains, 1 max each", two_kernel<1, 0>, iters, out, st, cus);
    run("two chains, 4 max each", two_kernel<4, 0>, iters, out, st, cus);
    run("two chains, 12 max each", two_kernel<12, 0>, iters, out, st, cus);
    run("switch per 8, nothing", two_kernel<0, 1>, iters, out, st, cus);
    run("switch per 8, 1 max at it", two_kernel<1, 1>, iters, out, st, cus);
    run("switch per 8, 8 max at it", two_kernel<8, 1>, iters, out, st, cus);
    run("switch per 8, 16 max at it", two_kernel<16, 1>, iters, out, st, cus);
    run("switch per 8, 32 max at it", two_kernel<32, 1>, iters, out, st, cus);
    printf("shapes: 0 = f32 16x16x4, 1 = bf16 16x16x32, 2 = bf16 32x32x16 (TFLOP/s column is for the f32 32x32x2 FLOP count: ignore)\n");
    SHAPE(0, 1, 0); SHAPE(0, 1, 1); SHAPE(0, 1, 2); SHAPE(0, 4, 0); SHAPE(0, 4, 1); SHAPE(0, 4, 2); SHAPE(0, 4, 4);
    SHAPE(1, 1, 0); SHAPE(1, 4, 0); SHAPE(1, 4, 1); SHAPE(1, 4, 2); SHAPE(1, 4, 3); SHAPE(1, 4, 4);
    MIX(0, 0, 0); MIX(1, 0, 0); MIX(2, 0, 0); MIX(3, 0, 0); MIX(4, 0, 0); MIX(0, 1, 0); MIX(1, 1, 0); MIX(2, 1, 0); MIX(0, 2, 0); MIX(1, 2, 0);
    MIX(0, 0, 1); MIX(1, 0, 1); MIX(1, 1, 1); MIX(0, 0, 2); MIX(1, 0, 2); MIX(1, 1, 2);
    SHAPE(2, 1, 0); SHAPE(2, 2, 0); SHAPE(2, 2, 1); SHAPE(2, 2, 2); SHAPE(2, 2, 4); SHAPE(2, 2, 8);
    return 0;
}
