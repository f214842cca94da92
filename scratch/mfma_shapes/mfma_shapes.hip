// Ceiling measurement for the bf16-MFMA MLP kernels (MI355X_MICROARCH.md "DVFS give-back" items 6 and 7):
// bare MFMA loops at the MLP kernel's per-wave output tile (32 out-rows x 64 samples, K = 256 per "layer"), one
// wave per SIMD, operands random, with the in-kernel clock stamped around the loop (s_memtime / s_memrealtime).
//   shape 0: v_mfma_f32_32x32x16_bf16   (2 MFMAs per 16-k step)
//   shape 1: v_mfma_f32_16x16x32_bf16   (8 MFMAs per 32-k step)
//   src 0: A fragments held in registers;   src 1: A fragments read from LDS (ds_read_b128) as in the MLP kernel
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form mfma_shapes.hip -o mfma_shapes
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <utility>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F &&f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F &&f) {
    static_for_impl(f, std::make_integer_sequence<int, N>{});
}

#define K 256
// LDS row stride of the weight image (bytes) that makes the b128 fragment reads bank-conflict free: +16 for the
// 32-row fragment (lane groups stay inside one k-half), +32 for the 16-row fragment (a lane group mixes two k-quarters)
#define ROWB0 (2 * K + 16)
#define ROWB1 (2 * K + 32)
#define NSLICE 4           // 32-row slices resident in LDS

struct Stamp {
    unsigned long long t0, r0, t1, r1;
};

template <int SHAPE, int SRC>
__global__ __launch_bounds__(256, 1) void loop_kernel(const bf16x8 *__restrict__ data, float *__restrict__ out,
                                                      Stamp *__restrict__ stamps, int iters,
                                                      const char *__restrict__ wstream) {
    // SRC 0: A in registers; 1: A from LDS; 2: + weight staging global -> VGPR -> LDS at the MLP kernel's rate (4 KiB per
    // wave per 32x64x256 tile); 3: the same staging by LDS-DMA; 4: SRC 2 + the bf16 re-pack VALU work (40 ops per tile)
    constexpr bool LDSA = SRC >= 1;
    __shared__ __attribute__((aligned(16))) char lds[NSLICE * 32 * ROWB1 + 16384];
    char *stage_dst = lds + NSLICE * 32 * ROWB1;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    f32x4 pf[4];
    if constexpr (SRC == 2 || SRC == 4)
        for (int i = 0; i < 4; ++i) pf[i] = *(const f32x4 *)(wstream + (wave * 4 + i) * 1024 + (threadIdx.x & 63) * 16);
    unsigned rp[8] = {};
    constexpr int ROWB = SHAPE == 0 ? ROWB0 : ROWB1;
    const int tid = threadIdx.x, lane = tid & 63;
    // fill LDS with random bf16
    for (int i = tid; i < (int)sizeof(lds) / 16; i += 256) ((bf16x8 *)lds)[i] = data[(i * 7 + blockIdx.x) & 4095];
    bf16x8 B[32];  // the activation file: 256 features x 64 samples of bf16 = 128 registers
#pragma unroll
    for (int i = 0; i < 32; ++i) B[i] = data[(tid * 32 + i + blockIdx.x * 17) & 4095];
    bf16x8 A[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) A[i] = data[(tid * 8 + i + 1000) & 4095];
    __syncthreads();
    f32x16 acc32[2] = {};
    f32x4 acc16[8] = {};
    unsigned long long t0 = 0, r0 = 0;
    if (stamps) {
        t0 = __builtin_amdgcn_s_memtime();
        r0 = __builtin_amdgcn_s_memrealtime();
    }
    for (int it = 0; it < iters; ++it) {
        const int sl = it & (NSLICE - 1);
        // the weight stream: 1.25 MiB image walked piece by piece (L2-resident, like the packed MLP weights)
        const char *wsrc = wstream + ((size_t)((it * 16 + wave * 4) & 1279) << 10) + lane * 16;
        // step q of 16: pieces are fetched in steps 0,4,8,12 and parked two steps later (previous iteration's for 0)
        auto extra = [&](auto qc, float accv) {
            constexpr int q = decltype(qc)::value;
            if constexpr (SRC == 2 || SRC == 4) {
                if constexpr (q % 4 == 0) {  // park the piece fetched one tile ago (4 pieces in flight), fetch the next
                    *(f32x4 *)(stage_dst + wave * 4096 + (q / 4) * 1024 + lane * 16) = pf[q / 4];
                    pf[q / 4] = *(const f32x4 *)(wsrc + (q / 4) * 1024);
                }
            }
            if constexpr (SRC == 3) {
                if constexpr (q % 4 == 0) {
                    unsigned keep, la = (unsigned)(size_t)(const __attribute__((address_space(3))) char *)(stage_dst + wave * 4096 + (q / 4) * 1024);
                    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                                 : "=&s"(keep) : "v"(wsrc + (q / 4) * 1024), "s"(la) : "memory");
                }
            }
            if constexpr (SRC == 4) {  // 40 re-pack instructions per tile: cvt_pk + pk_max (+ accvgpr write every other)
                typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
                typedef float f32x2 __attribute__((ext_vector_type(2)));
                typedef short s16x2 __attribute__((ext_vector_type(2)));
                f32x2 v = {accv, accv + 1.0f};
                bf16x2 r = __builtin_convertvector(v, bf16x2);
                const s16x2 z = {0, 0};
                r = __builtin_bit_cast(bf16x2, __builtin_elementwise_max(__builtin_bit_cast(s16x2, r), z));
                unsigned w = __builtin_bit_cast(unsigned, r), a;
                asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(a) : "v"(w));
                asm volatile("" ::"a"(a));
            }
        };
        if constexpr (SHAPE == 0) {
            // 16 k-steps of 16; A frag: lane l -> row l&31, k = 8(l>>5)+j
            const char *a_lane = lds + sl * 32 * ROWB + (lane & 31) * ROWB + 16 * (lane >> 5);
            bf16x8 ring[4];
            if constexpr (LDSA) {
                static_for<4>([&](auto p) { ring[p] = *(const bf16x8 *)(a_lane + 32 * p); });
            }
            __builtin_amdgcn_sched_barrier(0);
            static_for<16>([&](auto pc) {
                constexpr int p = decltype(pc)::value;
                const bf16x8 a = LDSA ? ring[p % 4] : A[p % 8];
#pragma unroll
                for (int c = 0; c < 2; ++c)
                    acc32[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, B[2 * p + c], acc32[c], 0, 0, 0);
                if constexpr (LDSA && p + 4 < 16) ring[p % 4] = *(const bf16x8 *)(a_lane + 32 * (p + 4));
                extra(std::integral_constant<int, p>{}, acc32[0][p]);
                __builtin_amdgcn_sched_barrier(0);
            });
        } else {
            // 8 k-steps of 32; A frag: lane l -> row l&15, k = 8(l>>4)+j ; two 16-row blocks per 32-row slice
            const char *a_lane = lds + sl * 32 * ROWB + (lane & 15) * ROWB + 16 * (lane >> 4);
            bf16x8 ring[4][2];
            if constexpr (LDSA) {
                static_for<4>([&](auto p) {
                    ring[p][0] = *(const bf16x8 *)(a_lane + 64 * p);
                    ring[p][1] = *(const bf16x8 *)(a_lane + 16 * ROWB + 64 * p);
                });
            }
            __builtin_amdgcn_sched_barrier(0);
            static_for<8>([&](auto pc) {
                constexpr int p = decltype(pc)::value;
                static_for<2>([&](auto rc) {
                    constexpr int r = decltype(rc)::value;
                    const bf16x8 a = LDSA ? ring[p % 4][r] : A[(2 * p + r) % 8];
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        acc16[4 * r + c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, B[4 * p + c], acc16[4 * r + c], 0, 0, 0);
                    if constexpr (LDSA && p + 4 < 8) ring[p % 4][r] = *(const bf16x8 *)(a_lane + r * 16 * ROWB + 64 * (p + 4));
                    extra(std::integral_constant<int, 2 * p + r>{}, acc16[(2 * p + r) & 7][r]);
                    __builtin_amdgcn_sched_barrier(0);
                });
            });
        }
    }
    if (stamps) {
        unsigned long long t1 = __builtin_amdgcn_s_memtime();
        unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
        if (tid == 0) stamps[blockIdx.x] = Stamp{t0, r0, t1, r1};
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc32[0][i] + acc32[1][i];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc16[i][0] + acc16[i][1] + acc16[i][2] + acc16[i][3];
    for (int i = 0; i < 8; ++i) s += (float)rp[i];
    s += ((const float *)stage_dst)[tid];
    if (s == 12345.678f) out[blockIdx.x * 256 + tid] = s;
}

// 8 waves per workgroup (two per SIMD), 32 samples per wave: the same 256-sample tile, A fragments feed two MFMAs each,
// every wave stages half as many pieces.  SRC as above (1: LDS reads, 2: + staging).
template <int SRC>
__global__ __launch_bounds__(512, 2) void loop_kernel8(const bf16x8 *__restrict__ data, float *__restrict__ out,
                                                       Stamp *__restrict__ stamps, int iters, const char *__restrict__ wstream) {
    __shared__ __attribute__((aligned(16))) char lds[NSLICE * 32 * ROWB1 + 16384];
    constexpr int ROWB = ROWB1;
    char *stage_dst = lds + NSLICE * 32 * ROWB1;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < (int)sizeof(lds) / 16; i += 512) ((bf16x8 *)lds)[i] = data[(i * 7 + blockIdx.x) & 4095];
    bf16x8 B[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) B[i] = data[(tid * 16 + i + blockIdx.x * 17) & 4095];
    f32x4 pf[2];
    if constexpr (SRC == 2)
        for (int i = 0; i < 2; ++i) pf[i] = *(const f32x4 *)(wstream + (wave * 2 + i) * 1024 + lane * 16);
    __syncthreads();
    f32x4 acc16[4] = {};
    unsigned long long t0 = 0, r0 = 0;
    if (stamps) {
        t0 = __builtin_amdgcn_s_memtime();
        r0 = __builtin_amdgcn_s_memrealtime();
    }
    for (int it = 0; it < iters; ++it) {
        const int sl = it & (NSLICE - 1);
        const char *wsrc = wstream + ((size_t)((it * 16 + wave * 2) & 1279) << 10) + lane * 16;
        const char *a_lane = lds + sl * 32 * ROWB + (lane & 15) * ROWB + 16 * (lane >> 4);
        bf16x8 ring[4][2];
        static_for<4>([&](auto p) {
            ring[p][0] = *(const bf16x8 *)(a_lane + 64 * p);
            ring[p][1] = *(const bf16x8 *)(a_lane + 16 * ROWB + 64 * p);
        });
        __builtin_amdgcn_sched_barrier(0);
        static_for<8>([&](auto pc) {
            constexpr int p = decltype(pc)::value;
            static_for<2>([&](auto rc) {
                constexpr int r = decltype(rc)::value;
                const bf16x8 a = ring[p % 4][r];
#pragma unroll
                for (int c = 0; c < 2; ++c)
                    acc16[2 * r + c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, B[2 * p + c], acc16[2 * r + c], 0, 0, 0);
                if constexpr (p + 4 < 8) ring[p % 4][r] = *(const bf16x8 *)(a_lane + r * 16 * ROWB + 64 * (p + 4));
                if constexpr (SRC == 2 && r == 0 && p % 4 == 0) {
                    *(f32x4 *)(stage_dst + wave * 2048 + (p / 4) * 1024 + lane * 16) = pf[p / 4];
                    pf[p / 4] = *(const f32x4 *)(wsrc + (p / 4) * 1024);
                }
                __builtin_amdgcn_sched_barrier(0);
            });
        });
    }
    if (stamps) {
        unsigned long long t1 = __builtin_amdgcn_s_memtime();
        unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
        if (tid == 0) stamps[blockIdx.x] = Stamp{t0, r0, t1, r1};
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) s += acc16[i][0] + acc16[i][1] + acc16[i][2] + acc16[i][3];
    s += ((const float *)stage_dst)[tid];
    if (s == 12345.678f) out[blockIdx.x * 512 + tid] = s;
}

#define CK(x)                                                                  \
    do {                                                                       \
        hipError_t e = (x);                                                    \
        if (e != hipSuccess) {                                                 \
            printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__);    \
            exit(1);                                                           \
        }                                                                      \
    } while (0)

template <int SHAPE, int SRC>
static void run(const char *name, const char *wstream, const bf16x8 *data, float *out, Stamp *stamps, int iters, int grid, bool zeros) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    // >= 2 s of back-to-back launches first (un-stamped build), then time, then one stamped pass
    float ms = 0;
    int reps = 0;
    for (int round = 0; round < 200; ++round) {
        CK(hipEventRecord(e0));
        for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((loop_kernel<SHAPE, SRC>), dim3(grid), dim3(256), 0, 0, data, out, (Stamp *)nullptr, iters, wstream);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float t;
        CK(hipEventElapsedTime(&t, e0, e1));
        ms += t;
        reps += 10;
        if (ms > 2500.f) break;
    }
    CK(hipEventRecord(e0));
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((loop_kernel<SHAPE, SRC>), dim3(grid), dim3(256), 0, 0, data, out, (Stamp *)nullptr, iters, wstream);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float t;
    CK(hipEventElapsedTime(&t, e0, e1));
    t /= 20;
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((loop_kernel<SHAPE, SRC>), dim3(grid), dim3(256), 0, 0, data, out, stamps, iters, wstream);
    CK(hipDeviceSynchronize());
    std::vector<Stamp> h(grid);
    CK(hipMemcpy(h.data(), stamps, grid * sizeof(Stamp), hipMemcpyDeviceToHost));
    std::vector<double> clk, cyc;
    for (auto &s : h) {
        clk.push_back((double)(s.t1 - s.t0) / (double)(s.r1 - s.r0) * 100.0);  // MHz
        cyc.push_back((double)(s.t1 - s.t0));
    }
    std::sort(clk.begin(), clk.end());
    std::sort(cyc.begin(), cyc.end());
    // per iteration per wave: 32 rows x 64 samples x 256 k MACs
    double flop = 2.0 * 32 * 64 * 256 * (double)iters * 4 * grid;
    double cyc_per_iter = cyc[grid / 2] / iters;
    printf("%-38s %s  %.4f ms  %7.1f TFLOP/s  frac %.3f | in-kernel clock median %.0f MHz (min %.0f max %.0f) | %.1f cycles per 32x64x256 tile (ideal 512)\n",
           name, zeros ? "zeros " : "random", t, flop / t / 1e9, flop / t / 1e9 / 2500.0, clk[grid / 2], clk.front(), clk.back(), cyc_per_iter);
    fflush(stdout);
}

template <int SRC>
static void run8(const char *name, const char *wstream, const bf16x8 *data, float *out, Stamp *stamps, int iters, int grid, bool zeros) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float ms = 0;
    for (int round = 0; round < 200 && ms < 2500.f; ++round) {
        CK(hipEventRecord(e0));
        for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((loop_kernel8<SRC>), dim3(grid), dim3(512), 0, 0, data, out, (Stamp *)nullptr, iters, wstream);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float t;
        CK(hipEventElapsedTime(&t, e0, e1));
        ms += t;
    }
    CK(hipEventRecord(e0));
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((loop_kernel8<SRC>), dim3(grid), dim3(512), 0, 0, data, out, (Stamp *)nullptr, iters, wstream);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float t;
    CK(hipEventElapsedTime(&t, e0, e1));
    t /= 20;
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((loop_kernel8<SRC>), dim3(grid), dim3(512), 0, 0, data, out, stamps, iters, wstream);
    CK(hipDeviceSynchronize());
    std::vector<Stamp> h(grid);
    CK(hipMemcpy(h.data(), stamps, grid * sizeof(Stamp), hipMemcpyDeviceToHost));
    std::vector<double> clk, cyc;
    for (auto &s : h) {
        clk.push_back((double)(s.t1 - s.t0) / (double)(s.r1 - s.r0) * 100.0);
        cyc.push_back((double)(s.t1 - s.t0));
    }
    std::sort(clk.begin(), clk.end());
    std::sort(cyc.begin(), cyc.end());
    double flop = 2.0 * 32 * 32 * 256 * (double)iters * 8 * grid;  // per iteration per wave: 32 rows x 32 samples x 256 k
    printf("%-38s %s  %.4f ms  %7.1f TFLOP/s  frac %.3f | in-kernel clock median %.0f MHz | %.1f cycles per wave iteration (two waves share a SIMD: ideal 1024)\n",
           name, zeros ? "zeros " : "random", t, flop / t / 1e9, flop / t / 1e9 / 2500.0, clk[grid / 2], cyc[grid / 2] / iters);
    fflush(stdout);
}

int main(int argc, char **argv) {
    int iters = argc > 1 ? atoi(argv[1]) : 2048;
    int grid = argc > 2 ? atoi(argv[2]) : 2048;
    std::vector<unsigned short> h(4096 * 8);
    srand(1);
    for (auto &v : h) {
        float f = (float)(rand() & 0xffffff) / 16777216.f * 2.f - 1.f;
        unsigned u;
        memcpy(&u, &f, 4);
        v = (unsigned short)((u + 0x7fff + ((u >> 16) & 1)) >> 16);
    }
    bf16x8 *data;
    float *out;
    Stamp *stamps;
    CK(hipMalloc(&data, h.size() * 2));
    CK(hipMalloc(&out, (size_t)grid * 256 * 4));
    CK(hipMalloc(&stamps, grid * sizeof(Stamp)));
    char *ws;
    CK(hipMalloc(&ws, 1280 * 1024 + 65536));
    {
        std::vector<unsigned short> hw((1280 * 1024 + 65536) / 2);
        for (size_t i = 0; i < hw.size(); ++i) hw[i] = h[i % h.size()];
        CK(hipMemcpy(ws, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
    }
    for (int pass = 0; pass < 2; ++pass) {
        bool zeros = pass == 1;
        if (zeros) CK(hipMemset(data, 0, h.size() * 2));
        else CK(hipMemcpy(data, h.data(), h.size() * 2, hipMemcpyHostToDevice));
        run<0, 0>("32x32x16 A in registers", ws, data, out, stamps, iters, grid, zeros);
        run<1, 0>("16x16x32 A in registers", ws, data, out, stamps, iters, grid, zeros);
        run<0, 1>("32x32x16 A from LDS", ws, data, out, stamps, iters, grid, zeros);
        run<1, 1>("16x16x32 A from LDS", ws, data, out, stamps, iters, grid, zeros);
        run<0, 2>("32x32x16 LDS + staging via VGPR", ws, data, out, stamps, iters, grid, zeros);
        run<1, 2>("16x16x32 LDS + staging via VGPR", ws, data, out, stamps, iters, grid, zeros);
        run<0, 3>("32x32x16 LDS + staging LDS-DMA", ws, data, out, stamps, iters, grid, zeros);
        run<1, 3>("16x16x32 LDS + staging LDS-DMA", ws, data, out, stamps, iters, grid, zeros);
        run<0, 4>("32x32x16 LDS + VGPR staging + repack", ws, data, out, stamps, iters, grid, zeros);
        run<1, 4>("16x16x32 LDS + VGPR staging + repack", ws, data, out, stamps, iters, grid, zeros);
        run8<1>("16x16x32 8 waves x 32 smp, LDS", ws, data, out, stamps, iters, grid, zeros);
        run8<2>("16x16x32 8 waves x 32 smp, LDS+staging", ws, data, out, stamps, iters, grid, zeros);
        run<0, 0>("32x32x16 A in registers (again)", ws, data, out, stamps, iters, grid, zeros);
    }
    return 0;
}
