// K2 (bf16), "two groups per weight pass": the same fused 8x256 MLP on v_mfma_f32_16x16x32_bf16 as mlp_bf16.hip (same transposed
// formulation, same fragment layouts, same arithmetic in the same order: bit-identical outputs), with a workgroup owning 512 samples --
// every wave two GROUPS of 64 -- so that a layer's weights are staged into LDS once and used by both groups one after the other: the
// L2 -> LDS weight stream drops to 54 % per sample.  The tile body is ONE block of assembly with every register assigned by its
// generator, gen_bf16_g2.py, whose header is the design note (registers, LDS slots, barriers, where the encoded features come from).
// This file is the shell around it: the weight image, the scalar arguments of the body, the tile loop.
#include "mlp_bf16_common.h"
#include <cstdlib>

#include "mlp_bf16_g2_tables.gen.inc"

// G2_HALF (mlp_f16_g2.hip compiles this file a second time with it): the fp16 tier -- fp16 operands (11 significant bits against
// bf16's 8), the body generated with the f16 forms of the MFMA and the re-pack conversion; everything else is shared.
#ifdef G2_HALF
#define G2_SYM(x) ddnerf_mlp_f16g2_##x
#define G2_G1_SYM(x) ddnerf_mlp_f16g1_##x
#define G2_ANY_SYM(x) ddnerf_mlp_f16_##x
#define G2_KERNEL mlp_f16g2_fwd_kernel
#define G2_PACK_KERNEL mlp_f16g2_pack_kernel
typedef _Float16 g2_elem;
#else
#define G2_SYM(x) ddnerf_mlp_bf16g2_##x
#define G2_G1_SYM(x) ddnerf_mlp_bf16g1_##x
#define G2_ANY_SYM(x) ddnerf_mlp_bf16_##x
#define G2_KERNEL mlp_bf16g2_fwd_kernel
#define G2_PACK_KERNEL mlp_bf16g2_pack_kernel
typedef __bf16 g2_elem;
extern "C" __attribute__((visibility("default"))) const char *ddnerf_bf16g2_generator_options(void) { return G2_GENERATOR_OPTIONS; }
#endif

#define G2_SLOT_BYTES (36 * 1024)
#define G2_LDS_BYTES (4 * G2_SLOT_BYTES)
#define G2_TILE 512

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

static constexpr int kG2K[11] = {96, 256, 256, 256, 256, 352, 256, 256, 256, 288, 128};
__host__ __device__ constexpr int g2_rowb(int K) { return 2 * K + 32; }

// ---- weight image: the slices of gen_bf16_g2.py's chunks (planes of 16 rows in k-order with row stride 2K + 32 bytes, then the 16 fp32
// biases), each chunk padded to whole 4-KiB rounds.  Parameter layout and row / column maps as in mlp_mfma16.inc (srcw / srcb).
struct G2Src {
    int w_src[13], b_src[13];
};
static G2Src g2_make_src(int depth_head) {
    G2Src p;
    static const int nout[13] = {256, 256, 256, 256, 256, 256, 256, 256, 256, 1, 128, 3, 2};
    static const int nin[13] = {96, 256, 256, 256, 256, 352, 256, 256, 256, 256, 283, 128, 128};
    int off = 0;
    for (int l = 0; l < 13; ++l) {
        p.w_src[l] = off;
        off += nout[l] * nin[l];
        p.b_src[l] = off;
        off += nout[l];
        if (l == 11 && !depth_head) {
            p.w_src[12] = p.b_src[12] = -1;
            break;
        }
    }
    return p;
}
__device__ __forceinline__ float g2_srcw(const float *__restrict__ P, const G2Src &pl, int l, int o, int c) {
    if (l == 5) return P[pl.w_src[5] + o * 352 + (c < 256 ? 96 + c : c - 256)];  // packed column order [h | xyz]; reference: cat(xyz, h)
    if (l <= 8) return P[pl.w_src[l] + o * kG2K[l] + c];
    if (l == 9) {
        if (o < 128) return c < 283 ? P[pl.w_src[10] + o * 283 + c] : 0.0f;
        if (o == 128) return c < 256 ? P[pl.w_src[9] + c] : 0.0f;
        return 0.0f;
    }
    if (o < 3) return P[pl.w_src[11] + o * 128 + c];
    if ((o == 4 || o == 5) && pl.w_src[12] >= 0) return P[pl.w_src[12] + (o - 4) * 128 + c];
    return 0.0f;
}
__device__ __forceinline__ float g2_srcb(const float *__restrict__ P, const G2Src &pl, int l, int o) {
    if (l <= 8) return P[pl.b_src[l] + o];
    if (l == 9) return o < 128 ? P[pl.b_src[10] + o] : (o == 128 ? P[pl.b_src[9]] : 0.0f);
    if (o < 3) return P[pl.b_src[11] + o];
    if ((o == 4 || o == 5) && pl.b_src[12] >= 0) return P[pl.b_src[12] + (o - 4)];
    return 0.0f;
}

__global__ __launch_bounds__(256) void G2_PACK_KERNEL(const float *__restrict__ P, G2Src pl, unsigned short *__restrict__ packed) {
    const int si = blockIdx.x;
    const int l = kG2Slice[si][0], K = kG2K[l], o0 = 16 * kG2Slice[si][1];
    unsigned short *dst = packed + kG2Slice[si][2] / 2;
    const int plane = 16 * g2_rowb(K);
    for (int idx = threadIdx.x; idx < (plane + 64) / 2; idx += 256) {
        const int r2 = idx * 2;
        unsigned short w;
        if (r2 < plane) {
            const int row = r2 / g2_rowb(K), col = (r2 % g2_rowb(K)) / 2;
            const float v = col < K ? g2_srcw(P, pl, l, o0 + row, korder32(col)) : 0.0f;
            w = __builtin_bit_cast(unsigned short, (g2_elem)v);
        } else {
            const int bi = (r2 - plane) / 4, half = ((r2 - plane) % 4) / 2;
            const unsigned u = __builtin_bit_cast(unsigned, g2_srcb(P, pl, l, o0 + bi));
            w = (unsigned short)(half ? (u >> 16) : (u & 0xffffu));
        }
        dst[idx] = w;
    }
}

DDN_EXPORT size_t G2_SYM(packed_bytes)(int depth_head) {
    (void)depth_head;
    return (size_t)G2_IMG_BYTES;
}
DDN_EXPORT int G2_SYM(pack)(const float *params, int depth_head, void *packed, ddnerf_stream_t stream) {
    DDN_REQUIRE(params && packed, DDNERF_E_ARG);
    DDN_REQUIRE(ddn_aligned(packed, 16), DDNERF_E_ALIGN);
    // (the chunk tails travel to LDS with their chunk and are never read: zeroed once, so that the image is a function of the parameters)
    if (hipMemsetAsync(packed, 0, (size_t)G2_IMG_BYTES, (hipStream_t)stream) != hipSuccess) return DDNERF_E_ARG;
    hipLaunchKernelGGL(G2_PACK_KERNEL, dim3(G2_NSLICE), dim3(256), 0, (hipStream_t)stream, params, g2_make_src(depth_head),
                       (unsigned short *)packed);
    return ddn_launch_status();
}

// ---- the kernel ----------------------------------------------------------------------------------------------------------------
#ifdef BF16_STAMP  // diagnostic build only: in-kernel clock, cycles per tile (six values per workgroup), the clock at the end of every
                   // period of the workgroup's last tile (192 values per workgroup behind those: slot 0 tile begin, slot p + 1 the end of period p, then the generator's optional per-block / per-k-step stamps)
__device__ unsigned long long *g_bf16g2_stamps;
DDN_EXPORT int ddnerf_debug_set_stamps_g2(void *p) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_bf16g2_stamps), &p, sizeof(p)); }
#define G2_BODY_D0 "mlp_bf16_g2_body_d0s.gen.inc"
#define G2_BODY_D1 "mlp_bf16_g2_body_d1s.gen.inc"
#elif defined(G2_HALF)
#define G2_BODY_D0 "mlp_f16_g2_body_d0.gen.inc"
#define G2_BODY_D1 "mlp_f16_g2_body_d1.gen.inc"
#else
#define G2_BODY_D0 "mlp_bf16_g2_body_d0.gen.inc"
#define G2_BODY_D1 "mlp_bf16_g2_body_d1.gen.inc"
#endif

// Everything vector lives in the body's own registers from its first-tile prologue on (the compiler is told that the body clobbers the
// whole vector file, and has nothing vector of its own alive across it: the loop below is scalar).
template <bool DEPTH_HEAD>
__global__ __launch_bounds__(256, 1) void G2_KERNEL(const char *__restrict__ feat, const char *__restrict__ packed,
                                                               float *__restrict__ raw, long M, long ntiles) {
    __shared__ __attribute__((aligned(16))) char lds[G2_LDS_BYTES];
    const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned lds0 = __builtin_amdgcn_readfirstlane(lds_addr_of(lds));
    // feature rows and outputs as raw buffers: the body puts the whole byte offset into the bounds-checked voffset, so rows past the end
    // load as zero and stores past the end are dropped (no branch, no clamp)
    // (resource words: base address, stride 0, number of bytes, the raw-buffer format word the compiler's own make_buffer_rsrc uses)
    const u32x4 frs = {(unsigned)(size_t)feat, (unsigned)((size_t)feat >> 32) & 0xffffu, (unsigned)(M * (2 * DDNERF_FEAT_LD)), 0x00020000u};
    const u32x4 rrs = {(unsigned)(size_t)raw, (unsigned)((size_t)raw >> 32) & 0xffffu, (unsigned)(M * (DEPTH_HEAD ? 24 : 16)), 0x00020000u};
    const unsigned plo = (unsigned)(size_t)packed, phi = (unsigned)((size_t)packed >> 32);
    const unsigned grid = gridDim.x, tile0 = blockIdx.x;
#ifdef BF16_STAMP
    // (layout of the stamp buffer: six values per workgroup for the whole grid, then 192 values per workgroup: sized by the GRID, so a
    // part with another CU count neither overruns the buffer nor overlaps the two areas)
    unsigned long long *const wg_stamps = g_bf16g2_stamps ? g_bf16g2_stamps + 6 * gridDim.x + 192 * blockIdx.x : nullptr;
    const unsigned long long st_entry = __builtin_amdgcn_s_memrealtime();
    const unsigned long long st_t0 = __builtin_amdgcn_s_memtime(), st_r0 = __builtin_amdgcn_s_memrealtime();
    unsigned done = 0;
#endif
    // (no peeling / unrolling: one copy of the body per kernel)
#pragma clang loop unroll(disable)
    for (unsigned tile = tile0; tile < (unsigned)ntiles; tile += grid) {
        if constexpr (DEPTH_HEAD) {
            asm volatile(
#include G2_BODY_D1
                :
                : "s"(frs), "s"(rrs), "s"(plo), "s"(phi), "s"(lds0), "s"(wave), "s"(tile), "s"(grid), "s"(tile0)
#ifdef BF16_STAMP
                  , "s"(wg_stamps)
#endif
                : G2_CLOBBERS);
        } else {
            asm volatile(
#include G2_BODY_D0
                :
                : "s"(frs), "s"(rrs), "s"(plo), "s"(phi), "s"(lds0), "s"(wave), "s"(tile), "s"(grid), "s"(tile0)
#ifdef BF16_STAMP
                  , "s"(wg_stamps)
#endif
                : G2_CLOBBERS);
        }
#ifdef BF16_STAMP
        ++done;
#endif
    }
    // (the last tile issued the next one's chunks and inputs: let them land before the wave ends)
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#ifdef BF16_STAMP
    {
        const unsigned long long st_t1 = __builtin_amdgcn_s_memtime(), st_r1 = __builtin_amdgcn_s_memrealtime();
        unsigned long long *sp = g_bf16g2_stamps;
        if (sp && threadIdx.x == 0) {
            sp[6 * blockIdx.x + 0] = st_t0;
            sp[6 * blockIdx.x + 1] = st_r0;
            sp[6 * blockIdx.x + 2] = st_t1;
            sp[6 * blockIdx.x + 3] = st_r1;
            sp[6 * blockIdx.x + 4] = done;
            sp[6 * blockIdx.x + 5] = st_entry;
        }
    }
#endif
}

// (the body addresses feature rows and outputs with 32-bit byte offsets, next tile included: launches of at most this many samples)
#define G2_MAX_LAUNCH (1L << 22)

DDN_EXPORT int G2_SYM(forward)(const void *feat, const void *packed, int depth_head, float *raw, long M, ddnerf_stream_t stream) {
    DDN_REQUIRE(feat && packed && raw, DDNERF_E_ARG);
    DDN_REQUIRE(M > 0, DDNERF_E_ARG);
    DDN_REQUIRE(ddn_aligned(feat, 16) && ddn_aligned(packed, 16) && ddn_aligned(raw, 16), DDNERF_E_ALIGN);
    const int n_cu = ddn_cu_count();
    for (long m0 = 0; m0 < M; m0 += G2_MAX_LAUNCH) {
        const long m = M - m0 < G2_MAX_LAUNCH ? M - m0 : G2_MAX_LAUNCH;
        const long ntiles = (m + G2_TILE - 1) / G2_TILE;
        const dim3 grid((unsigned)(ntiles < n_cu ? ntiles : n_cu));
        const char *f = (const char *)feat + (size_t)m0 * (2 * DDNERF_FEAT_LD);
        if (depth_head)
            hipLaunchKernelGGL(G2_KERNEL<true>, grid, dim3(256), 0, (hipStream_t)stream, f, (const char *)packed, raw + m0 * 6, m, ntiles);
        else
            hipLaunchKernelGGL(G2_KERNEL<false>, grid, dim3(256), 0, (hipStream_t)stream, f, (const char *)packed, raw + m0 * 4, m, ntiles);
    }
    return ddn_launch_status();
}

#ifdef BF16_DISPATCH
// ---- ddnerf_mlp_bf16_*: ONE weight image for both bf16 kernels (the one-group image, then this kernel's), the forward picks by size.
// Both kernels produce the same bits for the same sample (tests/test_hip_bf16_g2.py), so the choice is invisible in the results; it
// matters for time only: a 512-sample tile per workgroup wants enough tiles to fill the chip.
extern "C" size_t G2_G1_SYM(packed_bytes)(int depth_head);
extern "C" int G2_G1_SYM(pack)(const float *params, int depth_head, void *packed, ddnerf_stream_t stream);
extern "C" int G2_G1_SYM(forward)(const void *feat, const void *packed, int depth_head, float *raw, long M, ddnerf_stream_t stream);
static size_t g2_image_offset(int depth_head) { return (G2_G1_SYM(packed_bytes)(depth_head) + 255) & ~(size_t)255; }
// Which kernel?  Both are persistent (one workgroup per CU walks the tiles), so a launch takes ceil(tiles / CUs) rounds of one tile
// time; a 512-sample tile of the two-group kernel takes ~1.97x a 256-sample tile of the one-group kernel (it is 1 - 2 % faster per
// sample).  The two-group kernel runs when its rounds cost no more: always from 4 rounds on, never when its tiles would leave half
// the chip idle (65,536 samples: 128 tiles against 256).  DDNERF_BF16_G2_MIN=<samples> replaces the rule by a plain threshold
// (0: always the two-group kernel, -1: never).
static bool g2_wanted(long M) {
    // (a function-local static with an initialiser: initialised exactly once, thread-safe; -3 = no override)
    static const long forced = [] {
        const char *e = getenv("DDNERF_BF16_G2_MIN");
        return e && *e ? atol(e) : -3L;
    }();
    if (forced != -3) return forced >= 0 && M >= forced;
    const long n_cu = ddn_cu_count();
    const long r2 = ((M + G2_TILE - 1) / G2_TILE + n_cu - 1) / n_cu, r1 = ((M + 255) / 256 + n_cu - 1) / n_cu;
    return M >= 65536 && 197 * r2 <= 100 * r1;
}
DDN_EXPORT size_t G2_ANY_SYM(packed_bytes)(int depth_head) { return g2_image_offset(depth_head) + (size_t)G2_IMG_BYTES; }
DDN_EXPORT int G2_ANY_SYM(pack)(const float *params, int depth_head, void *packed, ddnerf_stream_t stream) {
    DDN_REQUIRE(params && packed, DDNERF_E_ARG);
    const int rc = G2_G1_SYM(pack)(params, depth_head, packed, stream);
    if (rc != 0) return rc;
    return G2_SYM(pack)(params, depth_head, (char *)packed + g2_image_offset(depth_head), stream);
}
DDN_EXPORT int G2_ANY_SYM(forward)(const void *feat, const void *packed, int depth_head, float *raw, long M, ddnerf_stream_t stream) {
    DDN_REQUIRE(feat && packed && raw, DDNERF_E_ARG);
    if (g2_wanted(M)) return G2_SYM(forward)(feat, (const char *)packed + g2_image_offset(depth_head), depth_head, raw, M, stream);
    return G2_G1_SYM(forward)(feat, packed, depth_head, raw, M, stream);
}
#endif
