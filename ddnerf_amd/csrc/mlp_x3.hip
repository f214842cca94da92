// K2 ("x3"): the whole 8x256 MLP (models/base_architectures.py:40-61, 103-126) as ONE kernel on the bf16 matrix
// cores at fp32-class accuracy.
//
// Every fp32 number splits exactly into hi + lo + r with hi = bf16(x), lo = bf16(x - hi), |r| <= 2^-17 |x|, so
//     w * a  =  hi_w hi_a  +  hi_w lo_a  +  lo_w hi_a  +  O(2^-16 |w a|):
// three v_mfma_f32_32x32x16_bf16 (fp32 accumulation) per 16 input features replace the eight v_mfma_f32_32x32x2_f32
// of the exact fp32 kernel -- 5.3x fewer matrix-pipe cycles.  Measured against an fp64 evaluation of the same network
// the outputs are off by ~1e-6 absolute (fp32 kernel: 6e-8; plain bf16: 6e-4): two orders of magnitude inside the
// 1e-4 RGB/depth parity bar, so this is the fast path of the fp32 tier, not a reduced-precision tier.
//
// Formulation as in mlp_bf16.hip: H_out^T[out, sample] = W[out, in] * H_in^T[in, sample]; A = W slices (hi and lo
// images) from LDS, B = the previous layer's activations (hi and lo files) in registers; the accumulator layout is the
// next layer's B layout in the permuted k order baked into the weight packing.  A wave owns 32 samples (the hi + lo
// files of this and the next layer are 4 x 64 registers: one wave per SIMD), 4 waves = 128 samples per workgroup
// share every LDS-staged weight byte; per k-step a wave issues 2 ds_read_b128 and 3 MFMAs (96 matrix-pipe cycles), so
// LDS reads sit at a third of the array's rate and the accumulator -> hi/lo re-pack (8 VALU ops per value pair) has
// 48 MFMAs per tile to hide behind.
#include "mlp_bf16_common.h"

// ---- schedule -------------------------------------------------------------------------------------------
#define NL 11
static constexpr int kK[NL] = {96, 256, 256, 256, 256, 352, 256, 256, 256, 288, 128};
static constexpr int kNB[NL] = {8, 8, 8, 8, 8, 8, 8, 8, 8, 5, 1};
// slices per stage (two stage buffers of <= 67 KiB): K=256 -> 2 (66.3 KiB), K=352 -> 1, K=96 -> 4, K=288 -> 1, K=128 -> 1
static constexpr int kSPS[NL] = {4, 2, 2, 2, 2, 1, 2, 2, 2, 1, 1};
#define STAGE_BYTES_MAX (67 * 1024)

// one slice = 32 rows of (K + 8) bf16 (hi), the same 32 rows (lo), then the 32 fp32 biases of those rows
__host__ __device__ constexpr int slice_bytes(int K) { return 64 * (2 * K + 16) + 128; }
__host__ __device__ constexpr int round_kib(int b) { return (b + 1023) / 1024 * 1024; }
__host__ __device__ constexpr int stage_bytes(int l, int st) {
    int first = st * kSPS[l];
    int ns = kNB[l] - first < kSPS[l] ? kNB[l] - first : kSPS[l];
    return round_kib(ns * slice_bytes(kK[l]));
}
__host__ __device__ constexpr int stages_of(int l) { return (kNB[l] + kSPS[l] - 1) / kSPS[l]; }

struct PlanX {
    int layer_off[NL];
    int w_src[13];
    int b_src[13];
    int total_bytes;
};

static PlanX make_plan_x(int depth_head) {
    PlanX p;
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
    off = 0;
    for (int l = 0; l < NL; ++l) {
        p.layer_off[l] = off;
        for (int st = 0; st < stages_of(l); ++st) off += stage_bytes(l, st);
    }
    p.total_bytes = off;
    return p;
}

DDN_EXPORT size_t ddnerf_mlp_x3_packed_bytes(int depth_head) { return (size_t)make_plan_x(depth_head).total_bytes; }

// same source mapping as the fp32 / bf16 kernels (see mlp_f32.hip)
__device__ __forceinline__ float srcw(const float *__restrict__ P, const PlanX &pl, int l, int o, int c) {
    if (l <= 8) return P[pl.w_src[l] + o * kK[l] + c];
    if (l == 9) {
        if (o < 128) return c < 283 ? P[pl.w_src[10] + o * 283 + c] : 0.0f;
        if (o == 128) return c < 256 ? P[pl.w_src[9] + c] : 0.0f;
        return 0.0f;
    }
    if (o < 3) return P[pl.w_src[11] + o * 128 + c];
    if ((o == 4 || o == 5) && pl.w_src[12] >= 0) return P[pl.w_src[12] + (o - 4) * 128 + c];
    return 0.0f;
}
__device__ __forceinline__ float srcb(const float *__restrict__ P, const PlanX &pl, int l, int o) {
    if (l <= 8) return P[pl.b_src[l] + o];
    if (l == 9) return o < 128 ? P[pl.b_src[10] + o] : (o == 128 ? P[pl.b_src[9]] : 0.0f);
    if (o < 3) return P[pl.b_src[11] + o];
    if ((o == 4 || o == 5) && pl.b_src[12] >= 0) return P[pl.b_src[12] + (o - 4)];
    return 0.0f;
}

__global__ void mlp_x3_pack_kernel(const float *__restrict__ P, PlanX pl, unsigned short *__restrict__ packed) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;  // one thread per 16-bit word of the packed buffer
    if (idx >= pl.total_bytes / 2) return;
    int byte = idx * 2, l = NL - 1;
    while (l > 0 && byte < pl.layer_off[l]) --l;
    int rel = byte - pl.layer_off[l], st = 0;
    while (rel >= stage_bytes(l, st)) {
        rel -= stage_bytes(l, st);
        ++st;
    }
    const int K = kK[l], rowb = 2 * K + 16;
    const int sl = rel / slice_bytes(K);
    const int first = st * kSPS[l];
    const int nsl = kNB[l] - first < kSPS[l] ? kNB[l] - first : kSPS[l];
    unsigned short w = 0;
    if (sl < nsl) {
        int r2 = rel - sl * slice_bytes(K);
        if (r2 < 64 * rowb) {
            const int part = r2 / (32 * rowb);  // 0: hi image, 1: lo image
            r2 -= part * 32 * rowb;
            int row = r2 / rowb, col = (r2 % rowb) / 2;
            float v = col < K ? srcw(P, pl, l, 32 * (first + sl) + row, korder(col)) : 0.0f;
            const __bf16 hi = (__bf16)v;
            const __bf16 b = part ? (__bf16)(v - (float)hi) : hi;
            w = __builtin_bit_cast(unsigned short, b);
        } else {  // fp32 bias of row (r2 - 64*rowb)/4, written as two 16-bit halves
            int bi = (r2 - 64 * rowb) / 4, half = ((r2 - 64 * rowb) % 4) / 2;
            unsigned u = __builtin_bit_cast(unsigned, srcb(P, pl, l, 32 * (first + sl) + bi));
            w = (unsigned short)(half ? (u >> 16) : (u & 0xffffu));
        }
    }
    packed[idx] = w;
}

DDN_EXPORT int ddnerf_mlp_x3_pack(const float *params, int depth_head, void *packed, ddnerf_stream_t stream) {
    DDN_REQUIRE(params && packed, DDNERF_E_ARG);
    DDN_REQUIRE(ddn_aligned(packed, 16), DDNERF_E_ALIGN);
    PlanX pl = make_plan_x(depth_head);
    int threads = pl.total_bytes / 2;
    hipLaunchKernelGGL(mlp_x3_pack_kernel, dim3((threads + 255) / 256), dim3(256), 0, (hipStream_t)stream, params, pl,
                       (unsigned short *)packed);
    return ddn_launch_status();
}

// ---- fused forward ----------------------------------------------------------------------------------------
#ifndef X3_DEPTH
#define X3_DEPTH 3  // A fragment pairs are read this many k-steps (x 96 matrix-pipe cycles) ahead of their MFMAs
#endif
#ifndef X3_PFD
#define X3_PFD 6  // weight pieces in flight per wave
#endif
#define WG_THREADS 256
#define WG_WAVES 4
#define WG_SAMPLES (WG_WAVES * 32)

__device__ __forceinline__ void dma_stage(const char *__restrict__ src, char *dst, int bytes, int wave, int lane) {
    const unsigned base = lds_addr_of(dst);
    for (int off = wave * 1024; off < bytes; off += WG_WAVES * 1024) dma_piece(src + off + lane * 16, base + off);
}

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// two floats -> the packed bf16 pair of their hi parts and the packed pair of their lo parts (x - hi is exact)
__device__ __forceinline__ void split_pair(float a, float b, unsigned &hi, unsigned &lo) {
    const f32x2 v = {a, b};
    hi = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
    const f32x2 r = {a - __builtin_bit_cast(float, hi << 16), b - __builtin_bit_cast(float, hi & 0xffff0000u)};
    lo = __builtin_bit_cast(unsigned, __builtin_convertvector(r, bf16x2));
}

// B-operand source of k-step ks of a layer: KIND 0 first layer (xyz, X[0..5]); 1 hidden (H[ks]); 2 skip layer
// (X[0..5] then H[0..15]); 3 dir layer (H[0..15] then X[6..7] = view dirs); 4 heads (H[0..7])
template <int KIND, int KS>
__device__ __forceinline__ const bf16x8 &bsrc(const bf16x8 (&H)[16], const bf16x8 (&X)[8]) {
    if constexpr (KIND == 0) return X[KS];
    else if constexpr (KIND == 2) {
        if constexpr (KS < 6) return X[KS];
        else return H[KS - 6];
    } else if constexpr (KIND == 3) {
        if constexpr (KS < 16) return H[KS];
        else return X[6 + (KS - 16)];
    } else return H[KS];
}

// One stage: NBLK 32-row slices out of LDS buffer `cur`.  The statement order IS the schedule (sched_barrier(0) per
// k-step): one A fragment pair -> 3 MFMAs on the block's accumulator tile (small terms first); pairs are read DEPTH
// k-steps ahead into a ring; the bias tile of block b+1 is read during block b; the hi/lo re-pack of the previous
// block's tile sits behind this block's first k-steps; the next stage's weight pieces travel global -> VGPR -> LDS in
// the gaps.  Hh/Hl: this layer's input files, Hnh/Hnl: the next layer's (k-steps 2(B0+b), 2(B0+b)+1 per block).
template <int KIND, int K, int NBLK, int B0, int NCONV, int DMA_BYTES, bool HN_AGPR>
__device__ __forceinline__ void stage_compute(const char *__restrict__ cur, const bf16x8 (&Hh)[16], const bf16x8 (&Hl)[16],
                                              const bf16x8 (&Xh)[8], const bf16x8 (&Xl)[8], bf16x8 (&Hnh)[16],
                                              bf16x8 (&Hnl)[16], f32x16 &keep, bool relu, int lane, int wave,
                                              const char *__restrict__ dma_src, char *dma_dst) {
    constexpr int NKS = K / 16, P = NBLK * NKS, DEPTH = X3_DEPTH, ROWB = 2 * K + 16, SLB = slice_bytes(K);
    constexpr int PIECES = DMA_BYTES / 1024, NP = (PIECES + WG_WAVES - 1) / WG_WAVES;  // pieces of this wave
    static_assert(NP <= P, "one staging slot per k-step at most");
    const char *a_lane = cur + (lane & 31) * ROWB + 16 * (lane >> 5);
    const char *b_lane = cur + 64 * ROWB + 16 * (lane >> 5);
    bf16x8 ring_h[DEPTH], ring_l[DEPTH];
    f32x16 acc[2];
    f32x4 pf[X3_PFD];
    auto read_a = [&](auto pc) {
        constexpr int p = decltype(pc)::value;
        const char *src = a_lane + (p / NKS) * SLB + 32 * (p % NKS);
        ring_h[p % DEPTH] = *(const bf16x8 *)(src);
        ring_l[p % DEPTH] = *(const bf16x8 *)(src + 32 * ROWB);
    };
    auto read_bias = [&](auto bc, auto gc) {  // rows 8g + 4h + (0..3) of block b -> accumulator registers 4g..4g+3
        constexpr int b = decltype(bc)::value, g = decltype(gc)::value;
        const f32x4 v = *(const f32x4 *)(b_lane + b * SLB + 32 * g);
        acc[b & 1][4 * g + 0] = v.x;
        acc[b & 1][4 * g + 1] = v.y;
        acc[b & 1][4 * g + 2] = v.z;
        acc[b & 1][4 * g + 3] = v.w;
    };
    const unsigned lane_off = wave * 1024 + lane * 16;  // uniform base + 32-bit lane offset: saddr-form global loads
    auto piece_ok = [&](int i) { return (i + 1) * WG_WAVES <= PIECES || wave + WG_WAVES * i < PIECES; };
    auto ld_piece = [&](auto ic) {
        constexpr int i = decltype(ic)::value;
#ifndef X3_NO_STAGE
        if (piece_ok(i)) pf[i % X3_PFD] = *(const f32x4 *)(dma_src + i * (WG_WAVES * 1024) + lane_off);
#endif
    };
    auto st_piece = [&](auto ic) {
        constexpr int i = decltype(ic)::value;
#ifndef X3_NO_STAGE
        if (piece_ok(i)) *(f32x4 *)(dma_dst + i * (WG_WAVES * 1024) + lane_off) = pf[i % X3_PFD];
#endif
    };
    auto repack = [&](auto pbc, auto uc) {  // unit u (registers 2u, 2u+1) of the tile of block pb -> one hi pair + one lo pair
        constexpr int pb = decltype(pbc)::value, u = decltype(uc)::value;
#ifdef X3_NO_REPACK
        if constexpr (u == 0) keep += acc[pb & 1];  // ablation: keeps the MFMAs alive without the hi/lo re-pack
        if constexpr (false) {
#else
        if constexpr (B0 + pb < NCONV) {
#endif
            float x0 = acc[pb & 1][2 * u], x1 = acc[pb & 1][2 * u + 1];
            if (relu) {  // on the bit patterns: one v_max_i32 each (fmaxf would add a canonicalising v_max per value)
                x0 = __builtin_bit_cast(float, max(__builtin_bit_cast(int, x0), 0));
                x1 = __builtin_bit_cast(float, max(__builtin_bit_cast(int, x1), 0));
            }
            unsigned hw, lw;
            split_pair(x0, x1, hw, lw);
            // registers 0..7 are k-step 2(B0+pb), 8..15 the next one; pair u&3 of that fragment
            u32x4 wh = __builtin_bit_cast(u32x4, Hnh[2 * (B0 + pb) + u / 4]);
            u32x4 wl = __builtin_bit_cast(u32x4, Hnl[2 * (B0 + pb) + u / 4]);
            wh[u & 3] = HN_AGPR ? to_agpr(hw) : hw;
            wl[u & 3] = HN_AGPR ? to_agpr(lw) : lw;
            Hnh[2 * (B0 + pb) + u / 4] = __builtin_bit_cast(bf16x8, wh);
            Hnl[2 * (B0 + pb) + u / 4] = __builtin_bit_cast(bf16x8, wl);
        }
    };
    static_for<4>([&](auto g) { read_bias(std::integral_constant<int, 0>{}, g); });
    static_for<(DEPTH < P ? DEPTH : P)>([&](auto p) { read_a(p); });
    __builtin_amdgcn_sched_barrier(0);
    static_for<P>([&](auto pc) {
        constexpr int p = decltype(pc)::value, b = p / NKS, ks = p % NKS;
        const bf16x8 &bh = bsrc<KIND, ks>(Hh, Xh), &bl = bsrc<KIND, ks>(Hl, Xl);
        // A 32x32x16 MFMA occupies the matrix pipe for 32 cycles but the wave's issue port for 4: everything else of the
        // step is dealt into the three 28-cycle gaps (one sched_barrier per gap), never piled up behind the third MFMA.
        constexpr int slot = NP > 0 ? (p * NP + P - 1) / P : 0;  // smallest i with (i * P) / NP >= p
        constexpr bool dma_here = NP > 0 && slot < NP && (slot * P) / (NP > 0 ? NP : 1) == p;
        // gap 1: this step's A-fragment prefetch + the park of the staging piece whose load has had PFD slots to land
        acc[b & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ring_l[p % DEPTH], bh, acc[b & 1], 0, 0, 0);
        if constexpr (dma_here && slot >= X3_PFD) st_piece(std::integral_constant<int, (slot >= X3_PFD ? slot - X3_PFD : 0)>{});
        __builtin_amdgcn_sched_barrier(0);
        // gap 2: re-pack of the previous block's tile, its 8 register pairs spread over k-steps 1 .. NKS-2 -- always
        // BEFORE the next block's bias piece: blocks b-1 and b+1 share an accumulator buffer, bias piece g (registers
        // 4g..4g+3) lands in step NKS-5+g, pair u (registers 2u, 2u+1) is re-packed no later than that
        acc[b & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ring_h[p % DEPTH], bl, acc[b & 1], 0, 0, 0);
        if constexpr (b > 0) {
            static_for<8>([&](auto uc) {
                constexpr int u = decltype(uc)::value;
                static_assert(1 + (u * (NKS - 2)) / 8 <= NKS - 5 + u / 2 || NBLK == 1, "re-pack after the bias overwrite");
                if constexpr (1 + (u * (NKS - 2)) / 8 == ks) repack(std::integral_constant<int, (b > 0 ? b - 1 : 0)>{}, uc);
            });
        }
        __builtin_amdgcn_sched_barrier(0);
        // gap 3: the A fragments DEPTH steps ahead, the next block's bias piece, the fetch of staging piece `slot`
        acc[b & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ring_h[p % DEPTH], bh, acc[b & 1], 0, 0, 0);
        if constexpr (p + DEPTH < P) read_a(std::integral_constant<int, p + DEPTH>{});
        if constexpr (b + 1 < NBLK && ks >= NKS - 5 && ks < NKS - 1)
            read_bias(std::integral_constant<int, b + 1>{}, std::integral_constant<int, ks - (NKS - 5)>{});
        if constexpr (dma_here) ld_piece(std::integral_constant<int, slot>{});
        __builtin_amdgcn_sched_barrier(0);
    });
    static_for<(NP < X3_PFD ? NP : X3_PFD)>([&](auto tc) {  // the pieces still in registers
        constexpr int t = decltype(tc)::value, first_left = NP < X3_PFD ? 0 : NP - X3_PFD;
        st_piece(std::integral_constant<int, first_left + t>{});
    });
    constexpr int lb = NBLK - 1;  // the stage's last block is re-packed right away
    static_for<8>([&](auto uc) { repack(std::integral_constant<int, lb>{}, uc); });
#ifndef X3_NO_REPACK
    keep = acc[lb & 1];  // the layer's last tile
#endif
    // the layer's last tile (heads: rgb / mu,sigma rows; dir layer: alpha row)
}

// One layer = its stages.  On entry the layer's first stage sits in LDS buffer PAR (parked by the previous stage).
// NEXT = layer whose first stage is fetched during this layer's last stage (-1: none).
template <int L, int KIND, int NEXT, int PAR, int NCONV, bool HN_AGPR>
__device__ __forceinline__ void layer(const char *__restrict__ &wp, char *lds, const bf16x8 (&Hh)[16],
                                      const bf16x8 (&Hl)[16], const bf16x8 (&Xh)[8], const bf16x8 (&Xl)[8],
                                      bf16x8 (&Hnh)[16], bf16x8 (&Hnl)[16], f32x16 &keep, bool relu, int wave, int lane) {
    constexpr int K = kK[L], NST = stages_of(L);
    static_for<NST>([&](auto stc) {
        constexpr int st = decltype(stc)::value;
        constexpr int first = st * kSPS[L];
        constexpr int nblk = kNB[L] - first < kSPS[L] ? kNB[L] - first : kSPS[L];
        char *cur = lds + ((PAR + st) & 1) * STAGE_BYTES_MAX;
        char *nxt = lds + ((PAR + st + 1) & 1) * STAGE_BYTES_MAX;
        dma_wait();       // (only the prologue's LDS-DMA of the very first stage is ever pending here)
#ifndef X3_NO_BARRIER
        __syncthreads();  // every wave has parked its pieces of stage `st`; the other buffer is free again
#endif
        wp += stage_bytes(L, st);
        constexpr int nbytes = st + 1 < NST ? stage_bytes(L, st + 1) : (NEXT >= 0 ? stage_bytes(NEXT >= 0 ? NEXT : 0, 0) : 0);
        stage_compute<KIND, K, nblk, first, NCONV, nbytes, HN_AGPR>(cur, Hh, Hl, Xh, Xl, Hnh, Hnl, keep, relu, lane, wave,
                                                                     wp, nxt);
    });
}

template <bool DEPTH_HEAD>
__global__ __launch_bounds__(WG_THREADS, 1) void mlp_x3_fwd_kernel(const float *__restrict__ feat,
                                                                   const char *__restrict__ packed,
                                                                   float *__restrict__ raw, long M) {
    __shared__ __attribute__((aligned(16))) char lds[2 * STAGE_BYTES_MAX];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, h = lane >> 5;
    const long m = (long)blockIdx.x * WG_SAMPLES + wave * 32 + j;
    const float *frow = feat + (size_t)(m < M ? m : M - 1) * DDNERF_FEAT_LD;
    bf16x8 HAh[16], HAl[16], HBh[16], HBl[16], Xh[8], Xl[8];
    f32x16 keep;
#ifdef X3_NO_REPACK
    for (int r = 0; r < 16; ++r) keep[r] = 0.f;
#endif
    const char *wp = packed;
    dma_stage(wp, lds, stage_bytes(0, 0), wave, lane);
    // fp32 features, natural column order: position j of lane half h in k-step g is column 16g + 8(j>>2) + 4h + (j&3)
    auto load_x = [&](auto g0c, auto g1c) {  // feature groups [g0, g1) (re-fetched when needed again, not held)
        constexpr int g0 = decltype(g0c)::value, g1 = decltype(g1c)::value;
#pragma unroll
        for (int g = g0; g < g1; ++g) {
#ifdef X3_NT_FEAT
            const f32x4 a = __builtin_nontemporal_load((const f32x4 *)(frow + 16 * g + 4 * h));
            const f32x4 b = __builtin_nontemporal_load((const f32x4 *)(frow + 16 * g + 8 + 4 * h));
#else
            const f32x4 a = *(const f32x4 *)(frow + 16 * g + 4 * h), b = *(const f32x4 *)(frow + 16 * g + 8 + 4 * h);
#endif
            unsigned wh[4], wl[4];
            split_pair(a.x, a.y, wh[0], wl[0]);
            split_pair(a.z, a.w, wh[1], wl[1]);
            split_pair(b.x, b.y, wh[2], wl[2]);
            split_pair(b.z, b.w, wh[3], wl[3]);
            Xh[g] = __builtin_bit_cast(bf16x8, u32x4{wh[0], wh[1], wh[2], wh[3]});
            Xl[g] = __builtin_bit_cast(bf16x8, u32x4{wl[0], wl[1], wl[2], wl[3]});
        }
    };
    load_x(std::integral_constant<int, 0>{}, std::integral_constant<int, 6>{});  // xyz: dead again after layer 0

    // parity of the LDS buffer holding a layer's first stage: layers 0..8 have an even number of stages (2, 4, 4, 4, 4, 8,
    // 4, 4, 4) so layers 0..9 start in buffer 0; the dir layer has 5, so the heads start in buffer 1
    layer<0, 0, 1, 0, 8, true>(wp, lds, HAh, HAl, Xh, Xl, HAh, HAl, keep, true, wave, lane);   // 96 -> 256 (H unused: KIND 0)
    layer<1, 1, 2, 0, 8, false>(wp, lds, HAh, HAl, Xh, Xl, HBh, HBl, keep, true, wave, lane);
    layer<2, 1, 3, 0, 8, true>(wp, lds, HBh, HBl, Xh, Xl, HAh, HAl, keep, true, wave, lane);
    layer<3, 1, 4, 0, 8, false>(wp, lds, HAh, HAl, Xh, Xl, HBh, HBl, keep, true, wave, lane);
    layer<4, 1, 5, 0, 8, true>(wp, lds, HBh, HBl, Xh, Xl, HAh, HAl, keep, true, wave, lane);
    load_x(std::integral_constant<int, 0>{}, std::integral_constant<int, 6>{});
    layer<5, 2, 6, 0, 8, false>(wp, lds, HAh, HAl, Xh, Xl, HBh, HBl, keep, true, wave, lane);  // cat(xyz, h) 352 -> 256
    layer<6, 1, 7, 0, 8, true>(wp, lds, HBh, HBl, Xh, Xl, HAh, HAl, keep, true, wave, lane);
    layer<7, 1, 8, 0, 8, false>(wp, lds, HAh, HAl, Xh, Xl, HBh, HBl, keep, true, wave, lane);
    layer<8, 1, 9, 0, 8, true>(wp, lds, HBh, HBl, Xh, Xl, HAh, HAl, keep, false, wave, lane);  // fc_feat: no activation
    load_x(std::integral_constant<int, 6>{}, std::integral_constant<int, 8>{});                // view-dir columns
    layer<9, 3, 10, 0, 4, false>(wp, lds, HAh, HAl, Xh, Xl, HBh, HBl, keep, true, wave, lane); // dir layer (128, ReLU) + alpha row
    const float alpha = keep[0];                                       // row 128 = block 4, register 0, lane half 0
    layer<10, 4, -1, 1, 0, true>(wp, lds, HBh, HBl, Xh, Xl, HAh, HAl, keep, false, wave, lane); // heads

    if (m < M) {
        if (DEPTH_HEAD) {
            float *op = raw + (size_t)m * 6;
            if (h == 0) {
                *(float2 *)(op) = make_float2(keep[0], keep[1]);
                *(float2 *)(op + 2) = make_float2(keep[2], alpha);
            } else {
                *(float2 *)(op + 4) = make_float2(keep[0], keep[1]);  // rows 4, 5 = raw mu, raw sigma
            }
        } else if (h == 0) {
            *(f32x4 *)(raw + (size_t)m * 4) = f32x4{keep[0], keep[1], keep[2], alpha};
        }
    }
}

DDN_EXPORT int ddnerf_mlp_x3_forward(const float *feat, const void *packed, int depth_head, float *raw, long M,
                                     ddnerf_stream_t stream) {
    DDN_REQUIRE(feat && packed && raw, DDNERF_E_ARG);
    DDN_REQUIRE(M > 0, DDNERF_E_ARG);
    DDN_REQUIRE(ddn_aligned(feat, 16) && ddn_aligned(packed, 16) && ddn_aligned(raw, 16), DDNERF_E_ALIGN);
    dim3 grid((unsigned)((M + WG_SAMPLES - 1) / WG_SAMPLES));
    if (depth_head)
        hipLaunchKernelGGL(mlp_x3_fwd_kernel<true>, grid, dim3(WG_THREADS), 0, (hipStream_t)stream, feat,
                           (const char *)packed, raw, M);
    else
        hipLaunchKernelGGL(mlp_x3_fwd_kernel<false>, grid, dim3(WG_THREADS), 0, (hipStream_t)stream, feat,
                           (const char *)packed, raw, M);
    return ddn_launch_status();
}
