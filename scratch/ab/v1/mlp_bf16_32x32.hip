// K2 (bf16): the whole 8x256 MLP (models/base_architectures.py:40-61, 103-126) as ONE kernel on the bf16
// matrix cores (v_mfma_f32_32x32x16_bf16, fp32 accumulation) -- the north-star roofline kernel.
//
// Same transposed formulation as the fp32 kernel:  H_out^T[out, sample] = W[out, in] * H_in^T[in, sample]
//   A operand = 32 out-rows x 16 in-features of W, read from LDS (ds_read_b128 per lane);
//   B operand = 16 in-features x 32 samples of the previous layer's output.  A 32x32 fp32 accumulator tile has
//               the sample on the lane and the feature on the register; converting registers 8s..8s+7 pairwise to
//               bf16 gives the B fragment of k-step s with NO lane movement -- in a permuted k order
//               (element j of lane half h = feature 16s + 8(j>>2) + 4h + (j&3)); W is packed in the same k
//               order (per 16 columns the four quads are stored [0,2,1,3]), and so are the bf16 feature rows the
//               encode kernel writes (include/ddnerf_hip.h, DDNERF "k-order").
// Activations never leave registers.  A wave owns 64 samples = two 32-sample column blocks, so every A fragment
// read from LDS feeds TWO MFMAs: measured on the first version (8 waves x 32 samples) the LDS array, not the matrix
// pipe, was the limiter (A reads + weight-stage writes ~80 % of LDS cycles).  4 waves (one per SIMD, 512-register
// file) = 256 samples per workgroup share every LDS-staged weight byte; a finished tile is re-packed to bf16 into
// the NEXT layer's B file right away (ping-pong), so only four accumulator tiles are live.
//
// Weights: repacked once per update into the exact LDS image -- 32-row slices, row stride K+8 bf16 (the 16-byte
// pad makes the b128 fragment reads bank-conflict free), grouped in STAGES of <= 66 KiB that are 1-KiB multiples --
// stored in consumption order.  While stage s feeds the MFMAs, stage s+1 is moved global -> 4 VGPRs -> LDS one 1-KiB
// piece per wave at a time, spread over the MFMA stream (LDS-DMA would need no VGPRs, but one global_load_lds costs
// its wave 60-180 issue cycles, and -- issued through the builtin -- makes hipcc turn every counted lgkmcnt(N) of the
// loop into lgkmcnt(0)); two stage buffers, one barrier per stage.
#include "../../../ddnerf_amd/csrc/mlp_bf16_common.h"

// ---- schedule -------------------------------------------------------------------------------------------
// 11 packed layers as in the fp32 kernel: K (in) / NB (32-row out blocks):
#define NL 11
static constexpr int kK[NL] = {96, 256, 256, 256, 256, 352, 256, 256, 256, 288, 128};
static constexpr int kNB[NL] = {8, 8, 8, 8, 8, 8, 8, 8, 8, 5, 1};
// slices per stage (a stage must fit one LDS buffer): K=256 -> 4 (66.5 KiB), K=352 -> 2, K=96 -> 8, K=288 -> 3+2, K=128 -> 1
static constexpr int kSPS[NL] = {8, 4, 4, 4, 4, 2, 4, 4, 4, 3, 1};
#define STAGE_BYTES_MAX (67 * 1024)

// one slice = 32 rows of (K + 8) bf16, then the 32 fp32 biases of those rows
__host__ __device__ constexpr int slice_bytes(int K) { return 32 * (2 * K + 16) + 128; }
__host__ __device__ constexpr int round_kib(int b) { return (b + 1023) / 1024 * 1024; }
// bytes of stage `st` (0-based) of layer l, padded to a 1-KiB multiple
__host__ __device__ constexpr int stage_bytes(int l, int st) {
    int first = st * kSPS[l];
    int ns = kNB[l] - first < kSPS[l] ? kNB[l] - first : kSPS[l];
    return round_kib(ns * slice_bytes(kK[l]));
}
__host__ __device__ constexpr int stages_of(int l) { return (kNB[l] + kSPS[l] - 1) / kSPS[l]; }

struct PlanB {
    int layer_off[NL];  // byte offset of the layer's first stage in the packed buffer
    int w_src[13];
    int b_src[13];
    int total_bytes;
};

static PlanB make_plan_b(int depth_head) {
    PlanB p;
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

DDN_EXPORT size_t ddnerf_mlp_bf16v1_packed_bytes(int depth_head) { return (size_t)make_plan_b(depth_head).total_bytes; }

// same source mapping as the fp32 kernel (see mlp_f32.hip)
__device__ __forceinline__ float srcw(const float *__restrict__ P, const PlanB &pl, int l, int o, int c) {
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
__device__ __forceinline__ float srcb(const float *__restrict__ P, const PlanB &pl, int l, int o) {
    if (l <= 8) return P[pl.b_src[l] + o];
    if (l == 9) return o < 128 ? P[pl.b_src[10] + o] : (o == 128 ? P[pl.b_src[9]] : 0.0f);
    if (o < 3) return P[pl.b_src[11] + o];
    if ((o == 4 || o == 5) && pl.b_src[12] >= 0) return P[pl.b_src[12] + (o - 4)];
    return 0.0f;
}

__global__ void mlp_bf16_pack_kernel(const float *__restrict__ P, PlanB pl, unsigned short *__restrict__ packed) {
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
    const int sl = rel / slice_bytes(K);  // slice inside the stage
    const int first = st * kSPS[l];
    const int nsl = kNB[l] - first < kSPS[l] ? kNB[l] - first : kSPS[l];
    unsigned short w = 0;
    if (sl < nsl) {
        int r2 = rel - sl * slice_bytes(K);
        if (r2 < 32 * rowb) {
            int row = r2 / rowb, col = (r2 % rowb) / 2;
            float v = col < K ? srcw(P, pl, l, 32 * (first + sl) + row, korder(col)) : 0.0f;
            __bf16 b = (__bf16)v;
            w = __builtin_bit_cast(unsigned short, b);
        } else {  // fp32 bias of row (r2 - 32*rowb)/4, written as two 16-bit halves
            int bi = (r2 - 32 * rowb) / 4, half = ((r2 - 32 * rowb) % 4) / 2;
            unsigned u = __builtin_bit_cast(unsigned, srcb(P, pl, l, 32 * (first + sl) + bi));
            w = (unsigned short)(half ? (u >> 16) : (u & 0xffffu));
        }
    }
    packed[idx] = w;
}

DDN_EXPORT int ddnerf_mlp_bf16v1_pack(const float *params, int depth_head, void *packed, ddnerf_stream_t stream) {
    DDN_REQUIRE(params && packed, DDNERF_E_ARG);
    DDN_REQUIRE(ddn_aligned(packed, 16), DDNERF_E_ALIGN);
    PlanB pl = make_plan_b(depth_head);
    int threads = pl.total_bytes / 2;
    hipLaunchKernelGGL(mlp_bf16_pack_kernel, dim3((threads + 255) / 256), dim3(256), 0, (hipStream_t)stream, params, pl,
                       (unsigned short *)packed);
    return ddn_launch_status();
}

// ---- fused forward ----------------------------------------------------------------------------------------
#ifndef BF16_DEPTH
#define BF16_DEPTH 4  // A fragments are read this many k-steps ahead of their MFMAs
#endif
#ifndef BF16_PFD
#define BF16_PFD 4  // weight pieces in flight per wave (load-to-park distance in DMA periods)
#endif
#define WG_THREADS 256
#define WG_WAVES 4
#define NCOL 2                       // 32-sample column blocks per wave
#define WG_SAMPLES (WG_WAVES * NCOL * 32)

__device__ __forceinline__ void dma_stage(const char *__restrict__ src, char *dst, int bytes, int wave, int lane) {
    const unsigned base = lds_addr_of(dst);
    for (int off = wave * 1024; off < bytes; off += WG_WAVES * 1024) dma_piece(src + off + lane * 16, base + off);
}

// two floats -> one packed bf16 pair (one v_cvt_pk_bf16_f32); ReLU on the bf16 bit patterns (v_pk_max_i16)
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef short s16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack_bf16(float a, float b, bool relu) {
    f32x2 v = {a, b};
    bf16x2 r = __builtin_convertvector(v, bf16x2);
    if (relu) {
        const s16x2 z = {0, 0};
        r = __builtin_bit_cast(bf16x2, __builtin_elementwise_max(__builtin_bit_cast(s16x2, r), z));
    }
    return __builtin_bit_cast(unsigned, r);
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

// One stage: NBLK 32-row slices out of LDS buffer `cur`.  The order of the statements below IS the instruction
// schedule (sched_barrier(0) after every step keeps hipcc from re-serialising it): one A fragment -> NCOL MFMAs;
// A fragments are read DEPTH k-steps ahead into a ring; the bias tile of block b+1 (its accumulators' start value,
// shared by the column blocks) is read during block b; the bf16 re-pack of the previous block's tiles is placed
// behind this block's first MFMAs; the next stage's weight pieces travel global -> VGPR -> LDS in the gaps.
template <int KIND, int K, int NBLK, int B0, int NCONV, int DMA_BYTES, bool HN_AGPR>
__device__ __forceinline__ void stage_compute(const char *__restrict__ cur, const bf16x8 (&H)[NCOL][16],
                                              const bf16x8 (&X)[NCOL][8], bf16x8 (&Hn)[NCOL][16],
                                              f32x16 (&keep)[NCOL][2], bool relu, int lane, int wave,
                                              const char *__restrict__ dma_src, char *dma_dst) {
    constexpr int NKS = K / 16, P = NBLK * NKS, DEPTH = BF16_DEPTH, ROWB = 2 * K + 16, SLB = slice_bytes(K);
    constexpr int PIECES = DMA_BYTES / 1024, NP = (PIECES + WG_WAVES - 1) / WG_WAVES;  // pieces of this wave
    constexpr int DMA_EVERY = NP > 0 ? P / NP : 1;
    static_assert(NP == 0 || DMA_EVERY >= 2, "every piece needs a load slot and a store slot");
    const char *a_lane = cur + (lane & 31) * ROWB + 16 * (lane >> 5);
    const char *b_lane = cur + 32 * ROWB + 16 * (lane >> 5);
    bf16x8 ring[DEPTH];
    f32x16 acc[2][NCOL];
    f32x4 pf[BF16_PFD];  // weight pieces in flight (global -> VGPR -> LDS)
    auto read_a = [&](auto pc) {
        constexpr int p = decltype(pc)::value;
        ring[p % DEPTH] = *(const bf16x8 *)(a_lane + (p / NKS) * SLB + 32 * (p % NKS));
    };
    auto read_bias = [&](auto bc, auto gc) {  // rows 8g + 4h + (0..3) of block b -> accumulator registers 4g..4g+3
        constexpr int b = decltype(bc)::value, g = decltype(gc)::value;
        const f32x4 v = *(const f32x4 *)(b_lane + b * SLB + 32 * g);
#pragma unroll
        for (int c = 0; c < NCOL; ++c) {
            acc[b & 1][c][4 * g + 0] = v.x;
            acc[b & 1][c][4 * g + 1] = v.y;
            acc[b & 1][c][4 * g + 2] = v.z;
            acc[b & 1][c][4 * g + 3] = v.w;
        }
    };
    auto piece_ok = [&](int i) { return (i + 1) * WG_WAVES <= PIECES || wave + WG_WAVES * i < PIECES; };
    auto ld_piece = [&](auto ic) {
        constexpr int i = decltype(ic)::value;
        if (piece_ok(i)) pf[i % BF16_PFD] = *(const f32x4 *)(dma_src + (wave + WG_WAVES * i) * 1024 + lane * 16);
    };
    auto st_piece = [&](auto ic) {
        constexpr int i = decltype(ic)::value;
        if (piece_ok(i)) *(f32x4 *)(dma_dst + (wave + WG_WAVES * i) * 1024 + lane * 16) = pf[i % BF16_PFD];
    };
    auto repack = [&](auto pbc, auto cc, auto qc) {  // quarter q of tile (block pb, column c) -> 2 packed pairs of Hn
        constexpr int pb = decltype(pbc)::value, c = decltype(cc)::value, q = decltype(qc)::value;
        if constexpr (B0 + pb < NCONV) {
            u32x4 w = __builtin_bit_cast(u32x4, Hn[c][2 * (B0 + pb) + q / 2]);
            unsigned w0 = pack_bf16(acc[pb & 1][c][4 * q + 0], acc[pb & 1][c][4 * q + 1], relu);
            unsigned w1 = pack_bf16(acc[pb & 1][c][4 * q + 2], acc[pb & 1][c][4 * q + 3], relu);
            w[2 * (q & 1) + 0] = HN_AGPR ? to_agpr(w0) : w0;
            w[2 * (q & 1) + 1] = HN_AGPR ? to_agpr(w1) : w1;
            Hn[c][2 * (B0 + pb) + q / 2] = __builtin_bit_cast(bf16x8, w);
        }
    };
    static_for<4>([&](auto g) { read_bias(std::integral_constant<int, 0>{}, g); });
    static_for<(DEPTH < P ? DEPTH : P)>([&](auto p) { read_a(p); });
    __builtin_amdgcn_sched_barrier(0);
    static_for<P>([&](auto pc) {
        constexpr int p = decltype(pc)::value, b = p / NKS, ks = p % NKS;
#pragma unroll
        for (int c = 0; c < NCOL; ++c)
            acc[b & 1][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ring[p % DEPTH], bsrc<KIND, ks>(H[c], X[c]),
                                                                    acc[b & 1][c], 0, 0, 0);
        if constexpr (p + DEPTH < P) read_a(std::integral_constant<int, p + DEPTH>{});
        // Blocks b-1 and b+1 share an accumulator buffer: the bias piece g of block b+1 (registers 4g..4g+3, step NKS-5+g)
        // must not land before quarter g of block b-1 has been re-packed.  For K >= 160 the re-pack is always steps
        // ahead and runs LAST in the step (its VALU work then overlaps the LDS / global latencies issued before it); for
        // K = 96 both walk the quarters in the same steps, so there the re-pack comes first.
        constexpr bool REPACK_FIRST = NKS < 10;
        auto repack_step = [&]() {
            if constexpr (b > 0) {  // re-pack the previous block's tiles: 4 quarters x NCOL units spread over this block
                static_for<4 * NCOL>([&](auto uc) {
                    constexpr int u = decltype(uc)::value;
                    static_assert(REPACK_FIRST || 1 + (u * (NKS - 2)) / (4 * NCOL) < NKS - 5 + u / NCOL, "bias lands on a live quarter");
                    if constexpr (1 + (u * (NKS - 2)) / (4 * NCOL) == ks)
                        repack(std::integral_constant<int, (b > 0 ? b - 1 : 0)>{}, std::integral_constant<int, u % NCOL>{},
                               std::integral_constant<int, u / NCOL>{});
                });
            }
        };
        if constexpr (REPACK_FIRST) repack_step();
        if constexpr (b + 1 < NBLK && ks >= NKS - 5 && ks < NKS - 1)  // next block's bias, one piece per step
            read_bias(std::integral_constant<int, b + 1>{}, std::integral_constant<int, ks - (NKS - 5)>{});
        if constexpr (NP > 0 && p / DMA_EVERY < NP && p % DMA_EVERY == 0) {
            // slot of piece i: first park piece i-2 (its load has had two periods, > 500 cycles, to land), then fetch i
            constexpr int i = p / DMA_EVERY;
            if constexpr (i >= BF16_PFD) st_piece(std::integral_constant<int, (i >= BF16_PFD ? i - BF16_PFD : 0)>{});
            ld_piece(std::integral_constant<int, i>{});
        }
        if constexpr (!REPACK_FIRST) repack_step();
        __builtin_amdgcn_sched_barrier(0);
    });
    static_for<(NP < BF16_PFD ? NP : BF16_PFD)>([&](auto tc) {  // the pieces still in registers
        constexpr int t = decltype(tc)::value, first_left = NP < BF16_PFD ? 0 : NP - BF16_PFD;
        st_piece(std::integral_constant<int, first_left + t>{});
    });
    constexpr int lb = NBLK - 1;  // the stage's last block is re-packed right away
    static_for<NCOL>([&](auto cc) {
        static_for<4>([&](auto qc) { repack(std::integral_constant<int, lb>{}, cc, qc); });
        constexpr int c = decltype(cc)::value;
        keep[c][0] = acc[lb & 1][c];  // the layer's last tile (heads: rgb / mu,sigma rows; dir layer: alpha row)
    });
}

// One layer = its stages.  On entry the layer's first stage sits in LDS buffer PAR (parked by the previous stage).
// NEXT = layer whose first stage is fetched during this layer's last stage (-1: none).
template <int L, int KIND, int NEXT, int PAR, int NCONV, bool HN_AGPR>
__device__ __forceinline__ void layer(const char *__restrict__ &wp, char *lds, const bf16x8 (&H)[NCOL][16],
                                      const bf16x8 (&X)[NCOL][8], bf16x8 (&Hn)[NCOL][16], f32x16 (&keep)[NCOL][2],
                                      bool relu, int wave, int lane) {
    constexpr int K = kK[L], NST = stages_of(L);
    static_for<NST>([&](auto stc) {
        constexpr int st = decltype(stc)::value;
        constexpr int first = st * kSPS[L];
        constexpr int nblk = kNB[L] - first < kSPS[L] ? kNB[L] - first : kSPS[L];
        char *cur = lds + ((PAR + st) & 1) * STAGE_BYTES_MAX;
        char *nxt = lds + ((PAR + st + 1) & 1) * STAGE_BYTES_MAX;
        dma_wait();       // (only the prologue's LDS-DMA of the very first stage is ever pending here)
        __syncthreads();  // every wave has parked its pieces of stage `st`; the other buffer is free again
        wp += stage_bytes(L, st);
        constexpr int nbytes = st + 1 < NST ? stage_bytes(L, st + 1) : (NEXT >= 0 ? stage_bytes(NEXT >= 0 ? NEXT : 0, 0) : 0);
        stage_compute<KIND, K, nblk, first, NCONV, nbytes, HN_AGPR>(cur, H, X, Hn, keep, relu, lane, wave, wp, nxt);
    });
}

template <bool DEPTH_HEAD>
__global__ __launch_bounds__(WG_THREADS, 1) void mlp_bf16_fwd_kernel(const unsigned short *__restrict__ feat,
                                                                     const char *__restrict__ packed,
                                                                     float *__restrict__ raw, long M) {
    __shared__ __attribute__((aligned(16))) char lds[2 * STAGE_BYTES_MAX];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, h = lane >> 5;
    long m[NCOL];
    const unsigned short *frow[NCOL];
#pragma unroll
    for (int c = 0; c < NCOL; ++c) {
        m[c] = (long)blockIdx.x * WG_SAMPLES + wave * (NCOL * 32) + c * 32 + j;
        frow[c] = feat + (size_t)(m[c] < M ? m[c] : M - 1) * DDNERF_FEAT_LD;
    }
    bf16x8 HA[NCOL][16], HB[NCOL][16], X[NCOL][8];
    f32x16 keep[NCOL][2];
    const char *wp = packed;
    dma_stage(wp, lds, stage_bytes(0, 0), wave, lane);
    // features (bf16, k-order): lane half h takes the second 16 bytes of every 16-column group
    auto load_x = [&](auto g0c, auto g1c) {  // feature groups [g0, g1) of both column blocks (re-fetched, not held)
        constexpr int g0 = decltype(g0c)::value, g1 = decltype(g1c)::value;
#pragma unroll
        for (int c = 0; c < NCOL; ++c)
#pragma unroll
            for (int g = g0; g < g1; ++g) X[c][g] = *(const bf16x8 *)(frow[c] + 16 * g + 8 * h);
    };
    load_x(std::integral_constant<int, 0>{}, std::integral_constant<int, 6>{});  // xyz: dead again after layer 0

    // parity of the LDS buffer holding a layer's first stage: L0 has 1 stage (buffer 0), every later layer has an
    // even number of stages and starts in buffer 1
    layer<0, 0, 1, 0, 8, true>(wp, lds, HA, X, HA, keep, true, wave, lane);    // 96 -> 256            (H unused: KIND 0)
    layer<1, 1, 2, 1, 8, false>(wp, lds, HA, X, HB, keep, true, wave, lane);
    layer<2, 1, 3, 1, 8, true>(wp, lds, HB, X, HA, keep, true, wave, lane);
    layer<3, 1, 4, 1, 8, false>(wp, lds, HA, X, HB, keep, true, wave, lane);
    layer<4, 1, 5, 1, 8, true>(wp, lds, HB, X, HA, keep, true, wave, lane);
    load_x(std::integral_constant<int, 0>{}, std::integral_constant<int, 6>{});
    layer<5, 2, 6, 1, 8, false>(wp, lds, HA, X, HB, keep, true, wave, lane);    // cat(xyz, h) 352 -> 256, 4 stages
    layer<6, 1, 7, 1, 8, true>(wp, lds, HB, X, HA, keep, true, wave, lane);
    layer<7, 1, 8, 1, 8, false>(wp, lds, HA, X, HB, keep, true, wave, lane);
    layer<8, 1, 9, 1, 8, true>(wp, lds, HB, X, HA, keep, false, wave, lane);   // fc_feat: no activation
    load_x(std::integral_constant<int, 6>{}, std::integral_constant<int, 8>{});  // view-dir columns
    layer<9, 3, 10, 1, 4, false>(wp, lds, HA, X, HB, keep, true, wave, lane);   // dir layer (128, ReLU) + alpha row
    float alpha[NCOL];
#pragma unroll
    for (int c = 0; c < NCOL; ++c) alpha[c] = keep[c][0][0];             // row 128 = block 4, register 0, lane half 0
    layer<10, 4, -1, 1, 0, true>(wp, lds, HB, X, HA, keep, false, wave, lane);  // heads

#pragma unroll
    for (int c = 0; c < NCOL; ++c) {
        const f32x16 &o = keep[c][0];
        if (m[c] < M) {
            if (DEPTH_HEAD) {
                float *op = raw + (size_t)m[c] * 6;
                if (h == 0) {
                    *(float2 *)(op) = make_float2(o[0], o[1]);
                    *(float2 *)(op + 2) = make_float2(o[2], alpha[c]);
                } else {
                    *(float2 *)(op + 4) = make_float2(o[0], o[1]);  // rows 4, 5 = raw mu, raw sigma
                }
            } else if (h == 0) {
                *(f32x4 *)(raw + (size_t)m[c] * 4) = f32x4{o[0], o[1], o[2], alpha[c]};
            }
        }
    }
}

DDN_EXPORT int ddnerf_mlp_bf16v1_forward(const void *feat, const void *packed, int depth_head, float *raw, long M,
                                       ddnerf_stream_t stream) {
    DDN_REQUIRE(feat && packed && raw, DDNERF_E_ARG);
    DDN_REQUIRE(M > 0, DDNERF_E_ARG);
    DDN_REQUIRE(ddn_aligned(feat, 16) && ddn_aligned(packed, 16) && ddn_aligned(raw, 16), DDNERF_E_ALIGN);
    dim3 grid((unsigned)((M + WG_SAMPLES - 1) / WG_SAMPLES));
    if (depth_head)
        hipLaunchKernelGGL(mlp_bf16_fwd_kernel<true>, grid, dim3(WG_THREADS), 0, (hipStream_t)stream,
                           (const unsigned short *)feat, (const char *)packed, raw, M);
    else
        hipLaunchKernelGGL(mlp_bf16_fwd_kernel<false>, grid, dim3(WG_THREADS), 0, (hipStream_t)stream,
                           (const unsigned short *)feat, (const char *)packed, raw, M);
    return ddn_launch_status();
}
