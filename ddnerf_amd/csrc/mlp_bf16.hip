// K2 (bf16): the whole 8x256 MLP (models/base_architectures.py:40-61, 103-126) as ONE persistent kernel on the bf16
// matrix cores (v_mfma_f32_16x16x32_bf16, fp32 accumulation) -- the north-star roofline kernel.
//
// Transposed formulation:  H_out^T[out, sample] = W[out, in] * H_in^T[in, sample]
//   A operand = 16 out-rows x 32 in-features of W, read from LDS (one ds_read_b128 per lane);
//   B operand = 32 in-features x 16 samples of the previous layer's output.  A 16x16 fp32 accumulator tile has the
//               sample on the lane (l & 15) and rows 4(l>>4)..+3 in its four registers; the tiles of row blocks 2t and
//               2t+1, converted pairwise to bf16, are the B fragment of k-step t with NO lane movement -- in a permuted
//               k order (element e of lane group g = feature 32t + 16(e>>2) + 4g + (e&3)).  W is packed in the same k
//               order, and so are the bf16 feature rows the encode kernel writes ("k-order", include/ddnerf_hip.h).
// Why the 16x16x32 shape: the chip is power-limited on bf16 MFMA loops and holds a 12-15 % higher clock on this shape
// than on 32x32x16 at equal cycles per FLOP (scratch/mfma_shapes).
//
// A wave owns 64 samples (four 16-sample column blocks): every A fragment feeds four MFMAs.  4 waves (one per SIMD,
// 512-register file) = 256 samples per tile share every LDS-staged weight byte.  Activations never leave registers:
// the two activation files (128 registers each) ping-pong, one homed in the accumulator half of the register file;
// the 128 encoded features of a sample are loaded ONCE per tile and held in registers too (so the kernel's HBM
// traffic is the algorithmic traffic: features in, raw out).
//
// The kernel is persistent (one workgroup per CU walks tiles blockIdx.x, +gridDim.x, ...): the next tile's features
// are fetched while the current tile computes, and the weight stream never stops.  Weights: repacked once per update
// into the exact LDS image -- 16-row slices, row stride 2K+32 bytes (conflict-free b128 fragment reads), each followed
// by its 16 fp32 biases -- grouped into 40 STAGES (<= 36 KiB, whole 1-KiB pieces) in consumption order.  LDS holds a
// ring of four stage buffers; while stage S feeds the MFMAs every wave moves its share of the second half of stage
// S+2 and the first half of stage S+3 global -> 4 VGPRs -> LDS, one 1-KiB piece at a time, spread over the MFMA
// stream.  One bare s_barrier per stage; a stage is certified (all of it parked, seen by every wave) two barriers
// before its first read, so fragment reads run straight across stage boundaries and no wait ever precedes a barrier
// (LDS operations of a wave complete in order: a park issued more than DEPTH+1 k-steps before the barrier has
// completed, because fragments read after it have already been consumed).
//
// The statement order below IS the instruction schedule: one MFMA, then at most a few "filler" instructions, fenced by
// sched_barrier(0) -- an MFMA of this shape holds the vector issue port for 8 of its 16 cycles, so fillers only hide
// when spread evenly, about two per MFMA.
#include "mlp_bf16_common.h"

// ---- the static plan ------------------------------------------------------------------------------------------
#define NLAY 11
#define NSTAGE 40
#define NBLOCK 154   // 16-row blocks per tile
#define NKSTEP 1205  // 32-deep k-steps per tile (4 MFMAs each per wave)
#define NBUF 4
#define BUF_BYTES (36 * 1024)
#ifndef BF16_DEPTH
#define BF16_DEPTH 5  // A fragments are read this many k-steps ahead of their MFMAs (NKSTEP % DEPTH == 0)
#endif
#ifndef BF16_PFD
#define BF16_PFD 4  // weight pieces in flight per wave (loaded, not yet parked)
#endif
#define WG_THREADS 256
#define WG_WAVES 4
#define NCB 4  // 16-sample column blocks per wave
#define TILE_SAMPLES (WG_WAVES * NCB * 16)

// 11 packed layers: K (in) / 16-row out blocks.  Layer 5 = skip layer with its columns ordered [h 256 | xyz 96];
// layer 9 = dir layer [h 256 | dir 27 + 5 zero] (128 rows) + the alpha row as row 128 of a ninth block; layer 10 = the
// heads in ONE block (rows 0-2 rgb, rows 4-5 raw mu / raw sigma).
static constexpr int kK[NLAY] = {96, 256, 256, 256, 256, 352, 256, 256, 256, 288, 128};
static constexpr int kNBLK[NLAY] = {16, 16, 16, 16, 16, 16, 16, 16, 16, 9, 1};
// stage -> (layer, first block, blocks); the last stage also carries the heads block
static constexpr int kStage[NSTAGE][3] = {
    {0, 0, 6}, {0, 6, 5}, {0, 11, 5},
    {1, 0, 4}, {1, 4, 4}, {1, 8, 4}, {1, 12, 4}, {2, 0, 4}, {2, 4, 4}, {2, 8, 4}, {2, 12, 4},
    {3, 0, 4}, {3, 4, 4}, {3, 8, 4}, {3, 12, 4}, {4, 0, 4}, {4, 4, 4}, {4, 8, 4}, {4, 12, 4},
    {5, 0, 3}, {5, 3, 3}, {5, 6, 3}, {5, 9, 3}, {5, 12, 2}, {5, 14, 2},
    {6, 0, 4}, {6, 4, 4}, {6, 8, 4}, {6, 12, 4}, {7, 0, 4}, {7, 4, 4}, {7, 8, 4}, {7, 12, 4},
    {8, 0, 4}, {8, 4, 4}, {8, 8, 4}, {8, 12, 4},
    {9, 0, 3}, {9, 3, 3}, {9, 6, 3}};

__host__ __device__ constexpr int rowb(int K) { return 2 * K + 32; }
__host__ __device__ constexpr int slice_bytes(int K) { return 16 * rowb(K) + 64; }  // 16 rows + 16 fp32 biases
__host__ __device__ constexpr int round_kib(int b) { return (b + 1023) / 1024 * 1024; }

struct Plan {
    int b_layer[NBLOCK], b_idx[NBLOCK], b_stage[NBLOCK], b_off[NBLOCK], b_k0[NBLOCK + 1];
    int s_bytes[NSTAGE], s_goff[NSTAGE + 1], s_k0[NSTAGE + 1], s_npieces[NSTAGE], s_npw[NSTAGE], s_p0[NSTAGE + 1];
    int npw, total_bytes;
};
constexpr Plan make_plan() {
    Plan p{};
    int gb = 0, k = 0, goff = 0, p0 = 0;
    for (int s = 0; s < NSTAGE; ++s) {
        p.s_k0[s] = k;
        p.s_goff[s] = goff;
        p.s_p0[s] = p0;
        int off = 0;
        const int nseg = s == NSTAGE - 1 ? 2 : 1;
        for (int seg = 0; seg < nseg; ++seg) {
            const int l = seg == 0 ? kStage[s][0] : 10, first = seg == 0 ? kStage[s][1] : 0, n = seg == 0 ? kStage[s][2] : 1;
            for (int b = 0; b < n; ++b) {
                p.b_layer[gb] = l;
                p.b_idx[gb] = first + b;
                p.b_stage[gb] = s;
                p.b_off[gb] = off;
                p.b_k0[gb] = k;
                off += slice_bytes(kK[l]);
                k += kK[l] / 32;
                ++gb;
            }
        }
        p.s_bytes[s] = (off + WG_WAVES * 1024 - 1) / (WG_WAVES * 1024) * (WG_WAVES * 1024);  // whole rounds of 4 pieces: no ragged wave
        p.s_npieces[s] = p.s_bytes[s] / 1024;
        p.s_npw[s] = p.s_npieces[s] / WG_WAVES;
        goff += p.s_bytes[s];
        p0 += p.s_npw[s];
    }
    p.b_k0[NBLOCK] = k;
    p.s_k0[NSTAGE] = k;
    p.s_goff[NSTAGE] = goff;
    p.s_p0[NSTAGE] = p0;
    p.npw = p0;
    p.total_bytes = goff;
    return p;
}
static constexpr Plan kPlan = make_plan();
static_assert(kPlan.b_k0[NBLOCK] == NKSTEP && NKSTEP % BF16_DEPTH == 0, "fragment ring must close over a tile");
constexpr bool plan_fits() {
    for (int s = 0; s < NSTAGE; ++s)
        if (kPlan.s_bytes[s] > BUF_BYTES) return false;
    return true;
}
static_assert(plan_fits(), "a stage exceeds its LDS buffer");
static_assert(NSTAGE % NBUF == 0, "stage -> buffer map must be the same for every tile");

// per-wave piece stream: NPWP pieces per tile (the real ones in stage order, then dummies up to a multiple of PFD so
// that the pf[] slot of a piece is the same in every tile)
#define NPWP ((kPlan.npw + BF16_PFD - 1) / BF16_PFD * BF16_PFD)
__host__ __device__ constexpr int wrapi(int i, int n) { return ((i % n) + n) % n; }
// first piece of the second half of stage s
__host__ __device__ constexpr int mid_piece(int s) { return kPlan.s_p0[wrapi(s, NSTAGE)] + kPlan.s_npw[wrapi(s, NSTAGE)] / 2; }
// pieces parked while stage s computes: [park_lo(s), park_lo(s) + park_n(s)) (cyclic over NPWP)
__host__ __device__ constexpr int park_lo(int s) { return mid_piece(s + 2); }
__host__ __device__ constexpr int park_n(int s) { return wrapi(mid_piece(s + 3) - mid_piece(s + 2), NPWP); }
// k-steps of stage s whose DMA gap may be used: a park must be more than DEPTH+1 k-steps ahead of the stage's barrier
__host__ __device__ constexpr int usable_n(int s) {
    int n = kPlan.s_k0[s + 1] - kPlan.s_k0[s] - (BF16_DEPTH + 2);
    return n < 1 ? 1 : n;
}
// stage of a (cyclic) piece index
__host__ __device__ constexpr int piece_stage(int q) {
    q = wrapi(q, NPWP);
    if (q >= kPlan.npw) return -1;  // dummy
    int s = 0;
    while (q >= kPlan.s_p0[s + 1]) ++s;
    return s;
}
constexpr bool windows_ok() {  // every piece of stage T is parked during stage T-3 or T-2
    int total = 0;
    for (int s = 0; s < NSTAGE; ++s) {
        total += park_n(s);
        for (int i = 0; i < park_n(s); ++i) {
            const int T = piece_stage(park_lo(s) + i);
            if (T >= 0 && wrapi(T - s, NSTAGE) != 2 && wrapi(T - s, NSTAGE) != 3) return false;
        }
    }
    return total == NPWP;
}
static_assert(windows_ok(), "piece stream violates the certification window");

struct Src {
    int w_src[13], b_src[13];
};
static Src make_src(int depth_head) {
    Src p;
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

DDN_EXPORT size_t ddnerf_mlp_bf16_packed_bytes(int depth_head) {
    (void)depth_head;
    return (size_t)kPlan.total_bytes;
}

// packed layer l, out row o, LOGICAL column c (layer 5: [h | xyz], layer 9: [h | dir]) -> fp32 parameter
__device__ __forceinline__ float srcw(const float *__restrict__ P, const Src &pl, int l, int o, int c) {
    if (l == 5) return P[pl.w_src[5] + o * 352 + (c < 256 ? 96 + c : c - 256)];  // reference input: cat(xyz, h)
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
__device__ __forceinline__ float srcb(const float *__restrict__ P, const Src &pl, int l, int o) {
    if (l <= 8) return P[pl.b_src[l] + o];
    if (l == 9) return o < 128 ? P[pl.b_src[10] + o] : (o == 128 ? P[pl.b_src[9]] : 0.0f);
    if (o < 3) return P[pl.b_src[11] + o];
    if ((o == 4 || o == 5) && pl.b_src[12] >= 0) return P[pl.b_src[12] + (o - 4)];
    return 0.0f;
}

__global__ void mlp_bf16_pack_kernel(const float *__restrict__ P, Src pl, unsigned short *__restrict__ packed) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;  // one thread per 16-bit word of the packed buffer
    if (idx >= kPlan.total_bytes / 2) return;
    const int byte = idx * 2;
    int s = 0;
    while (byte >= kPlan.s_goff[s + 1]) ++s;
    const int rel = byte - kPlan.s_goff[s];
    unsigned short w = 0;
    for (int gb = 0; gb < NBLOCK; ++gb) {
        if (kPlan.b_stage[gb] != s) continue;
        const int l = kPlan.b_layer[gb], K = kK[l], r2 = rel - kPlan.b_off[gb];
        if (r2 < 0 || r2 >= slice_bytes(K)) continue;
        const int o0 = 16 * kPlan.b_idx[gb];
        if (r2 < 16 * rowb(K)) {
            const int row = r2 / rowb(K), col = (r2 % rowb(K)) / 2;
            const float v = col < K ? srcw(P, pl, l, o0 + row, korder32(col)) : 0.0f;
            const __bf16 b = (__bf16)v;
            w = __builtin_bit_cast(unsigned short, b);
        } else {  // fp32 bias of row (r2 - 16*rowb)/4, written as two 16-bit halves
            const int bi = (r2 - 16 * rowb(K)) / 4, half = ((r2 - 16 * rowb(K)) % 4) / 2;
            const unsigned u = __builtin_bit_cast(unsigned, srcb(P, pl, l, o0 + bi));
            w = (unsigned short)(half ? (u >> 16) : (u & 0xffffu));
        }
        break;
    }
    packed[idx] = w;
}

DDN_EXPORT int ddnerf_mlp_bf16_pack(const float *params, int depth_head, void *packed, ddnerf_stream_t stream) {
    DDN_REQUIRE(params && packed, DDNERF_E_ARG);
    DDN_REQUIRE(ddn_aligned(packed, 16), DDNERF_E_ALIGN);
    const Src pl = make_src(depth_head);
    const int threads = kPlan.total_bytes / 2;
    hipLaunchKernelGGL(mlp_bf16_pack_kernel, dim3((threads + 255) / 256), dim3(256), 0, (hipStream_t)stream, params, pl,
                       (unsigned short *)packed);
    return ddn_launch_status();
}

// ---- fused forward ----------------------------------------------------------------------------------------
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef short s16x2 __attribute__((ext_vector_type(2)));
template <int V>
using ic = std::integral_constant<int, V>;

__device__ __forceinline__ unsigned cvt_bf16(float a, float b) {  // one v_cvt_pk_bf16_f32
    const f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ unsigned relu_bf16(unsigned w) {  // ReLU on the bf16 bit patterns: one v_pk_max_i16
    const s16x2 z = {0, 0};
    return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2, w), z));
}

// (The tile body must stay ONE basic block: the re-pack instructions are pure and their results are needed only a layer
// later, so with a branch in the stream machine sinking -- which ignores sched_barrier -- moves them out of the MFMA gaps
// they were written into and issues them in one burst behind the branch.)
__device__ __forceinline__ unsigned to_agpr_here(unsigned v) {
    unsigned a;
    asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(a) : "v"(v));
    return a;
}

// everything a wave keeps in registers
struct Regs {
    bf16x8 HA[NCB][8], HB[NCB][8];  // the two activation files (HA homed in the accumulator half)
    bf16x8 X[NCB][4];               // the tile's 128 encoded features (k-steps 0..2 xyz, 3 view dirs), accumulator half
    bf16x8 Xt[NCB];                 // the next tile's features on their way in
    f32x4 acc[2][NCB], biasv[2];
    bf16x8 ring[BF16_DEPTH];
    f32x4 pf[BF16_PFD];
    unsigned rp[NCB][2];            // re-pack words between their conversion and their home
    // LDS addresses of this lane's fragment / bias slot in the stage buffers (stage parity; the heads block apart):
    // every fragment read is base + a 16-bit immediate
    unsigned abase[2], bbase[2], abase_h, bbase_h;
    unsigned pbase[2];  // this lane's slot in the buffers being filled (parity of the stage the piece belongs to)
};

template <int L>
__device__ __forceinline__ constexpr bool out_in_a() { return (L & 1) == 0; }  // layer L writes HA (even) or HB (odd)

// B operand of k-step ks of layer L, column block c
template <int L, int KS>
__device__ __forceinline__ const bf16x8 &bsrc(const Regs &r, int c) {
    if constexpr (L == 0) return r.X[c][KS];
    else if constexpr (L == 5 && KS >= 8) return r.X[c][KS - 8];
    else if constexpr (L == 9 && KS >= 8) return r.X[c][3];
    else if constexpr ((L & 1) == 1) return r.HA[c][KS];
    else return r.HB[c][KS];
}

// One matrix instruction, written as asm so that the accumulator stays IN PLACE (vdst = srcC).  hipcc's own selection --
// once a kernel uses the accumulator half of the register file at all -- is the untied form with an early-clobber
// destination: the tile hops to fresh registers on every instruction, and each register it leaves behind costs wait
// states when the next load or VALU result lands in it.  Nothing is padded inside asm (cdna_hip_programming.md 5.7), so
// the schedule keeps every dependent pair far apart by construction: a tile is read by the VALU >= 3 MFMAs after its last
// write, re-pack results are read by MFMAs at least a k-step later, same-tile MFMAs are 4 apart.
// B_IN_A: the B operand lives in the accumulator half (HA, X) / in arch VGPRs (HB).  FIRST: start from the bias tile.
#define BF16_STR2(x) #x
#define BF16_STR(x) BF16_STR2(x)
#ifdef BF16_MFMA_PAD  // debug build: wait states in front of every MFMA (or only the first of a block: BF16_PAD_FIRST_ONLY)
#define MFMA_OP_PAD "s_nop " BF16_STR(BF16_MFMA_PAD) "\n\tv_mfma_f32_16x16x32_bf16"
#else
#define MFMA_OP_PAD "v_mfma_f32_16x16x32_bf16"
#endif
#ifdef BF16_PAD_FIRST_ONLY
#define MFMA_OP "v_mfma_f32_16x16x32_bf16"
#else
#define MFMA_OP MFMA_OP_PAD
#endif
template <bool B_IN_A, bool FIRST>
__device__ __forceinline__ void mfma(f32x4 &acc, const bf16x8 &a, const bf16x8 &b, const f32x4 &bias) {
    if constexpr (FIRST) {
        if constexpr (B_IN_A) asm volatile(MFMA_OP_PAD " %0, %1, %2, %3" : "=&v"(acc) : "v"(a), "a"(b), "v"(bias));
        else asm volatile(MFMA_OP_PAD " %0, %1, %2, %3" : "=&v"(acc) : "v"(a), "v"(b), "v"(bias));
    } else {
        if constexpr (B_IN_A) asm volatile(MFMA_OP " %0, %1, %2, %0" : "+v"(acc) : "v"(a), "a"(b));
        else asm volatile(MFMA_OP " %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
    }
}

// LDS byte address (relative to the ring base) of the A fragment of global k-step n (cyclic over tiles)
__host__ __device__ constexpr int kstep_block(int n) {
    int gb = 0;
    while (n >= kPlan.b_k0[gb + 1]) ++gb;
    return gb;
}
__host__ __device__ constexpr int block_lds(int gb) { return (kPlan.b_stage[gb] % NBUF) * BUF_BYTES + kPlan.b_off[gb]; }

__device__ __forceinline__ const char *lds_ptr(unsigned a) { return (const char *)(const __attribute__((address_space(3))) char *)(size_t)a; }
__host__ __device__ constexpr int stage_k(int s) { return kK[kStage[s % NSTAGE][0]]; }
// bases of stage S (computed while stage S-1 runs): one v_mad each; `la16`/`lg16` = (lane & 15) and 16 * (lane >> 4)
// behind an opaque copy, so that the five la * rowb(K) products are not kept in registers for the whole kernel
template <int S>
__device__ __forceinline__ void stage_bases(Regs &r, unsigned lds0, int wave, int lane) {
    constexpr int s = S % NSTAGE;
    unsigned la = lane & 15, lg16 = 16 * (lane >> 4), lpark = wave * 1024 + lane * 16;
    asm volatile("" : "+v"(la), "+v"(lg16), "+v"(lpark));
    r.abase[S & 1] = lds0 + (s % NBUF) * BUF_BYTES + la * rowb(stage_k(s)) + lg16;
    r.bbase[S & 1] = lds0 + (s % NBUF) * BUF_BYTES + 16 * rowb(stage_k(s)) + lg16;
    // the first pieces of stage S+2 are parked while stage S-1 runs
    r.pbase[S & 1] = lds0 + ((s + 2) % NBUF) * BUF_BYTES + lpark;
    if constexpr (s == NSTAGE - 1) {  // the heads block rides in the last stage with its own row stride
        r.abase_h = lds0 + (s % NBUF) * BUF_BYTES + kPlan.b_off[NBLOCK - 1] + la * rowb(kK[10]) + lg16;
        r.bbase_h = lds0 + (s % NBUF) * BUF_BYTES + kPlan.b_off[NBLOCK - 1] + 16 * rowb(kK[10]) + lg16;
    }
}
template <int N>
__device__ __forceinline__ void read_a(Regs &r) {
    constexpr int n = N % NKSTEP, gb = kstep_block(n), ks = n - kPlan.b_k0[gb], S = kPlan.b_stage[gb];
    if constexpr (kPlan.b_layer[gb] == 10) r.ring[N % BF16_DEPTH] = *(const bf16x8 *)(lds_ptr(r.abase_h) + 64 * ks);
    else r.ring[N % BF16_DEPTH] = *(const bf16x8 *)(lds_ptr(r.abase[S & 1]) + kPlan.b_off[gb] + 64 * ks);
}
template <int GB>
__device__ __forceinline__ void read_bias(Regs &r) {
    constexpr int gb = GB % NBLOCK, S = kPlan.b_stage[gb];
    if constexpr (kPlan.b_layer[gb] == 10) r.biasv[GB & 1] = *(const f32x4 *)(lds_ptr(r.bbase_h));
    else r.biasv[GB & 1] = *(const f32x4 *)(lds_ptr(r.bbase[S & 1]) + kPlan.b_off[gb]);
}

// piece q (cyclic per-wave index) of the weight stream: wave w moves KiB (4i + w) of its stage.  Loads are buffer loads
// (descriptor of the packed image in SGPRs, the lane's 16-byte slot as the VGPR offset, the piece as an SGPR offset: one
// s_add per piece); no branch anywhere in a tile, so the tile body is ONE basic block and stays in the written order.
struct Dma {
    __amdgpu_buffer_rsrc_t rsrc;
    int swave;       // wave * 1024 (SGPR)
    unsigned lane16; // lane * 16
};
template <int Q, bool PARK>
__device__ __forceinline__ void dma_item(Regs &r, const Dma &d) {
    constexpr int q = wrapi(Q, NPWP), T = piece_stage(q);
    if constexpr (T >= 0) {
        constexpr int i = q - kPlan.s_p0[T];
#if defined(BF16_NO_PARK)  // ablation builds (wrong results): loads only / parks only
        if constexpr (PARK) asm volatile("" ::"v"(r.pf[q % BF16_PFD]));
#else
        if constexpr (PARK) *(f32x4 *)(lds_ptr(r.pbase[T & 1]) + WG_WAVES * i * 1024) = r.pf[q % BF16_PFD];
#endif
#if defined(BF16_NO_LOAD)
        else asm volatile("" : "+v"(r.pf[q % BF16_PFD]));
#else
        else
            r.pf[q % BF16_PFD] = __builtin_bit_cast(
                f32x4, __builtin_amdgcn_raw_buffer_load_b128(d.rsrc, d.lane16, d.swave + (kPlan.s_goff[T] + WG_WAVES * i * 1024), 0));
#endif
    }
}

// the DMA items hosted by the gap of k-step n: item j of stage S = park of piece park_lo+j/2 (j even) or load of piece
// park_lo+j/2+PFD (j odd), spread evenly over the stage's usable k-steps
template <int N>
__device__ __forceinline__ void dma_gap(Regs &r, const Dma &d) {
    constexpr int S = kPlan.b_stage[kstep_block(N)], u = N - kPlan.s_k0[S], U = usable_n(S), ni = 2 * park_n(S);
#ifdef BF16_NO_DMA  // ablation build (wrong results): no weight staging at all
    if constexpr (false) {
#else
    if constexpr (u < U) {
#endif
        constexpr int jlo = (u * ni + U - 1) / U, jhi = ((u + 1) * ni + U - 1) / U;
        static_for<(jhi > jlo ? jhi - jlo : 0)>([&](auto jc) {
            constexpr int j = jlo + decltype(jc)::value;
            if constexpr (j % 2 == 0) dma_item<park_lo(S) + j / 2, true>(r, d);
            else dma_item<park_lo(S) + j / 2 + BF16_PFD, false>(r, d);
        });
    }
}

// re-pack of block PB (global index) = 3 pair-steps per column block: convert, ReLU, home
template <int PB, int C, int STEP>
__device__ __forceinline__ void repack_step(Regs &r) {
    constexpr int L = kPlan.b_layer[PB], b = kPlan.b_idx[PB], par = PB & 1;
#ifdef BF16_NO_REPACK  // ablation build (wrong results): activations are never written back (the tiles stay live)
    constexpr bool conv = false;
    if constexpr (STEP == 0) asm volatile("" ::"v"(r.acc[par][C]));
#else
    constexpr bool conv = L < 9 || (L == 9 && b < 8);
#endif
    if constexpr (conv) {
        if constexpr (STEP == 0) {
            r.rp[C][0] = cvt_bf16(r.acc[par][C][0], r.acc[par][C][1]);
            r.rp[C][1] = cvt_bf16(r.acc[par][C][2], r.acc[par][C][3]);
        } else if constexpr (STEP == 1) {
            if constexpr (L != 8) {  // fc_feat has no activation
                r.rp[C][0] = relu_bf16(r.rp[C][0]);
                r.rp[C][1] = relu_bf16(r.rp[C][1]);
            }
        } else {
            if constexpr (out_in_a<L>()) {
                u32x4 w = __builtin_bit_cast(u32x4, r.HA[C][b / 2]);
                unsigned a0, a1;  // both words in ONE asm statement: hipcc pads a wait state between two asm statements
                asm volatile("v_accvgpr_write_b32 %0, %2\n\tv_accvgpr_write_b32 %1, %3" : "=a"(a0), "=a"(a1) : "v"(r.rp[C][0]), "v"(r.rp[C][1]));
                w[2 * (b & 1)] = a0;
                w[2 * (b & 1) + 1] = a1;
                r.HA[C][b / 2] = __builtin_bit_cast(bf16x8, w);
                // the fragment is complete: from here on it is ONE 128-bit value born in the accumulator half, which is
                // what the MFMA asm asks for -- otherwise LLVM assembles some fragments in arch VGPRs and copies them
                // over right in front of the MFMA, where nothing pads the VALU-write -> MFMA-read wait states
                if constexpr ((b & 1) == 1) asm volatile("" : "+a"(r.HA[C][b / 2]));
            } else {
                u32x4 w = __builtin_bit_cast(u32x4, r.HB[C][b / 2]);
                w[2 * (b & 1)] = r.rp[C][0];
                w[2 * (b & 1) + 1] = r.rp[C][1];
                r.HB[C][b / 2] = __builtin_bit_cast(bf16x8, w);
                if constexpr ((b & 1) == 1) asm volatile("" : "+v"(r.HB[C][b / 2]));
            }
        }
    }
}

// Feature rows are read exactly once: a non-temporal load keeps them from displacing the weight image in the XCD's L2
#ifdef BF16_FEAT_PLAIN
#define FEAT_LOAD(p) (*(p))
#else
#define FEAT_LOAD(p) __builtin_nontemporal_load(p)
#endif

// feature events: the NEXT tile's features travel global -> Xt (arch VGPRs) -> X (accumulator half), one 32-column
// group at a time, at blocks where the current tile no longer needs that group
__device__ __forceinline__ bf16x8 to_agpr8(bf16x8 v) {
    u32x4 w = __builtin_bit_cast(u32x4, v);
    w[0] = to_agpr_here(w[0]);
    w[1] = to_agpr_here(w[1]);
    w[2] = to_agpr_here(w[2]);
    w[3] = to_agpr_here(w[3]);
    bf16x8 o = __builtin_bit_cast(bf16x8, w);
    asm volatile("" : "+a"(o));
    return o;
}
// event of block gb: 0 none, 1+q load group q, 5+q convert group q
__host__ __device__ constexpr int x_event(int gb) {
    const int l = kPlan.b_layer[gb], b = kPlan.b_idx[gb];
    if (l == 6) {
        if (b == 0 || b == 4 || b == 8) return 1 + b / 4;
        if (b == 3 || b == 7 || b == 11) return 5 + b / 4;
    }
    if (l == 10) return 1 + 3;
    if (l == 0 && b == 12) return 5 + 3;
    return 0;
}
template <int GB, int HALF>
__device__ __forceinline__ void x_gap(Regs &r, const char *__restrict__ feat, const unsigned (&xoff)[NCB], int lane) {
    constexpr int ev = x_event(GB);
    if constexpr (ev >= 1 && ev <= 4) {
#pragma unroll
        for (int c = 2 * HALF; c < 2 * HALF + 2; ++c)
            r.Xt[c] = FEAT_LOAD((const bf16x8 *)(feat + (size_t)xoff[c] + 64 * (ev - 1) + 16 * (lane >> 4)));
    } else if constexpr (ev >= 5) {
#pragma unroll
        for (int c = 2 * HALF; c < 2 * HALF + 2; ++c) r.X[c][ev - 5] = to_agpr8(r.Xt[c]);
    }
}

// One 16-row block: NKS k-steps x NCB MFMAs; the fillers of each MFMA gap are listed right behind it.
template <int GB>
__device__ __forceinline__ void block_compute(Regs &r, const Dma &d, const char *__restrict__ feat,
                                              const unsigned (&xoff)[NCB], unsigned lds0, int wave, int lane) {
    constexpr int L = kPlan.b_layer[GB], NKS = kK[L] / 32, K0 = kPlan.b_k0[GB], par = GB & 1;
    constexpr int KMAX = NKS <= 4 ? NKS : (NKS - 1 < 7 ? NKS - 1 : 7);  // re-pack of the previous block ends before k-step 7
    constexpr int NSLOT = 2 * KMAX, NPAIR = 3 * NCB;                    // (ks, c in {1,3}) slots; pair-steps to place
    if constexpr (GB == 0 || kPlan.b_stage[GB] != kPlan.b_stage[GB > 0 ? GB - 1 : 0])  // first block of a stage
        stage_bases<kPlan.b_stage[GB] + 1>(r, lds0, wave, lane);
    static_for<NKS>([&](auto ksc) {
        constexpr int ks = decltype(ksc)::value, n = K0 + ks;
        static_for<NCB>([&](auto cc) {
            constexpr int c = decltype(cc)::value;
            mfma<(L == 0 || (L == 5 && ks >= 8) || (L == 9 && ks >= 8) || (L & 1) == 1), ks == 0>(
                r.acc[par][c], r.ring[n % BF16_DEPTH], bsrc<L, ks>(r, c), r.biasv[par]);
            // the bias tile stays allocated until the MFMAs that read it as srcC are well under way (a dead register is
            // re-used at once, and a VALU write within 3 wait states of such an MFMA corrupts its srcC read)
            if constexpr (ks == 1 && c == 0) asm volatile("" ::"v"(r.biasv[par]));
            if constexpr (c == 0) {
                // the slot of the PREVIOUS k-step's fragment is refilled (one MFMA behind its last reader: a load into a
                // register that the MFMA just issued still reads costs wait states)
                read_a<n - 1 + BF16_DEPTH>(r);
                if constexpr (ks == (NKS >= 3 ? NKS - 3 : 0)) read_bias<GB + 1>(r);
            }
            if constexpr (c == 1) dma_gap<n>(r, d);
            if constexpr ((c == 2 || c == 3) && GB > 0 && ks < KMAX) {  // re-pack pair-steps of the previous block
                constexpr int slot = 2 * ks + (c - 2);
                constexpr int tlo = (slot * NPAIR + NSLOT - 1) / NSLOT, thi = ((slot + 1) * NPAIR + NSLOT - 1) / NSLOT;
                static_for<(thi > tlo ? thi - tlo : 0)>([&](auto tc) {
                    constexpr int t = tlo + decltype(tc)::value;
                    repack_step<(GB > 0 ? GB - 1 : 0), t / 3, t % 3>(r);
                });
            }
            if constexpr ((c == 2 || c == 3) && ks == NKS - 1 && ks >= KMAX) x_gap<GB, c - 2>(r, feat, xoff, lane);
            __builtin_amdgcn_sched_barrier(0);
        });
    });
    if constexpr (x_event(GB) != 0 && NKS <= 4) {  // short blocks have no free gap: the event trails the block
        x_gap<GB, 0>(r, feat, xoff, lane);
        x_gap<GB, 1>(r, feat, xoff, lane);
        __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (kPlan.b_stage[GB] != kPlan.b_stage[(GB + 1) % NBLOCK]) {  // stage boundary: bare barrier, no wait
        asm volatile("" ::: "memory");
#ifndef BF16_NO_BARRIER  // ablation build (races): no stage barriers
        __builtin_amdgcn_s_barrier();
#endif
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    }
}

#ifdef BF16_STAMP
// Diagnostic build only (scratch/ab/build_bf16_variants.sh stamp; never in libddnerf_hip.so): the in-kernel clock,
// MI355X_MICROARCH.md "DVFS give-back" item 6.  Every workgroup stamps s_memtime / s_memrealtime around its tile loop into a
// buffer of its own (set through ddnerf_debug_set_stamps); no output value depends on a stamp.
__device__ unsigned long long *g_bf16_stamps;
DDN_EXPORT int ddnerf_debug_set_stamps(void *p) {
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_bf16_stamps), &p, sizeof(p));
}
#endif

template <bool DEPTH_HEAD>
__global__ __launch_bounds__(WG_THREADS, 1) void mlp_bf16_fwd_kernel(const char *__restrict__ feat,
                                                                     const char *__restrict__ packed,
                                                                     float *__restrict__ raw, long M, long ntiles) {
    __shared__ __attribute__((aligned(16))) char lds[NBUF * BUF_BYTES];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 15, g = lane >> 4;
    Regs r;
    // byte offset of this lane's feature row of column block c in tile t (rows past M clamp to the last row)
    auto row_off = [&](long t, int c) -> unsigned {
        long m = t * TILE_SAMPLES + wave * (NCB * 16) + c * 16 + j;
        return (unsigned)((m < M ? m : M - 1) * (2 * DDNERF_FEAT_LD));
    };
    long tile = blockIdx.x;
    // ---- prologue: what the steady state assumes at the start of a tile
    // (a) stages 0, 1 and the first half of stage 2 parked (LDS-DMA straight from the packed image)
    static_for<mid_piece(2)>([&](auto qc) {
        constexpr int q = decltype(qc)::value, T = piece_stage(q), i = q - kPlan.s_p0[T];
        dma_piece(packed + kPlan.s_goff[T] + (wave + WG_WAVES * i) * 1024 + lane * 16,
                  lds_addr_of(lds + (T % NBUF) * BUF_BYTES + (wave + WG_WAVES * i) * 1024));
    });
    // (b) the first tile's features
    {
        unsigned xo[NCB];
#pragma unroll
        for (int c = 0; c < NCB; ++c) xo[c] = row_off(tile, c);
        bf16x8 t[NCB][4];
#pragma unroll
        for (int c = 0; c < NCB; ++c)
#pragma unroll
            for (int q = 0; q < 4; ++q) t[c][q] = FEAT_LOAD((const bf16x8 *)(feat + (size_t)xo[c] + 64 * q + 16 * g));
#pragma unroll
        for (int c = 0; c < NCB; ++c)
#pragma unroll
            for (int q = 0; q < 4; ++q) r.X[c][q] = to_agpr8(t[c][q]);
        // the view-dir group of a tile is converted at block 12 of layer 0 (from the load behind the previous tile's heads)
#pragma unroll
        for (int c = 0; c < NCB; ++c) r.Xt[c] = t[c][3];
    }
    dma_wait();
    __syncthreads();
    // (c) the pieces in flight at a tile start, the first fragments and the first bias
    Dma d;
    d.rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)packed, 0, kPlan.total_bytes, 0x00020000);
    d.swave = wave * 1024;
    d.lane16 = lane * 16;
    static_for<BF16_PFD>([&](auto ic_) { dma_item<park_lo(0) + decltype(ic_)::value, false>(r, d); });
    stage_bases<0>(r, lds_addr_of(lds), wave, lane);
    static_for<BF16_DEPTH - 1>([&](auto nc) { read_a<decltype(nc)::value>(r); });  // (fragment DEPTH-1 follows k-step 0's first MFMA)
    read_bias<0>(r);
    __builtin_amdgcn_sched_barrier(0);

#ifdef BF16_STAMP
    const unsigned long long st_t0 = __builtin_amdgcn_s_memtime(), st_r0 = __builtin_amdgcn_s_memrealtime();
    const long st_first = tile;
#endif
    for (; tile < ntiles; tile += gridDim.x) {
        unsigned xoff[NCB];  // next tile's feature rows (clamped: the loads of a tile that does not exist are harmless)
#pragma unroll
        for (int c = 0; c < NCB; ++c) xoff[c] = row_off(tile + gridDim.x, c);
        // Per-tile opaque copies of the address bases: without them every one of the tile's ~2000 constant-offset
        // addresses is loop-invariant, gets hoisted out of the tile loop and spilled.
        unsigned lds0 = lds_addr_of(lds);
        asm volatile("" : "+s"(d.swave), "+s"(lds0));
        static_for<NBLOCK>([&](auto gbc) { block_compute<decltype(gbc)::value>(r, d, feat, xoff, lds0, wave, lane); });
        // outputs: heads block = global block 153 (rows 0-2 rgb on lane group 0, rows 4-5 mu/sigma on lane group 1),
        // alpha = row 128 of the dir layer = block 152, register 0, lane group 0
#pragma unroll
        for (int c = 0; c < NCB; ++c) {
            const long m = tile * TILE_SAMPLES + wave * (NCB * 16) + c * 16 + j;
            const f32x4 o = r.acc[(NBLOCK - 1) & 1][c];
            const float alpha = r.acc[(NBLOCK - 2) & 1][c][0];
            if (m < M) {
                if (DEPTH_HEAD) {
                    float *op = raw + (size_t)m * 6;
                    if (g == 0) {
                        *(float2 *)(op) = make_float2(o[0], o[1]);
                        *(float2 *)(op + 2) = make_float2(o[2], alpha);
                    } else if (g == 1) {
                        *(float2 *)(op + 4) = make_float2(o[0], o[1]);
                    }
                } else if (g == 0) {
                    *(f32x4 *)(raw + (size_t)m * 4) = f32x4{o[0], o[1], o[2], alpha};
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
#ifdef BF16_STAMP
    {
        const unsigned long long st_t1 = __builtin_amdgcn_s_memtime(), st_r1 = __builtin_amdgcn_s_memrealtime();
        unsigned long long *sp = g_bf16_stamps;
        if (sp && tid == 0) {
            sp[5 * blockIdx.x + 0] = st_t0;
            sp[5 * blockIdx.x + 1] = st_r0;
            sp[5 * blockIdx.x + 2] = st_t1;
            sp[5 * blockIdx.x + 3] = st_r1;
            sp[5 * blockIdx.x + 4] = (unsigned long long)((tile - st_first) / gridDim.x);
        }
    }
#endif
}

DDN_EXPORT int ddnerf_mlp_bf16_forward(const void *feat, const void *packed, int depth_head, float *raw, long M,
                                       ddnerf_stream_t stream) {
    DDN_REQUIRE(feat && packed && raw, DDNERF_E_ARG);
    DDN_REQUIRE(M > 0, DDNERF_E_ARG);
    DDN_REQUIRE(ddn_aligned(feat, 16) && ddn_aligned(packed, 16) && ddn_aligned(raw, 16), DDNERF_E_ALIGN);
    static int n_cu = 0;
    if (n_cu == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return DDNERF_E_ARG;
        n_cu = prop.multiProcessorCount;
    }
    // feature rows are addressed with 32-bit byte offsets: at most 2^24 - 256 samples per launch
    const long CHUNK = ((1L << 24) - TILE_SAMPLES);
    for (long m0 = 0; m0 < M; m0 += CHUNK) {
        const long Mc = M - m0 < CHUNK ? M - m0 : CHUNK;
        const long ntiles = (Mc + TILE_SAMPLES - 1) / TILE_SAMPLES;
        const dim3 grid((unsigned)(ntiles < n_cu ? ntiles : n_cu));
        const char *f = (const char *)feat + (size_t)m0 * (2 * DDNERF_FEAT_LD);
        float *o = raw + (size_t)m0 * (depth_head ? 6 : 4);
        if (depth_head)
            hipLaunchKernelGGL(mlp_bf16_fwd_kernel<true>, grid, dim3(WG_THREADS), 0, (hipStream_t)stream, f,
                               (const char *)packed, o, Mc, ntiles);
        else
            hipLaunchKernelGGL(mlp_bf16_fwd_kernel<false>, grid, dim3(WG_THREADS), 0, (hipStream_t)stream, f,
                               (const char *)packed, o, Mc, ntiles);
    }
    return ddn_launch_status();
}
