// K2 training kernels (fp32): the forward that also records the activations, and the fused backward-data pass.
//
// Gradient structure: the MLP needs parameter gradients only -- no gradient w.r.t. its 123 input
// features.  The work is split the MI355X way:
//   * forward_train: the forward kernel of mlp_f32.hip that additionally stores every layer's output TRANSPOSED,
//     acts[row = feature][col = sample]: in the accumulator layout a register holds one feature for 32 consecutive
//     samples per lane half, so each store instruction writes two full 128-byte row segments.
//   * backward_data: ONE fused kernel chains d(raw) back through all layers exactly like the forward chains
//     activations -- delta_l^T = (W_{l+1}^T delta_{l+1}^T) * relu'(h_l), the accumulator tile of one step is the B
//     operand of the next, deltas never leave registers between layers -- and stores every delta transposed too.
//   * weight gradients dW_l = delta_l^T (h_{l-1}^T)^T contract over the sample axis of the two stored
//     [feature][sample] matrices (both operands K-contiguous): mlp_f32_wgrad.hip (split over the samples, order-fixed
//     reduce), which also produces the bias gradients (row sums).
// Row map of `acts` and `deltas` (ld = samples rounded up to 128):
//   rows 256*l .. 256*l+255 : layers_xyz.l output (post-ReLU) / its pre-activation gradient,   l = 0..7
//   rows 2048 .. 2303       : fc_feat output (no activation)  / its gradient
//   rows 2304 .. 2431       : layers_dir.0 output (post-ReLU) / its pre-activation gradient
//   rows 2432 .. 2559       : acts: the input features (xyz 0..95, view dirs 96..122, zeros), transposed, so the first-layer /
//                             skip / dir weight gradients contract over contiguous samples too;
//                             deltas: rows 2432..2437 = d(raw) columns 0..5 transposed (head weight gradients)
#include "mlp_f32_common.h"

#define ACT_ROWS 2560
#define ROW_X 2432  // acts: the 128 input feature columns, transposed; deltas: the d(raw) tile (rows 0..5 used)
#define ROW_FEAT 2048
#define ROW_DIR 2304

// ---- packed images ------------------------------------------------------------------------------------------
// forward plan (identical to mlp_f32.hip)
#define NLAYERS 11
static constexpr int kLayerK[NLAYERS] = {96, 256, 256, 256, 256, 352, 256, 256, 256, 288, 128};
static constexpr int kLayerNB[NLAYERS] = {8, 8, 8, 8, 8, 8, 8, 8, 8, 5, 1};
// backward plan: step d -> (K = rows of the incoming delta, NB = 32-row blocks of the outgoing delta)
//   d0: heads^T          K=32  (d raw tile)            -> d(dir hidden) 128 rows
//   d1: [dir|alpha]^T    K=160 (d dir hidden + d raw)  -> d(feat)       256 rows
//   d2: fc_feat^T        K=256                         -> d(h7)
//   d3..d9: layers_xyz.{7..1}^T (layer 5: its hidden columns only)      -> d(h6) .. d(h0)
#define NBSTEPS 10
static constexpr int kBK[NBSTEPS] = {32, 160, 256, 256, 256, 256, 256, 256, 256, 256};
static constexpr int kBNB[NBSTEPS] = {4, 8, 8, 8, 8, 8, 8, 8, 8, 8};

struct PlanT {
    int step_off[NBSTEPS];  // float offset of the step's first slice
    int w_src[13];
    int total;
};

static PlanT make_plan_t(int depth_head) {
    PlanT p;
    static const int nout[13] = {256, 256, 256, 256, 256, 256, 256, 256, 256, 1, 128, 3, 2};
    static const int nin[13] = {96, 256, 256, 256, 256, 352, 256, 256, 256, 256, 283, 128, 128};
    int off = 0;
    for (int l = 0; l < 13; ++l) {
        p.w_src[l] = off;
        off += nout[l] * nin[l] + nout[l];
        if (l == 11 && !depth_head) {
            p.w_src[12] = -1;
            break;
        }
    }
    off = 0;
    for (int d = 0; d < NBSTEPS; ++d) {
        p.step_off[d] = off;
        off += kBNB[d] * slice_floats(kBK[d]);
    }
    p.total = off;
    return p;
}

#ifndef F32_REC
DDN_EXPORT size_t ddnerf_mlp_f32_packed_t_floats(int depth_head) { return (size_t)make_plan_t(depth_head).total; }
DDN_EXPORT size_t ddnerf_mlp_act_rows(void) { return ACT_ROWS; }
#endif

// A^T element of backward step d: row c (feature of the outgoing delta), column o (row of the incoming delta)
__device__ __forceinline__ float src_wt(const float *__restrict__ P, const PlanT &pl, int d, int c, int o) {
    if (d == 0) {  // heads: incoming rows = raw columns (0..2 rgb, 3 alpha [not an input of this step], 4,5 mu,sigma)
        if (o < 3) return P[pl.w_src[11] + o * 128 + c];
        if ((o == 4 || o == 5) && pl.w_src[12] >= 0) return P[pl.w_src[12] + (o - 4) * 128 + c];
        return 0.0f;
    }
    if (d == 1) {  // d(feat)[c] = sum_o W_dir[o][c] d(dir)[o]  +  W_alpha[c] d(raw)[3]
        if (o < 128) return P[pl.w_src[10] + o * 283 + c];
        if (o == 128 + 3) return P[pl.w_src[9] + c];
        return 0.0f;
    }
    if (d == 2) return P[pl.w_src[8] + o * 256 + c];  // fc_feat
    const int l = 10 - d;                              // d3 -> layers_xyz.7 ... d9 -> layers_xyz.1
    if (l == 5) return P[pl.w_src[5] + o * 352 + 96 + c];
    return P[pl.w_src[l] + o * 256 + c];
}

#ifndef F32_REC
__global__ void mlp_f32_pack_t_kernel(const float *__restrict__ P, PlanT pl, float *__restrict__ packed) {
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= pl.total) return;
    int d = NBSTEPS - 1;
    while (d > 0 && idx < pl.step_off[d]) --d;
    const int ld = kBK[d] + 4, local = idx - pl.step_off[d];
    const int sl = local / slice_floats(kBK[d]), within = local % slice_floats(kBK[d]);
    float v = 0.0f;
    if (within < 32 * ld) {
        const int c = 32 * sl + within / ld, o = within % ld;
        if (o < kBK[d]) v = src_wt(P, pl, d, c, o);
    }
    packed[idx] = v;
}

DDN_EXPORT int ddnerf_mlp_f32_pack_t(const float *params, int depth_head, float *packed_t, ddnerf_stream_t stream) {
    DDN_REQUIRE(params && packed_t, DDNERF_E_ARG);
    DDN_REQUIRE(ddn_aligned(packed_t, 16), DDNERF_E_ALIGN);
    PlanT pl = make_plan_t(depth_head);
    hipLaunchKernelGGL(mlp_f32_pack_t_kernel, dim3((pl.total + 255) / 256), dim3(256), 0, (hipStream_t)stream, params, pl,
                       packed_t);
    return ddn_launch_status();
}
#endif

#define TADDR(row, ld, col) ((size_t)(row) * (ld) + (col))
// streaming data is written / read non-temporally: it must not evict the weight slices every block re-reads from L2
#ifdef F32_REC  // the record build writes two 64-byte half lines per store: write-back caching merges them (19.9 -> 19.4 ms per
                // fp32 training step; in the MFMA-bound x3 kernels the same change costs time, mlp_mfma16.inc)
#define TSTORE(v, p) (*(p) = (v))
#else
#define TSTORE(v, p) __builtin_nontemporal_store(v, p)
#endif
#define TLOAD(p) __builtin_nontemporal_load(p)
// ---- shared: store / load one accumulator tile in the transposed [feature][sample] matrices -------------------
#ifdef F32_REC
// Second build of this file (mlp_f32_train_rec.hip): `acts` / `deltas` are RECORDS of blocked hi/lo words, the operand format of
// the packed weight-gradient kernel (include/ddnerf_hip.h, ddnerf_mlp_x3_wgrad_packed): element (row, sample m) = one word
// (bf16 hi << 16) | bf16 lo at word index ((m >> 4) * 2560 + row) * 16 + (m & 15).  The bf16x3 weight gradients split the
// fp32 values exactly like this anyway; recording the split lets them stream contiguous runs at 4.5 - 5 TB/s instead of
// [feature][sample] rows at 3.  A value is > 0 iff its word is != 0.
#if F32_REC == 2
// Third build (mlp_f32_train_recp.hip): records of bf16 ROW PAIRS, the x3 training tier's format (include/ddnerf_hip.h,
// ddnerf_mlp_x3_wgrad_pairs): ONE word = bf16(row 2p) | bf16(row 2p + 1) << 16 at word index ((m >> 4) * 1280 + p) * 16 + (m & 15) -- half
// the record bytes of the hi/lo words, and weight gradients with one MFMA per product on bf16-ROUNDED operands (2.2e-3 of the
// gradient's norm from the exact tier's): an opt-in speed mode of the fp32 tier (DDNERF_WGRAD=pairs), never its default.  Rows
// 2k and 2k + 1 of a 32x32 accumulator tile are consecutive registers of one lane, so a pair word is one v_cvt_pk_bf16_f32.
#define F32_NAME(x) x##_recp
typedef __bf16 rp_bf16x2 __attribute__((ext_vector_type(2)));
typedef float rp_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ size_t rec_base(int row0, size_t col, int h) { return ((col >> 4) * (ACT_ROWS / 2) + (row0 + 4 * h) / 2) * 16 + (col & 15); }
__device__ __forceinline__ void store_tile_t(float *__restrict__ mat, size_t, int row0, size_t col, int h, const f32x16 &v) {
    unsigned *p = (unsigned *)mat + rec_base(row0, col, h);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const rp_f32x2 two = {v[2 * k], v[2 * k + 1]};
        TSTORE(__builtin_bit_cast(unsigned, __builtin_convertvector(two, rp_bf16x2)), p + (tile_row(2 * k, 0) / 2) * 16);
    }
}
// (the backward pass only asks whether a recorded activation is > 0: the two halves of a word travel as two "floats" whose bits are
// the bf16 patterns)
__device__ __forceinline__ f32x16 load_tile_t(const float *__restrict__ mat, size_t, int row0, size_t col, int h) {
    const unsigned *p = (const unsigned *)mat + rec_base(row0, col, h);
    f32x16 v;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const unsigned w = TLOAD(p + (tile_row(2 * k, 0) / 2) * 16);
        v[2 * k] = __builtin_bit_cast(float, w & 0xffffu);
        v[2 * k + 1] = __builtin_bit_cast(float, w >> 16);
    }
    return v;
}
__device__ __forceinline__ bool tile_positive(float w) { return (__builtin_bit_cast(unsigned, w) & 0x7fffu) != 0u; }
#elif F32_REC == 3
// Fourth build (mlp_f32_train_recf.hip, round 5): records in the SAME blocked layout that hold the fp32 VALUES -- element (row, sample m) at
// word index ((m >> 4) * 2560 + row) * 16 + (m & 15) -- for the weight-gradient kernel that splits them into bf16 hi / lo while it builds
// its MFMA fragments (ddnerf_mlp_x3_wgrad_blocked: the same split, the same products, bit-identical weight gradients).  What it buys: the
// split was 3.5 vector-ALU instructions per element in kernels whose fp32 MFMA chain stops for every one of them (4 gaps x (12 + 4 x 14)
// cycles per 32 x 32 tile of the forward, 3.3 % of its 8192 MFMA cycles; 2.7 % of the backward); the weight-gradient kernels wait for
// HBM and have the vector ALU idle.
#define F32_NAME(x) x##_recf
typedef unsigned u32x4_rec __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned rec_word(float x) { return __builtin_bit_cast(unsigned, x); }
__device__ __forceinline__ void rec_word2(float x0, float x1, unsigned &w0, unsigned &w1) {
    w0 = __builtin_bit_cast(unsigned, x0);
    w1 = __builtin_bit_cast(unsigned, x1);
}
__device__ __forceinline__ size_t rec_base(int row0, size_t col, int h) { return ((col >> 4) * ACT_ROWS + row0 + 4 * h) * 16 + (col & 15); }
__device__ __forceinline__ void store_tile_t(float *__restrict__ mat, size_t, int row0, size_t col, int h, const f32x16 &v) {
    float *p = mat + rec_base(row0, col, h);
#pragma unroll
    for (int r = 0; r < 16; ++r) TSTORE(v[r], p + tile_row(r, 0) * 16);
}
__device__ __forceinline__ f32x16 load_tile_t(const float *__restrict__ mat, size_t, int row0, size_t col, int h) {
    const float *p = mat + rec_base(row0, col, h);
    f32x16 v;
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = TLOAD(p + tile_row(r, 0) * 16);
    return v;
}
__device__ __forceinline__ bool tile_positive(float a) { return a > 0.0f; }
#else
#define F32_NAME(x) x##_rec
__device__ __forceinline__ unsigned rec_word(float x) {
    const __bf16 hi = (__bf16)x;
    const __bf16 lo = (__bf16)(x - (float)hi);
    return ((unsigned)__builtin_bit_cast(unsigned short, hi) << 16) | __builtin_bit_cast(unsigned short, lo);
}
// two at once, 3.5 instructions per element: both hi parts by one v_cvt_pk_bf16_f32, a shift / mask back to fp32, the subtraction, and one
// v_cvt_pk_bf16_f32 per word (its high half converts hi's fp32 image, exactly, its low half rounds lo) -- the words rec_word gives
typedef unsigned u32x4_rec __attribute__((ext_vector_type(4)));
typedef __bf16 rw_bf16x2 __attribute__((ext_vector_type(2)));
typedef float rw_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void rec_word2(float x0, float x1, unsigned &w0, unsigned &w1) {
    const unsigned h01 = __builtin_bit_cast(unsigned, __builtin_convertvector((rw_f32x2){x0, x1}, rw_bf16x2));
    const float hf0 = __builtin_bit_cast(float, h01 << 16), hf1 = __builtin_bit_cast(float, h01 & 0xffff0000u);
    w0 = __builtin_bit_cast(unsigned, __builtin_convertvector((rw_f32x2){x0 - hf0, hf0}, rw_bf16x2));
    w1 = __builtin_bit_cast(unsigned, __builtin_convertvector((rw_f32x2){x1 - hf1, hf1}, rw_bf16x2));
}
__device__ __forceinline__ size_t rec_base(int row0, size_t col, int h) { return ((col >> 4) * ACT_ROWS + row0 + 4 * h) * 16 + (col & 15); }
__device__ __forceinline__ void store_tile_t(float *__restrict__ mat, size_t, int row0, size_t col, int h, const f32x16 &v) {
    unsigned *p = (unsigned *)mat + rec_base(row0, col, h);
#pragma unroll
    for (int r = 0; r < 16; ++r) TSTORE(rec_word(v[r]), p + tile_row(r, 0) * 16);
}
__device__ __forceinline__ f32x16 load_tile_t(const float *__restrict__ mat, size_t, int row0, size_t col, int h) {
    const float *p = mat + rec_base(row0, col, h);
    f32x16 v;
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = TLOAD(p + tile_row(r, 0) * 16);
    return v;
}
__device__ __forceinline__ bool tile_positive(float w) { return __builtin_bit_cast(unsigned, w) != 0u; }
#endif
#else
#define F32_NAME(x) x
__device__ __forceinline__ void store_tile_t(float *__restrict__ mat, size_t ld, int row0, size_t col, int h,
                                             const f32x16 &v) {
#pragma unroll
    for (int r = 0; r < 16; ++r) TSTORE(v[r], &mat[TADDR(row0 + tile_row(r, h), ld, col)]);
}
__device__ __forceinline__ f32x16 load_tile_t(const float *__restrict__ mat, size_t ld, int row0, size_t col, int h) {
    f32x16 v;
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = TLOAD(&mat[TADDR(row0 + tile_row(r, h), ld, col)]);
    return v;
}
__device__ __forceinline__ bool tile_positive(float a) { return a > 0.0f; }
#endif

// ---- forward that records activations -----------------------------------------------------------------------
// The inference forward (mlp_f32_fwd.inc: persistent tiles, every output tile takes an LDS round trip behind the next slice's MFMAs)
// with a recorder: the tile that round trip left in registers -- post-ReLU where the layer has one -- is converted to the record's
// format in ONE gap between two MFMAs (an fp32 MFMA does not overlap the wave's own vector-ALU instructions: every gap that holds any
// costs 12 cycles, every instruction 4) and stored by buffer stores, which cost the MFMA chain nothing (scalar tile base + 32-bit lane
// offset; a store with a 64-bit vector address costs 12 cycles).  Round 3's form (activation, conversion and sixteen 64-bit-address
// stores behind each slice's last MFMA, 1.2 vector-ALU instructions per MFMA) ran at 0.80 of the inference kernel's rate.
#include "mlp_f32_fwd.inc"

// ReLU masks (round 5, the values-record build): one bit per recorded activation of layers_xyz.0-7, laid out for the BACKWARD kernel's wave:
// in the 32 x 32 accumulator layout a v_cmp on register r of a tile gives 64 bits -- bit 32 h + j = (row tile_row(r, h), sample j) > 0 --
// in a scalar register pair, which is exactly the condition operand of the v_cndmask that masks register r of the delta tile of the same
// rows and samples.  Sign record: [128-sample tile][32-row block rb = row / 32, 64 of them][wave, 4][register r, 16] 64-bit masks: 32 KiB
// per tile, 1 bit per value (the backward read 32 bits per value only to take this bit: 4.8 GB per fine pass).  Written by scalar stores
// (s_store_dwordx2: no vector-ALU instruction between the compare and memory; data read at issue, tools/calib/sstore.hip), read by scalar
// loads.
#define F32_SIGN_TILE_BYTES (64 * 4 * 16 * 8)
#if defined(F32_REC) && F32_REC == 3 && !defined(F32_NO_SIGNS)   // (-DF32_NO_SIGNS: the A/B build of tools/train_kernels_ab.py -- same
#define F32_SIGNS 1                                               // entry points, masks from the recorded activations as in the other builds)
#else
#define F32_SIGNS 0
#endif
struct Recorder {
    static constexpr bool kActive = true;
#ifdef F32_TRAIN_DMA   // (the weight slices by LDS-DMA, as in the inference kernel; the record stores stay outstanding across the slice's vmcnt wait)
    static constexpr bool kDma = true;
#else
    static constexpr bool kDma = false;
#endif
#if F32_SIGNS
    static constexpr bool kSigns = true;
    const char *signs;        // the sign record
    const char *sbase;        // ... of this tile and wave (uniform)
    unsigned long sg[4];      // the four masks of a gap, until their stores
    __device__ __forceinline__ void sign4(const f32x16 &t, int g) {
#pragma unroll
        for (int k = 0; k < 4; ++k) asm volatile("v_cmp_lt_f32 %0, 0, %1" : "=s"(sg[k]) : "v"(t[4 * g + k]));
    }
    __device__ __forceinline__ void sign_store(int rec_row, int g) {
#ifdef F32_EXP_NOSSTORE   // (diagnostic builds, tools/train_kernels_ab.py: timing only)
        return;
#endif
        // (the scheduler otherwise hoists the stores over the MFMA in front of them, right behind the compares that wrote their data)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < 4; ++k)
            asm volatile("s_store_dwordx2 %0, %1, %2" ::"s"(sg[k]), "s"(sbase), "s"((unsigned)((rec_row / 32) * 512 + (4 * g + k) * 8)) : "memory");
    }
#else
    static constexpr bool kSigns = false;
    __device__ __forceinline__ void sign4(const f32x16 &, int) {}
    __device__ __forceinline__ void sign_store(int, int) {}
#endif
#if defined(F32_REC) && (F32_REC == 1 || F32_REC == 3) && !defined(F32_EXP_NOQUADS)   // (F32_EXP_NOQUADS: diagnostic build with the 4-byte stores)
    static constexpr bool kQuads = true;   // hi/lo words: one word per value, four consecutive samples of a row = 16 contiguous bytes
    unsigned voff_t;                       // this lane's byte offset for a quad (mlp_f32_fwd.inc: the transposed read of the scratch)
#else
    static constexpr bool kQuads = false;
    __device__ __forceinline__ void store_quad(int, int, f32x4) {}
#endif
    float *acts;
    size_t ld;
#ifdef F32_REC
    __amdgpu_buffer_rsrc_t rs;  // this tile's 8 sample blocks of the record
    unsigned voff;              // this lane's byte offset inside them
#if F32_REC == 2
    static constexpr unsigned kBlockWords = (ACT_ROWS / 2) * 16;
#else
    static constexpr unsigned kBlockWords = ACT_ROWS * 16;
#endif
    __device__ __forceinline__ void begin_tile(long tile, int wave, int j, int h) {
#ifdef F32_EXP_SMALLSTORE   // (diagnostic builds: the workgroups of an XCD share the records of two tiles: the stores stay in its L2)
        tile = blockIdx.x & 15;   // (2 tiles = 2.6 MB per XCD)
#endif
        rs = __builtin_amdgcn_make_buffer_rsrc((void *)((unsigned *)acts + (size_t)tile * 8 * kBlockWords), 0, 8 * kBlockWords * 4, 0x00020000);
#if F32_SIGNS
        sbase = signs + (size_t)tile * F32_SIGN_TILE_BYTES + (size_t)__builtin_amdgcn_readfirstlane(wave) * 128;
#endif
#if F32_REC == 2
        voff = 4u * ((unsigned)(wave * 2 + (j >> 4)) * kBlockWords + 2u * h * 16u + (j & 15));
#else
        voff = 4u * ((unsigned)(wave * 2 + (j >> 4)) * kBlockWords + 4u * h * 16u + (j & 15));
#ifndef F32_EXP_NOQUADS
        const unsigned L = 32u * h + j;   // quad i of lane L: row 8 i + (L >> 4) + 4 ((L >> 3) & 1), samples 4 (L & 7) .. + 3 of the wave's 32
        voff_t = 4u * ((unsigned)(wave * 2 + ((L & 7) >> 2)) * kBlockWords + ((L >> 4) + 4u * ((L >> 3) & 1)) * 16u + 4u * (L & 3));
#endif
#endif
    }
#if (F32_REC == 1 || F32_REC == 3) && !defined(F32_EXP_NOQUADS)
    __device__ __forceinline__ void store_quad(int row0, int i, f32x4 x) {
        unsigned w0, w1, w2, w3;
        rec_word2(x.x, x.y, w0, w1);
        rec_word2(x.z, x.w, w2, w3);
        const u32x4_rec w = {w0, w1, w2, w3};
#ifdef F32_EXP_NOSTORE
        asm volatile("" ::"v"(w));
#else
        __builtin_amdgcn_raw_buffer_store_b128(w, rs, voff_t, 64u * (row0 + 8 * i), 0);
#endif
    }
#endif
    // registers 4g .. 4g+3 of a tile = its rows 8g .. 8g+3 (+ 4h)
    __device__ __forceinline__ void store4(int row0, const f32x16 &v, int g) {
#if F32_REC == 2
#pragma unroll
        for (int k = 2 * g; k < 2 * g + 2; ++k) {
            const rp_f32x2 two = {v[2 * k], v[2 * k + 1]};
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, __builtin_convertvector(two, rp_bf16x2)), rs, voff,
                                                  64u * (row0 / 2 + tile_row(2 * k, 0) / 2), 0);
        }
#else
#pragma unroll
        for (int r = 4 * g; r < 4 * g + 4; r += 2) {
            unsigned w0, w1;
            rec_word2(v[r], v[r + 1], w0, w1);
#ifdef F32_EXP_NOSTORE   // (diagnostic builds, tools/f32_clock.py: timing only)
            asm volatile("" ::"v"(w0), "v"(w1));
#else
            __builtin_amdgcn_raw_buffer_store_b32(w0, rs, voff, 64u * (row0 + tile_row(r, 0)), 0);
            __builtin_amdgcn_raw_buffer_store_b32(w1, rs, voff, 64u * (row0 + tile_row(r + 1, 0)), 0);
#endif
        }
#endif
    }
    __device__ __forceinline__ void store(int row0, const f32x16 &v) {
#pragma unroll
        for (int g = 0; g < 4; ++g) store4(row0, v, g);
    }
#else
    // fp32 matrices [feature][sample]: the row's address is scalar arithmetic, this lane's column -- and its lane half's four rows -- a
    // 32-bit byte offset (global store with a scalar base: nothing for the vector ALU to add)
    unsigned voff;
    size_t ld_tile;   // = ld, opaque per tile: the 1280 row addresses are computed where they are used (a few scalar instructions each), not
                      // hoisted out of the tile loop into 2560 scalar registers
    float *col0;
    __device__ __forceinline__ void begin_tile(long tile, int wave, int j, int h) {
        voff = 4u * (unsigned)(4 * (size_t)h * ld + wave * 32 + j);   // the launcher checks 20 ld < 2^32 bytes
        ld_tile = ld;
        asm volatile("" : "+s"(ld_tile));
        col0 = acts + (size_t)tile * 128;
    }
    __device__ __forceinline__ void store4(int row0, const f32x16 &v, int g) {
#pragma unroll
        for (int r = 4 * g; r < 4 * g + 4; ++r) *(float *)((char *)(col0 + (size_t)(row0 + tile_row(r, 0)) * ld_tile) + voff) = v[r];
    }
    __device__ __forceinline__ void store(int row0, const f32x16 &v) {
#pragma unroll
        for (int g = 0; g < 4; ++g) store4(row0, v, g);
    }
#endif
};

struct FwdOffsets {
    int bias_off[NLAYERS];
};

// (the values-record build takes one more buffer: the sign record, F32_SIGN_TILE_BYTES per 128-sample tile)
#if defined(F32_REC) && F32_REC == 3
DDN_EXPORT size_t ddnerf_mlp_f32_sign_bytes(long ld) { return ld > 0 ? (size_t)((ld + 127) / 128) * F32_SIGN_TILE_BYTES : 0; }
#define F32_SIGNS_PARAM , void *__restrict__ signs
#define F32_SIGNS_CPARAM , const void *__restrict__ signs
#define F32_SIGNS_ARG , signs
// ... and, optionally, the view-direction columns from a per-ray table [n,32] (ddnerf_encode_rays) instead of columns 96..127 of the sample's
// own row, as the inference kernel takes them (mlp_f32_fwd.inc, F32Dirs); the record of the input columns is the same either way
#define F32_DIRS_KPARAM , F32Dirs ds
#define F32_DIRS_KARG , ds
#define F32_DIRS_PARAM , const float *dirs, int S
#else
#define F32_SIGNS_PARAM
#define F32_SIGNS_CPARAM
#define F32_SIGNS_ARG
#define F32_DIRS_KPARAM
#define F32_DIRS_KARG
#define F32_DIRS_PARAM
#endif
template <bool DEPTH>
__global__ __launch_bounds__(256, 1) void F32_NAME(mlp_f32_fwd_train_kernel)(const float *__restrict__ feat,
                                                                   const float *__restrict__ packed, FwdOffsets fo,
                                                                   float *__restrict__ raw, float *__restrict__ acts,
                                                                   long M, long ld F32_SIGNS_PARAM F32_DIRS_KPARAM) {
    __shared__ __attribute__((aligned(16))) float lds[F32_FWD_LDS_FLOATS];
    Recorder rec;
    rec.acts = acts;
    rec.ld = (size_t)ld;
#if F32_SIGNS
    rec.signs = (const char *)signs;
#endif
    // (tiles past M inside ld -- ld is a multiple of 128 -- are recorded too, from the last sample's features: their deltas are zero)
    mlp_f32_forward_tiles<DEPTH>(lds, feat, packed, 4u * (unsigned)fo.bias_off[0], raw, M, rec F32_DIRS_KARG);
#if F32_SIGNS
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_dcache_wb" ::: "memory");   // the scalar stores leave the scalar data cache
#endif
}

static FwdOffsets fwd_offsets() {
    FwdOffsets fo;
    int off = 0;
    for (int l = 0; l < NLAYERS; ++l) off += kLayerNB[l] * slice_floats(kLayerK[l]);
    for (int l = 0; l < NLAYERS; ++l) {
        fo.bias_off[l] = off;
        off += kLayerNB[l] * 32;
    }
    return fo;
}

DDN_EXPORT int F32_NAME(ddnerf_mlp_f32_forward_train)(const float *feat, const float *packed, int depth_head, float *raw,
                                            float *acts F32_SIGNS_PARAM F32_DIRS_PARAM, long M, long ld, ddnerf_stream_t stream) {
    DDN_REQUIRE(feat && packed && raw && acts, DDNERF_E_ARG);
#if defined(F32_REC) && F32_REC == 3
    DDN_REQUIRE(signs && ddn_aligned(signs, 128), DDNERF_E_ARG);
    F32Dirs ds = {nullptr, 0u};
    if (dirs) {   // (n S^2 < 2^32: the ray of a sample comes from a 32-bit multiply-high, as in ddnerf_mlp_f32_forward_rays)
        DDN_REQUIRE(S > 1 && M % S == 0 && ddn_aligned(dirs, 16), DDNERF_E_ARG);
        DDN_REQUIRE((unsigned long long)M * (unsigned long long)S < (1ull << 32), DDNERF_E_RANGE);
        ds = F32Dirs{dirs, (unsigned)(((1ull << 32) + (unsigned)S - 1) / (unsigned)S)};
    }
#endif
    DDN_REQUIRE(M > 0, DDNERF_E_ARG);
    DDN_REQUIRE(ld >= M && ld % 128 == 0, DDNERF_E_RANGE);
#ifndef F32_REC
    DDN_REQUIRE(ld <= (1l << 27), DDNERF_E_RANGE);  // (the fp32 matrices' 32-bit store offsets: 20 ld bytes)
#endif
    DDN_REQUIRE(ddn_aligned(feat, 16) && ddn_aligned(packed, 16) && ddn_aligned(raw, 16), DDNERF_E_ALIGN);
    FwdOffsets fo = fwd_offsets();
    const long ntiles = (M + 127) / 128, cus = ddn_cu_count();
    dim3 grid((unsigned)(ntiles < cus ? ntiles : cus));   // persistent (mlp_f32_fwd.inc)
    if (depth_head)
        hipLaunchKernelGGL(F32_NAME(mlp_f32_fwd_train_kernel)<true>, grid, dim3(256), 0, (hipStream_t)stream, feat, packed, fo, raw,
                           acts, M, ld F32_SIGNS_ARG F32_DIRS_KARG);
    else
        hipLaunchKernelGGL(F32_NAME(mlp_f32_fwd_train_kernel)<false>, grid, dim3(256), 0, (hipStream_t)stream, feat, packed, fo, raw,
                           acts, M, ld F32_SIGNS_ARG F32_DIRS_KARG);
    return ddn_launch_status();
}

// ---- fused backward-data ------------------------------------------------------------------------------------
// One backward step: NB tiles of the outgoing delta.  MASK: multiply by relu'(recorded activation), whose tile is
// requested at the start of its slice and used a whole slice later.
// (step d0 only -- K = 32: four MFMA chunks per slice, too short for the schedule below)
template <int KIND, int K, int NB, int NEXT_K, int PAR, bool MASK>
__device__ __forceinline__ void step_bwd(const float *__restrict__ &wp, float *lds, const f32x16 (&Breg)[12],
                                         f32x16 (&out)[8], const float *__restrict__ acts, float *__restrict__ deltas,
                                         size_t ld, int row0, size_t col, int tid, int lane) {
    constexpr int N4 = slice_floats(K) / 4;
    constexpr int NEXT_N4 = NEXT_K > 0 ? slice_floats(NEXT_K) / 4 : 0;
    const int h = lane >> 5;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const float *cur = lds + ((PAR + b) & 1) * MAX_SLICE_FLOATS;
        float *nxt = lds + ((PAR + b + 1) & 1) * MAX_SLICE_FLOATS;
        wp += 4 * N4;
        f32x16 act;
        auto init = [&](f32x16 &a) {
#pragma unroll
            for (int r = 0; r < 16; ++r) a[r] = 0.0f;
            if (MASK) act = load_tile_t(acts, ld, row0 + 32 * b, col, h);
        };
        auto mid = [](int, int) {};
        auto post = [&](f32x16 &a) {
            if (MASK) {
#pragma unroll
                for (int r = 0; r < 16; ++r) a[r] = tile_positive(act[r]) ? a[r] : 0.0f;
            }
            store_tile_t(deltas, ld, row0 + 32 * b, col, h, a);
        };
        if (b + 1 < NB) slice_step_hooks<KIND, K, N4>(wp, cur, nxt, Breg, out[b], tid, lane, init, mid, post);
        else slice_step_hooks<KIND, K, NEXT_N4>(wp, cur, nxt, Breg, out[b], tid, lane, init, mid, post);
    }
}

// The recorded activations of one tile, for the ReLU masks: the record words of rows row0 + tile_row(r, h), as load_tile_t gives them
// (tile_positive() reads them), by buffer loads where the record's blocked layout allows a 32-bit offset.
struct ActLoader {
    const float *acts;
    size_t ld;
#ifdef F32_REC
    __amdgpu_buffer_rsrc_t rs;
    unsigned voff;
    __device__ __forceinline__ void begin_tile(long tile, int wave, int j, int h) {
#ifdef F32_EXP_SMALLLOAD   // (diagnostic builds: the activation records of two tiles per XCD serve every tile: the fetches stay in L2)
        tile = blockIdx.x & 15;
#endif
        rs = __builtin_amdgcn_make_buffer_rsrc((void *)((const unsigned *)acts + (size_t)tile * 8 * Recorder::kBlockWords), 0, 8 * Recorder::kBlockWords * 4, 0x00020000);
#if F32_REC == 2
        voff = 4u * ((unsigned)(wave * 2 + (j >> 4)) * Recorder::kBlockWords + 2u * h * 16u + (j & 15));
#else
        voff = 4u * ((unsigned)(wave * 2 + (j >> 4)) * Recorder::kBlockWords + 4u * h * 16u + (j & 15));
#endif
    }
    // registers 4g .. 4g+3 of the tile's record words
    __device__ __forceinline__ void load4(f32x16 &v, int row0, int g) const {
#if F32_REC == 2
#pragma unroll
        for (int k = 2 * g; k < 2 * g + 2; ++k) {
            const unsigned w = __builtin_amdgcn_raw_buffer_load_b32(rs, voff, 64u * (row0 / 2 + tile_row(2 * k, 0) / 2), 0);
            v[2 * k] = __builtin_bit_cast(float, w & 0xffffu);
            v[2 * k + 1] = __builtin_bit_cast(float, w >> 16);
        }
#else
#pragma unroll
        for (int r = 4 * g; r < 4 * g + 4; ++r) v[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff, 64u * (row0 + tile_row(r, 0)), 0));
#endif
    }
    __device__ __forceinline__ f32x16 load(int row0) const {
        f32x16 v;
#if F32_REC == 2
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const unsigned w = __builtin_amdgcn_raw_buffer_load_b32(rs, voff, 64u * (row0 / 2 + tile_row(2 * k, 0) / 2), 0);
            v[2 * k] = __builtin_bit_cast(float, w & 0xffffu);
            v[2 * k + 1] = __builtin_bit_cast(float, w >> 16);
        }
#else
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff, 64u * (row0 + tile_row(r, 0)), 0));
#endif
        return v;
    }
#else
    size_t col;
    int h;
    __device__ __forceinline__ void begin_tile(long tile, int wave, int j, int h_) {
        col = (size_t)tile * 128 + wave * 32 + j;
        h = h_;
    }
    __device__ __forceinline__ f32x16 load(int row0) const { return load_tile_t(acts, ld, row0, col, h); }
    __device__ __forceinline__ void load4(f32x16 &v, int row0, int g) const {
#pragma unroll
        for (int r = 4 * g; r < 4 * g + 4; ++r) v[r] = TLOAD(&acts[TADDR(row0 + tile_row(r, h), ld, col)]);
    }
#endif
};

// Steps d1 .. d9 on the forward's slice step (mlp_f32_fwd.inc): the tile of slice b - 1 goes through the wave's LDS scratch behind the
// MFMAs of slice b -- pushed and pulled back into vector registers by LDS instructions (free beside MFMAs; an accumulator-file register
// cannot be a vector-ALU source) -- and is masked (a compare and a select per element), converted and recorded behind the slice's
// last weight fetch.  The accumulators start from the MFMA's zero operand, not from sixteen moves.  The step's last tile makes the trip
// behind the next step's slice 0 (PEND: 1 masked by `act_pend`, the activation tile this step fetched for it; 2 unmasked).
// The recorded activations (the masks) come from HBM, and vmcnt retires loads in order: a record fetch issued in FRONT of a slice's
// weight fetches must have arrived before the first of those may be parked, half a slice (1.8 us) later.  They are issued BEHIND the
// slice's last weight fetch and its record stores instead, two slices ahead of their use (tile b + 1 during slice b, used behind
// slice b + 2; the next step's tile 0 during this step's last slice: NEXT_MASK, next_row0): three activation tiles in flight.
template <int KIND, int K, int NB, int NEXT_K, int PAR, bool MASK, int PEND, bool NEXT_MASK>
__device__ __forceinline__ void step_bwd_early(__amdgpu_buffer_rsrc_t wsrc, unsigned &woff, float *lds, float *scr_wave, f32x16 (&Breg)[12],
                                               f32x16 (&out)[8], f32x4 (&carry)[2], f32x16 &act_pend, f32x16 &act_first, const ActLoader &al,
                                               Recorder &rec, int row0, int pend_row0, int next_row0, int tid, int lane) {
    static_assert(!PEND || bsel<KIND>(9) != 7, "PEND: tile 7 must not feed the first ten chunks");
    static_assert(K / 8 >= 20, "the round-trip schedule uses chunks 1 .. q0 + 7");
    constexpr int ROUNDS = slice_floats(K) / 1024, NEXT_ROUNDS = NEXT_K > 0 ? slice_floats(NEXT_K) / 1024 : 0, NQ = K / 8;
    float *scr = scr_wave + lane;
    f32x16 act_prev = act_pend;    // the activation tile of the tile in flight through the scratch (slice 0: the step before's last)
    f32x16 act_hold = act_first;   // ... of the tile this slice computes
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const float *cur = lds + ((PAR + b) & 1) * MAX_SLICE_FLOATS;
        float *nxt = lds + ((PAR + b + 1) & 1) * MAX_SLICE_FLOATS;
        woff += 4096u * ROUNDS;
        f32x16 act_load = act_hold, bunused;   // ... of the tile the NEXT slice computes, fetched during this one
        auto init = [&](f32x16 &a) { a = f32x16{0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f}; };
        auto mid = [&](int q, int i) {
            const int R = b + 1 < NB ? ROUNDS : NEXT_ROUNDS;
            const int after = R > 0 ? ((R - 1) * NQ) / (2 * R) + 1 : 7, q0 = after > 7 ? after : 7;
            static_assert(NQ >= 20, "");
            if (i == 1 && q >= q0 + 4 && q < q0 + 8) {
                if (b + 1 < NB) {
                    if (MASK) al.load4(act_load, row0 + 32 * (b + 1), q - q0 - 4);
                } else if (NEXT_MASK) {
                    al.load4(act_load, next_row0, q - q0 - 4);
                }
            }
            const bool mine = b >= 1, pend = PEND != 0 && b == 0;
            if (!mine && !pend) return;
            f32x16 &t = mine ? out[b >= 1 ? b - 1 : 0] : Breg[7];
            const bool masked = mine ? MASK : PEND == 1;
            if (q >= 1 && q <= 4) plain_push(scr, t[4 * (q - 1) + i], 4 * (q - 1) + i);
            if (q >= 5 && q <= 8) t[4 * (q - 5) + i] = relu_pull(scr, 4 * (q - 5) + i);
            if (i == 0 && q >= q0 && q < q0 + 4) {   // mask (compare + select) and record four elements in one gap
                if (masked) {
#pragma unroll
                    for (int r = 4 * (q - q0); r < 4 * (q - q0) + 4; ++r) t[r] = tile_positive(act_prev[r]) ? t[r] : 0.0f;
                }
                // (4-byte stores here: the forward's 16-byte form -- the masked tile written back into the scratch, read transposed, stored as
                // quads a few chunks later -- measured 2.8 % SLOWER in this kernel, 4.91 against 4.77 ms)
                rec.store4(mine ? row0 + 32 * (b - 1) : pend_row0, t, q - q0);
            }
        };
        if (b + 1 < NB) slice_step_early<KIND, K, ROUNDS, K, false>(wsrc, nullptr, woff, cur, nxt, Breg, out[b], carry, bunused, tid, lane, init, mid);
        else slice_step_early<KIND, K, NEXT_ROUNDS, NEXT_K, false>(wsrc, nullptr, woff, cur, nxt, Breg, out[b], carry, bunused, tid, lane, init, mid);
        act_prev = act_hold;
        act_hold = act_load;
    }
    act_pend = act_prev;
    act_first = act_hold;
}

#if F32_SIGNS
// The values-record build (round 5): the ReLU masks of steps d2 .. d9 come from the forward's SIGN record (Recorder above) instead of from
// the recorded activations -- per 32 x 32 tile two s_load_dwordx16 (one 64-bit mask per accumulator register, 128 contiguous bytes per
// wave and tile) instead of sixteen 4-byte vector loads per lane (4.8 GB per fine pass, and every vector-memory instruction costs the fp32
// MFMA chain ~12 cycles), and ONE v_cndmask per register on the scalar pair instead of a compare and a select.
// The scalar loads are inline assembly the compiler's s_waitcnt insertion does not see (a compiler-issued scalar load would make the
// NEXT wait for an LDS result an lgkmcnt(0) -- scalar loads return out of order -- and stall the MFMA chain for the load's full latency):
//   * the masks of the tile a slice computes are requested behind that slice's record stores (chunks q0 + 4, q0 + 5), into the ONE
//     set of 16 scalar pairs whose last use (the tile before's) was chunk q0 + 3;
//   * `s_waitcnt lgkmcnt(0)` sits behind the last MFMA of chunk q0 - 1 of the NEXT slice (3 us later; nothing of the wave's own LDS
//     traffic is in flight there), in front of their first use;
//   * a hidden outstanding scalar load only makes the compiler's own lgkmcnt(N) waits conservative (one more LDS operation has to be
//     back than it asked for), never wrong;
//   * between a request and that wait NO instruction may touch the pairs (a register-allocator copy would read them early):
//     csrc/check_asm_hazards.py --sload scans the kernel's assembly for exactly that and fails the build.
typedef unsigned u32x16s __attribute__((ext_vector_type(16)));
struct SignMasks {   // the sixteen 64-bit lane masks of one tile: registers 0..7 in `a`, 8..15 in `b` (two 16-register scalar tuples)
    u32x16s a, b;
    // the data-dependent form of the wait: everything that reads the masks afterwards depends on THIS statement's results, so the
    // compiler cannot move a use (or the scalar arithmetic of one) in front of it
    __device__ __forceinline__ void arrived() { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(a), "+s"(b)::"memory"); }
    __device__ __forceinline__ unsigned long of(int r) const {
        const u32x16s &v = r < 8 ? a : b;
        return (unsigned long)v[2 * (r & 7)] | ((unsigned long)v[2 * (r & 7) + 1] << 32);
    }
};
struct SignLoader {
    const char *sbase;
    __device__ __forceinline__ void begin_tile(const void *signs, long tile, int wave) {
        sbase = (const char *)signs + (size_t)tile * F32_SIGN_TILE_BYTES + (size_t)__builtin_amdgcn_readfirstlane(wave) * 128;
    }
    // TWO loads per tile, not sixteen: every outstanding scalar load the compiler does not know of makes each of its lgkmcnt(N) waits for
    // an LDS result wait for one operation more than it meant to -- with sixteen in flight (N <= 15) every such wait lasted until
    // the scalar loads were back, 1 - 2 us per tile: the kernel ran 6 % SLOWER than with the fp32 activation loads
    __device__ __forceinline__ void load_half(SignMasks &sm, int row0, int half) const {
        const unsigned off = (unsigned)((row0 / 32) * 512 + half * 64);
        if (half == 0) asm volatile("s_load_dwordx16 %0, %1, %2" : "=&s"(sm.a) : "s"(sbase), "s"(off) : "memory");
        else asm volatile("s_load_dwordx16 %0, %1, %2" : "=&s"(sm.b) : "s"(sbase), "s"(off) : "memory");
    }
};
__device__ __forceinline__ float sign_mask(float x, unsigned long m) {
    asm volatile("v_cndmask_b32 %0, 0, %0, %1" : "+v"(x) : "s"(m));
    return x;
}

template <int KIND, int K, int NB, int NEXT_K, int PAR, bool MASK, int PEND>
__device__ __forceinline__ void step_bwd_signs(__amdgpu_buffer_rsrc_t wsrc, const char *wptr, unsigned &woff, float *lds, float *scr_wave, f32x16 (&Breg)[12],
                                               f32x16 (&out)[8], f32x4 (&carry)[2], SignMasks &sm, const SignLoader &sl,
                                               Recorder &rec, int row0, int pend_row0, int tid, int lane) {
    static_assert(!PEND || bsel<KIND>(9) != 7, "PEND: tile 7 must not feed the first ten chunks");
    static_assert(K / 8 >= 20, "the round-trip schedule uses chunks 1 .. q0 + 7");
    constexpr int ROUNDS = slice_floats(K) / 1024, NEXT_ROUNDS = NEXT_K > 0 ? slice_floats(NEXT_K) / 1024 : 0, NQ = K / 8;
    float *scr = scr_wave + lane;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const float *cur = lds + ((PAR + b) & 1) * MAX_SLICE_FLOATS;
        float *nxt = lds + ((PAR + b + 1) & 1) * MAX_SLICE_FLOATS;
        woff += 4096u * ROUNDS;
        f32x16 bunused;
        auto init = [&](f32x16 &a) { a = f32x16{0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f}; };
        auto mid = [&](int q, int i) {
            const int R = b + 1 < NB ? ROUNDS : NEXT_ROUNDS;
            const int after = R > 0 ? ((R - 1) * NQ) / (2 * R) + 1 : 7, q0 = after > 7 ? after : 7;
            static_assert(NQ >= 20, "");
            // the masks of the tile THIS slice computes (rows row0 + 32 b ..): used behind the next slice, or the next step's slice 0
            if (MASK && i == 1 && q >= q0 + 4 && q < q0 + 6) sl.load_half(sm, row0 + 32 * b, q - q0 - 4);
            const bool mine = b >= 1, pend = PEND != 0 && b == 0;
            if (!mine && !pend) return;
            f32x16 &t = mine ? out[b >= 1 ? b - 1 : 0] : Breg[7];
            const bool masked = mine ? MASK : PEND == 1;
            if (q >= 1 && q <= 4) plain_push(scr, t[4 * (q - 1) + i], 4 * (q - 1) + i);
            if (q >= 5 && q <= 8) t[4 * (q - 5) + i] = relu_pull(scr, 4 * (q - 5) + i);
            if (masked && q == q0 - 1 && i == 3) sm.arrived();   // the masks requested a slice ago
            if (i == 0 && q >= q0 && q < q0 + 4) {   // mask (one select per element) and record four elements in one gap
                if (masked) {
#pragma unroll
                    for (int r = 4 * (q - q0); r < 4 * (q - q0) + 4; ++r) t[r] = sign_mask(t[r], sm.of(r));
                }
                rec.store4(mine ? row0 + 32 * (b - 1) : pend_row0, t, q - q0);
            }
        };
        const int vm_after = (b >= 1 || PEND != 0) ? 16 : 0;   // (this slice's record stores, behind its last piece)
        if (b + 1 < NB) slice_step_early<KIND, K, ROUNDS, K, Recorder::kDma>(wsrc, wptr, woff, cur, nxt, Breg, out[b], carry, bunused, tid, lane, init, mid, vm_after);
        else slice_step_early<KIND, K, NEXT_ROUNDS, NEXT_K, Recorder::kDma>(wsrc, wptr, woff, cur, nxt, Breg, out[b], carry, bunused, tid, lane, init, mid, vm_after);
    }
}
#endif

template <bool DEPTH>
__global__ __launch_bounds__(256, 1) void F32_NAME(mlp_f32_bwd_data_kernel)(const float *__restrict__ g_raw,
                                                                  const float *__restrict__ packed_t, unsigned packed_bytes,
                                                                  const float *__restrict__ acts,
                                                                  float *__restrict__ deltas, long M, long ld F32_SIGNS_CPARAM) {
    __shared__ __attribute__((aligned(16))) float lds[F32_LDS_FLOATS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, h = lane >> 5;
    const long m = (long)blockIdx.x * 128 + wave * 32 + j;
    const bool valid = m < M;
    float *scr_wave = lds + 2 * MAX_SLICE_FLOATS + SCR_FLOATS * wave;
#pragma unroll
    for (int k = 0; k < 4; ++k) relu_rezero(scr_wave, lane, k);
    f32x16 Breg[12];
    f32x16 out[8];
    const float *wp = packed_t;
    {
        constexpr int ROUNDS = slice_floats(32) / 1024;
        f32x4 pf[ROUNDS];
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) pf[r] = *(const f32x4 *)(wp + 4 * (size_t)(r * 256 + tid));
        // the d(raw) tile: tile row c = raw column c (rows 0..3 on lane half 0, rows 4,5 on lane half 1)
#pragma unroll
        for (int r = 0; r < 16; ++r) Breg[8][r] = 0.0f;
        if (valid) {
            const float *g = g_raw + (size_t)m * (DEPTH ? 6 : 4);
            if (h == 0) {
                Breg[8][0] = g[0];
                Breg[8][1] = g[1];
                Breg[8][2] = g[2];
                Breg[8][3] = g[3];
            } else if (DEPTH) {
                Breg[8][0] = g[4];
                Breg[8][1] = g[5];
            }
        }
        store_tile_t(deltas, ld, ROW_X, m, h, Breg[8]);
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) *(f32x4 *)(lds + 4 * (r * 256 + tid)) = pf[r];
    }
    __syncthreads();
    // d0: d(dir hidden) = heads^T d(raw), masked by relu'(dir hidden)            4 slices: buffers 0,1,0,1 -> next in 0
    step_bwd<10, 32, 4, 160, 0, true>(wp, lds, Breg, out, acts, deltas, ld, ROW_DIR, m, tid, lane);
#pragma unroll
    for (int b = 0; b < 4; ++b) Breg[b] = out[b];
    // from here on: the forward's slice step (barrier in front of a slice's last chunk, first fragments handed over)
    const __amdgpu_buffer_rsrc_t wsrc = __builtin_amdgcn_make_buffer_rsrc((void *)packed_t, 0, packed_bytes, 0x00020000);
    unsigned woff = 4u * (unsigned)(wp - packed_t);
    f32x4 carry[2];
    carry[0] = *(const f32x4 *)(lds + j * (160 + 4) + 4 * h);
    carry[1] = *(const f32x4 *)(lds + j * (160 + 4) + 4 * h + 8);
    Recorder rec;
    rec.acts = deltas;
    rec.ld = (size_t)ld;
#define NEXT_STEP() _Pragma("unroll") for (int b = 0; b < 8; ++b) Breg[b] = out[b];
#if F32_SIGNS
    rec.signs = nullptr;   // (the backward writes no sign record)
    rec.begin_tile(blockIdx.x, wave, j, h);
    SignLoader sl;
    sl.begin_tile(signs, blockIdx.x, wave);
    SignMasks sm;
    sm.a = sm.b = u32x16s{0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
    step_bwd_signs<11, 160, 8, 256, 0, false, 0>(wsrc, (const char *)packed_t, woff, lds, scr_wave, Breg, out, carry, sm, sl, rec, ROW_FEAT, 0, tid, lane);
    NEXT_STEP();
    step_bwd_signs<1, 256, 8, 256, 0, true, 2>(wsrc, (const char *)packed_t, woff, lds, scr_wave, Breg, out, carry, sm, sl, rec, 256 * 7, ROW_FEAT + 224, tid, lane);
    NEXT_STEP();
    step_bwd_signs<1, 256, 8, 256, 0, true, 1>(wsrc, (const char *)packed_t, woff, lds, scr_wave, Breg, out, carry, sm, sl, rec, 256 * 6, 256 * 7 + 224, tid, lane);
    NEXT_STEP();
    step_bwd_signs<1, 256, 8, 256, 0, true, 1>(wsrc, (const char *)packed_t, woff, lds, scr_wave, Breg, out, carry, sm, sl, rec, 256 * 5, 256 * 6 + 224, tid, lane);
    NEXT_STEP();
    step_bwd_signs<1, 256, 8, 256, 0, true, 1>(wsrc, (const char *)packed_t, woff, lds, scr_wave, Breg, out, carry, sm, sl, rec, 256 * 4, 256 * 5 + 224, tid, lane);
    NEXT_STEP();
    step_bwd_signs<1, 256, 8, 256, 0, true, 1>(wsrc, (const char *)packed_t, woff, lds, scr_wave, Breg, out, carry, sm, sl, rec, 256 * 3, 256 * 4 + 224, tid, lane);
    NEXT_STEP();
    step_bwd_signs<1, 256, 8, 256, 0, true, 1>(wsrc, (const char *)packed_t, woff, lds, scr_wave, Breg, out, carry, sm, sl, rec, 256 * 2, 256 * 3 + 224, tid, lane);
    NEXT_STEP();
    step_bwd_signs<1, 256, 8, 256, 0, true, 1>(wsrc, (const char *)packed_t, woff, lds, scr_wave, Breg, out, carry, sm, sl, rec, 256 * 1, 256 * 2 + 224, tid, lane);
    NEXT_STEP();
    step_bwd_signs<1, 256, 8, 0, 0, true, 1>(wsrc, (const char *)packed_t, woff, lds, scr_wave, Breg, out, carry, sm, sl, rec, 0, 256 * 1 + 224, tid, lane);
#undef NEXT_STEP
    {   // the last tile of the last step: nothing left to hide it behind
        f32x16 a = out[7];
        sm.arrived();
#pragma unroll
        for (int r = 0; r < 16; ++r) a[r] = sign_mask(a[r], sm.of(r));
        rec.store(224, a);
    }
    return;
#else
    ActLoader al;
    al.acts = acts;
    al.ld = (size_t)ld;
    al.begin_tile(blockIdx.x, wave, j, h);
    rec.begin_tile(blockIdx.x, wave, j, h);
    f32x16 act_pend = {}, act_first = {};
    // d1: d(feat) = W_dir[:, :256]^T d(dir hidden) + W_alpha^T d(raw)[3]; fc_feat has no activation
    step_bwd_early<11, 160, 8, 256, 0, false, 0, true>(wsrc, woff, lds, scr_wave, Breg, out, carry, act_pend, act_first, al, rec, ROW_FEAT, 0, 256 * 7, tid, lane);
    NEXT_STEP();
    // d2..d9: d(h_l) = W_{l+1}^T d(h_{l+1}) * relu'(h_l),  l = 7..0 (straight-line: one step's tiles ARE the next one's B operands)
    step_bwd_early<1, 256, 8, 256, 0, true, 2, true>(wsrc, woff, lds, scr_wave, Breg, out, carry, act_pend, act_first, al, rec, 256 * 7, ROW_FEAT + 224, 256 * 6, tid, lane);
    NEXT_STEP();
    step_bwd_early<1, 256, 8, 256, 0, true, 1, true>(wsrc, woff, lds, scr_wave, Breg, out, carry, act_pend, act_first, al, rec, 256 * 6, 256 * 7 + 224, 256 * 5, tid, lane);
    NEXT_STEP();
    step_bwd_early<1, 256, 8, 256, 0, true, 1, true>(wsrc, woff, lds, scr_wave, Breg, out, carry, act_pend, act_first, al, rec, 256 * 5, 256 * 6 + 224, 256 * 4, tid, lane);
    NEXT_STEP();
    step_bwd_early<1, 256, 8, 256, 0, true, 1, true>(wsrc, woff, lds, scr_wave, Breg, out, carry, act_pend, act_first, al, rec, 256 * 4, 256 * 5 + 224, 256 * 3, tid, lane);
    NEXT_STEP();
    step_bwd_early<1, 256, 8, 256, 0, true, 1, true>(wsrc, woff, lds, scr_wave, Breg, out, carry, act_pend, act_first, al, rec, 256 * 3, 256 * 4 + 224, 256 * 2, tid, lane);
    NEXT_STEP();
    step_bwd_early<1, 256, 8, 256, 0, true, 1, true>(wsrc, woff, lds, scr_wave, Breg, out, carry, act_pend, act_first, al, rec, 256 * 2, 256 * 3 + 224, 256 * 1, tid, lane);
    NEXT_STEP();
    step_bwd_early<1, 256, 8, 256, 0, true, 1, true>(wsrc, woff, lds, scr_wave, Breg, out, carry, act_pend, act_first, al, rec, 256 * 1, 256 * 2 + 224, 0, tid, lane);
    NEXT_STEP();
    step_bwd_early<1, 256, 8, 0, 0, true, 1, false>(wsrc, woff, lds, scr_wave, Breg, out, carry, act_pend, act_first, al, rec, 0, 256 * 1 + 224, 0, tid, lane);
#undef NEXT_STEP
    // the last tile of the last step: nothing left to hide it behind
    {
        f32x16 a = out[7];
#pragma unroll
        for (int r = 0; r < 16; ++r) a[r] = tile_positive(act_pend[r]) ? a[r] : 0.0f;
        rec.store(224, a);
    }
#endif
}

DDN_EXPORT int F32_NAME(ddnerf_mlp_f32_backward_data)(const float *g_raw, const float *packed_t, const float *acts F32_SIGNS_CPARAM,
                                            int depth_head, float *deltas, long M, long ld, ddnerf_stream_t stream) {
    DDN_REQUIRE(g_raw && packed_t && acts && deltas, DDNERF_E_ARG);
#if defined(F32_REC) && F32_REC == 3
    DDN_REQUIRE(signs && ddn_aligned(signs, 128), DDNERF_E_ARG);
#endif
    DDN_REQUIRE(M > 0, DDNERF_E_ARG);
    DDN_REQUIRE(ld >= M && ld % 128 == 0, DDNERF_E_RANGE);
#ifndef F32_REC
    DDN_REQUIRE(ld <= (1l << 27), DDNERF_E_RANGE);  // (the fp32 matrices' 32-bit store offsets: 20 ld bytes)
#endif
    DDN_REQUIRE(ddn_aligned(packed_t, 16), DDNERF_E_ALIGN);
    const unsigned packed_bytes = 4u * (unsigned)make_plan_t(depth_head).total;
    dim3 grid((unsigned)((M + 127) / 128));
    if (depth_head)
        hipLaunchKernelGGL(F32_NAME(mlp_f32_bwd_data_kernel)<true>, grid, dim3(256), 0, (hipStream_t)stream, g_raw, packed_t, packed_bytes, acts,
                           deltas, M, ld F32_SIGNS_ARG);
    else
        hipLaunchKernelGGL(F32_NAME(mlp_f32_bwd_data_kernel)<false>, grid, dim3(256), 0, (hipStream_t)stream, g_raw, packed_t, packed_bytes, acts,
                           deltas, M, ld F32_SIGNS_ARG);
    return ddn_launch_status();
}
