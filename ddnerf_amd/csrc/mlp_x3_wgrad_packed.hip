// K2b'' weight gradients of the "x3" training tier on PRE-SPLIT operands.
//
// Same contraction and the same three-MFMA product as mlp_x3_wgrad.hip (dW[out][in] = sum_s delta^T[out][s] act^T[in][s],
// a*b = lo_a hi_b + hi_a lo_b + hi_a hi_b on v_mfma_f32_32x32x16_bf16), but the operands arrive as the words the x3 training
// kernels now record: every element of `deltas` / `acts` is one 32-bit word  (bf16 hi << 16) | bf16 lo,  hi = bf16(x),
// lo = bf16(x - hi) -- the split those kernels compute anyway for their own next layer.  That removes what bounded the
// fp32-operand kernel (one 32-sample tile in flight per workgroup, held in registers, split in VALU, parked in LDS: 3.7 TB/s):
//   * the words go global -> LDS by LDS-DMA (global_load_lds_dwordx4), no registers, no split, no park pass;
//   * LDS is a ring of 16-sample slots (one k = 16 MFMA step of all NOP + NIN rows, 64 B per row); while slot u feeds the
//     MFMAs the transfers of slots u+1 .. u+NS-2 are in flight and slot u+NS-1 is issued: one bare s_barrier per slot;
//   * a fragment is two ds_read_b128 (8 words of a row) and 8 v_perm_b32 (hi pairs, lo pairs).
// LDS layout of a slot: row r at 64 r, its four 16-byte chunks XOR-swizzled with (r >> 2) & 3 -- the 16 lanes of a b128 read
// phase (16 consecutive rows, same chunk) then cover all 64 banks.  A transfer instruction moves 16 rows x 64 B; the lane ->
// LDS position map of LDS-DMA is fixed (lane * 16 B), so the swizzle is applied to the SOURCE chunk each lane fetches.
// The two 64-byte halves of a 128-byte HBM line are fetched by consecutive slots, a slot period apart: the second is an L2 hit.
//
// Split-K partition, MFMA order and the slab reduction are those of mlp_x3_wgrad.hip, so for operands that are the exact
// splits of fp32 matrices the weight gradients are bit-identical to that kernel's; the bias sums add hi + lo per element
// (relative 2^-17 per term instead of exact).
#include "mlp_f32_common.h"
#include "wgrad_reduce.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define XP_ROWS 2560                // rows of a record (X3_ACT_ROWS, mlp_x3_common.h)
#define XP_STEP (XP_ROWS * 64)      // bytes from one 16-sample block to the next
#define XP_TILE 32  // samples per split-K granule (two ring slots), as in mlp_x3_wgrad.hip

__device__ __forceinline__ unsigned xp_lds_addr(const void *p) {
    return (unsigned)(size_t)(const __attribute__((address_space(3))) char *)p;
}
// 8 packed words (k = 0..7 of one row) -> the bf16x8 of their hi halves and of their lo halves
__device__ __forceinline__ void xp_unpack(const u32x4 w0, const u32x4 w1, bf16x8 &hi, bf16x8 &lo) {
    const u32x4 h = {__builtin_amdgcn_perm(w0.y, w0.x, 0x07060302u), __builtin_amdgcn_perm(w0.w, w0.z, 0x07060302u),
                     __builtin_amdgcn_perm(w1.y, w1.x, 0x07060302u), __builtin_amdgcn_perm(w1.w, w1.z, 0x07060302u)};
    const u32x4 l = {__builtin_amdgcn_perm(w0.y, w0.x, 0x05040100u), __builtin_amdgcn_perm(w0.w, w0.z, 0x05040100u),
                     __builtin_amdgcn_perm(w1.y, w1.x, 0x05040100u), __builtin_amdgcn_perm(w1.w, w1.z, 0x05040100u)};
    hi = __builtin_bit_cast(bf16x8, h);
    lo = __builtin_bit_cast(bf16x8, l);
}
// 8 fp32 VALUES (k = 0..7 of one row of a blocked fp32 record) -> the same two fragments: hi = bf16(x), lo = bf16(x - hi), round to nearest
// even -- the split the record-writing kernels computed before round 5 (mlp_f32_train.hip, rec_word2), instruction for instruction
typedef float xp_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 xp_bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void xp_split2(unsigned x0, unsigned x1, unsigned &h01, unsigned &l01) {
    const float f0 = __builtin_bit_cast(float, x0), f1 = __builtin_bit_cast(float, x1);
    h01 = __builtin_bit_cast(unsigned, __builtin_convertvector((xp_f32x2){f0, f1}, xp_bf16x2));
    const float hf0 = __builtin_bit_cast(float, h01 << 16), hf1 = __builtin_bit_cast(float, h01 & 0xffff0000u);
    l01 = __builtin_bit_cast(unsigned, __builtin_convertvector((xp_f32x2){f0 - hf0, f1 - hf1}, xp_bf16x2));
}
__device__ __forceinline__ void xp_split(const u32x4 w0, const u32x4 w1, bf16x8 &hi, bf16x8 &lo) {
    unsigned h0, h1, h2, h3, l0, l1, l2, l3;
    xp_split2(w0.x, w0.y, h0, l0);
    xp_split2(w0.z, w0.w, h1, l1);
    xp_split2(w1.x, w1.y, h2, l2);
    xp_split2(w1.z, w1.w, h3, l3);
    hi = __builtin_bit_cast(bf16x8, (u32x4){h0, h1, h2, h3});
    lo = __builtin_bit_cast(bf16x8, (u32x4){l0, l1, l2, l3});
}
__device__ __forceinline__ float xp_value(unsigned w) {
    return __builtin_bit_cast(float, w & 0xffff0000u) + __builtin_bit_cast(float, w << 16);
}
// (by value: __builtin_bit_cast applied to a vector ELEMENT expression reads element 0 of the vector with this compiler)
__device__ __forceinline__ float xp_f32(unsigned w) { return __builtin_bit_cast(float, w); }
__device__ __forceinline__ float xp_hi16(unsigned w) { return __builtin_bit_cast(float, w & 0xffff0000u); }
__device__ __forceinline__ float xp_lo16(unsigned w) { return __builtin_bit_cast(float, w << 16); }

#ifndef XP_NS_BIG
#define XP_NS_BIG 5
#endif
template <int ROWS> constexpr int xp_nslot() { return ROWS * 64 * 8 <= 160 * 1024 ? 8 : (ROWS * 64 * XP_NS_BIG <= 160 * 1024 ? XP_NS_BIG : 4); }

// NINA > 0: the activation rows come from TWO row ranges of the record, the first NINA from aT, the others from aT2 (layers_xyz.5
// contracts its deltas over cat(xyz, h4): rows 2432.. and rows 1024..; one pass over the deltas instead of two jobs)
// PAIRS: the operands are records of bf16 ROW PAIRS (the x3 training tier, include/ddnerf_hip.h: ddnerf_mlp_x3_wgrad_pairs): one
// 32-bit word holds bf16(row 2p) | bf16(row 2p + 1) << 16 of one sample, a record row is a row PAIR; every product is ONE MFMA.
// Everything is the same kernel on half as many LDS rows (a lane's row is the low or the high half of its pair's words: one
// v_perm_b32 selector per lane); a bias item sums both rows of its pair.
// RAW (round 5): the operands are blocked records of fp32 VALUES (mlp_f32_train_recf.hip): same layout, same transfers.  The hi / lo split
// happens here, ONCE per value: when slot u + 1 has landed, the workgroup's threads convert it in place -- an item is one row's 8 samples
// of a k-half (two 16-byte chunks of raw fp32 -> the chunk of their 8 bf16 hi parts and the chunk of their 8 lo parts, 24 vector-ALU
// instructions, the split of mlp_f32_train.hip's rec_word2), the bias sums ride on the raw values -- while slot u feeds the MFMAs; a
// fragment is then two ds_read_b128 and NO permute.  Per thread and slot: 2 items = 48 instructions (the word records' 48 permutes + 28 for
// the bias sums); one slot less of transfers in flight (a slot must land an iteration earlier).  Splitting per fragment instead (each of a
// row's 2 - 4 reading waves its own: 144 instructions per thread and slot) measured 330 against the word kernel's 279 us.
template <int RT, int CT, int RG, int CG, bool BIAS, int NINA = 0, bool PAIRS = false, bool RAW = false>
__global__ __launch_bounds__(64 * RG * CG) void wgrad_x3p_kernel(const unsigned *__restrict__ dT, const unsigned *__restrict__ aT,
                                                                  const unsigned *__restrict__ aT2, long M, long ld, int tiles_per_wg,
                                                                  float *__restrict__ slabs, float *__restrict__ bias_slabs) {
    constexpr int RPW = PAIRS ? 2 : 1;  // operand rows per LDS row
    constexpr int NOP = 32 * RT * RG, NIN = 32 * CT * CG, NW = RG * CG, THREADS = 64 * NW, ROWS = (NOP + NIN) / RPW;
    constexpr int NOPL = NOP / RPW, NINAL = NINA / RPW, TILE_LR = 32 / RPW;  // LDS rows of the deltas / of the first range / of a 32-row tile
    constexpr int NI = ROWS / 16, IPW = (NI + NW - 1) / NW;  // transfer instructions per slot / per wave (surplus ones repeat the last)
    constexpr int SLOT = ROWS * 64, NS = xp_nslot<ROWS>();
    constexpr int STEP_BYTES = XP_STEP / RPW;                // bytes from one 16-sample block of the record to the next
    constexpr int NBI = NOPL * 4, MAXB = (NBI + THREADS - 1) / THREADS;  // bias items (LDS row, chunk) per slot / per thread
    static_assert(NOPL % 16 == 0 && NINAL % 16 == 0 && ROWS % 16 == 0, "whole 16-row transfers");
    static_assert((NS - 2) * IPW <= 63, "vmcnt range");
    constexpr int NITEM = ROWS * 2, IPT = (NITEM + THREADS - 1) / THREADS;  // RAW: conversion items (LDS row, k-half) per slot / per thread
    static_assert(!(RAW && PAIRS) && NS >= 4, "");
    extern __shared__ __attribute__((aligned(16))) char lds_raw[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, h = lane >> 5;
    const int rg = wave % RG, cg = wave / RG;
    const long tile0 = (long)blockIdx.x * tiles_per_wg;
    const long ntiles_total = (M + XP_TILE - 1) / XP_TILE;
    const int ntiles = (int)max(0L, min((long)tiles_per_wg, ntiles_total - tile0));
    const int nht = 2 * ntiles;

    f32x16 acc[RT][CT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[rt][ct][r] = 0.0f;
    float bsum[MAXB][RPW];
#pragma unroll
    for (int r = 0; r < MAXB; ++r)
#pragma unroll
        for (int e = 0; e < RPW; ++e) bsum[r][e] = 0.0f;

    // transfer i of this wave: rows 16k .. 16k+15 of the slot, k = wave + NW i (clamped: a surplus transfer repeats the last
    // one, same data to the same place, so that every wave issues IPW per slot and one vmcnt value certifies a slot)
    unsigned voff[IPW], ldst[IPW];
    const char *src[IPW];
    const unsigned lds0 = xp_lds_addr(lds_raw);
#pragma unroll
    for (int t = 0; t < IPW; ++t) {
        const int k = min(wave + NW * t, NI - 1);                      // (uniform)
        const int row = 16 * k + (lane >> 2);
        const int q = (lane & 3) ^ ((lane >> 4) & 3);                  // source chunk of LDS chunk (lane & 3): (row >> 2) & 3 swizzle
        const bool in_d = 16 * k < NOPL;                               // (uniform; NOPL is a multiple of 16)
        const bool in_a2 = NINA > 0 && 16 * k >= NOPL + NINAL;         // (uniform; NINAL is a multiple of 16)
        voff[t] = (unsigned)((in_d ? row : (in_a2 ? row - NOPL - NINAL : row - NOPL)) * 64 + q * 16);
        src[t] = (const char *)(in_d ? dT : (in_a2 ? aT2 : aT)) + (size_t)tile0 * 2 * STEP_BYTES;
        ldst[t] = lds0 + k * 1024;
    }
    // ragged tail: ld is a multiple of 32 >= M; the pad columns' words are masked below, whatever they hold
    auto issue = [&](int u, int ring_pos) {
        const int uc = min(u, nht - 1);  // past the end: harmless repeats into a slot nobody reads again
        const unsigned slot = (unsigned)ring_pos * SLOT;
#pragma unroll
        for (int t = 0; t < IPW; ++t)
            asm volatile("s_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff[t]), "s"(src[t] + (size_t)uc * STEP_BYTES), "{m0}"(ldst[t] + slot)
                         : "memory");
    };
    const int lr = PAIRS ? (i >> 1) : i;  // this lane's LDS row inside a 32-row tile
    const int x = (lr >> 2) & 3;
    const unsigned o0 = lr * 64 + (((2 * h) ^ x) * 16), o1 = o0 ^ 16;
    const unsigned half_sel = (i & 1) ? 0x07060302u : 0x05040100u;  // PAIRS: odd rows sit in the high halves of the pair's words
    (void)half_sel;

    if (nht > 0) {
#pragma unroll
        for (int u = 0; u < NS - 1; ++u) issue(u, u);
    }
    // RAW: the in-place conversion of a landed slot.  Item (row, hf) -> LDS positions co (samples 8 hf .. + 3) and co ^ 16 (+ 4 .. + 7)
    unsigned co[IPT];
    float csum[IPT];
#pragma unroll
    for (int k = 0; k < IPT; ++k) {
        const int item = k * THREADS + tid, row = item >> 1, hf = item & 1;
        co[k] = (unsigned)(row * 64 + (((2 * hf) ^ ((row >> 2) & 3)) * 16));
        csum[k] = 0.0f;
    }
    auto nv_of = [&](int u) {   // samples of slot u that exist (uniform)
        const long nv_l = M - (tile0 * 2 + u) * 16;
        return nv_l >= 16 ? 16 : (int)(nv_l > 0 ? nv_l : 0);
    };
    auto convert = [&](char *sbw, int nv) {
#pragma unroll
        for (int k = 0; k < IPT; ++k) {
            const int item = k * THREADS + tid, hf = item & 1;
            if ((k + 1) * THREADS <= NITEM || item < NITEM) {
                u32x4 w0 = *(const u32x4 *)(sbw + co[k]), w1 = *(const u32x4 *)(sbw + (co[k] ^ 16u));
                if (nv < 16) {   // (a record's pad columns may hold anything, NaN included: zero in BOTH operands)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (8 * hf + e >= nv) w0[e] = 0u;
                        if (8 * hf + 4 + e >= nv) w1[e] = 0u;
                    }
                }
                if (BIAS && ((k + 1) * THREADS <= 2 * NOPL || item < 2 * NOPL))
                    csum[k] += ((xp_f32(w0.x) + xp_f32(w0.y)) + (xp_f32(w0.z) + xp_f32(w0.w))) + ((xp_f32(w1.x) + xp_f32(w1.y)) + (xp_f32(w1.z) + xp_f32(w1.w)));
                bf16x8 hi, lo;
                xp_split(w0, w1, hi, lo);
                *(bf16x8 *)(sbw + co[k]) = hi;
                *(bf16x8 *)(sbw + (co[k] ^ 16u)) = lo;
            }
        }
    };
    if (RAW && nht > 0) {   // slot 0: landed, converted; the loop's first barrier publishes it
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * IPW) : "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        convert(lds_raw, nv_of(0));
    }
    int rd = 0, wr = NS - 1;  // ring positions of slot u and of slot u + NS - 1
    for (int u = 0; u < nht; ++u) {
        // slot u has landed (this wave's share: all but the transfers of the NS-2 slots issued after it), for every wave; and
        // every wave is done with slot u-1, whose buffer the next transfers overwrite
        // (RAW: slot u + 1 has landed, slot u is converted -- this thread's LDS writes are out --, every wave is done with slot u - 1)
        if constexpr (RAW) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((NS - 3) * IPW) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * IPW) : "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        issue(u + NS - 1, wr);
        const char *sb = lds_raw + (size_t)rd * SLOT;
        rd = rd + 1 == NS ? 0 : rd + 1;
        wr = wr + 1 == NS ? 0 : wr + 1;
        if constexpr (RAW) {
            bf16x8 ah[RT], al[RT];
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                const char *p = sb + (rg * RT + rt) * TILE_LR * 64;
                ah[rt] = *(const bf16x8 *)(p + o0);
                al[rt] = *(const bf16x8 *)(p + o1);
            }
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                const char *p = sb + (NOPL + (cg * CT + ct) * TILE_LR) * 64;
                const bf16x8 bh = *(const bf16x8 *)(p + o0), bl = *(const bf16x8 *)(p + o1);
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {  // small terms first
                    acc[rt][ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[rt], bh, acc[rt][ct], 0, 0, 0);
                    acc[rt][ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[rt], bl, acc[rt][ct], 0, 0, 0);
                    acc[rt][ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[rt], bh, acc[rt][ct], 0, 0, 0);
                }
            }
            if (u + 1 < nht) convert(lds_raw + (size_t)rd * SLOT, nv_of(u + 1));   // (rd: already the ring position of slot u + 1)
            continue;
        }
        // samples of this slot that exist: 16, except in the last granule of a ragged M -- there the words of the samples >= M are
        // replaced by zeros in BOTH operands (a record's pad columns may hold anything, NaN included: 0 x NaN is NaN)
        const long nv_l = M - (tile0 * 2 + u) * 16;
        const int nv = nv_l >= 16 ? 16 : (int)(nv_l > 0 ? nv_l : 0);  // (uniform)
        const bool ragged = nv < 16;
        auto frag = [&](const char *p, bf16x8 &fh, bf16x8 &fl) {
            u32x4 w0 = *(const u32x4 *)(p + o0), w1 = *(const u32x4 *)(p + o1);  // samples 8 h + 0..3 and 8 h + 4..7 of the slot
            if (ragged) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (8 * h + e >= nv) w0[e] = 0u;
                    if (8 * h + 4 + e >= nv) w1[e] = 0u;
                }
            }
            if constexpr (PAIRS) {  // this row's half of the eight words: k = 0..7 of the fragment
                const u32x4 f = {__builtin_amdgcn_perm(w0.y, w0.x, half_sel), __builtin_amdgcn_perm(w0.w, w0.z, half_sel),
                                 __builtin_amdgcn_perm(w1.y, w1.x, half_sel), __builtin_amdgcn_perm(w1.w, w1.z, half_sel)};
                fh = __builtin_bit_cast(bf16x8, f);
            } else {
                xp_unpack(w0, w1, fh, fl);
            }
        };
        bf16x8 ah[RT], al[RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) frag(sb + (rg * RT + rt) * TILE_LR * 64, ah[rt], al[rt]);
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            bf16x8 bh, bl;
            frag(sb + (NOPL + (cg * CT + ct) * TILE_LR) * 64, bh, bl);
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                if constexpr (PAIRS) {
                    acc[rt][ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[rt], bh, acc[rt][ct], 0, 0, 0);
                } else {  // small terms first
                    acc[rt][ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[rt], bh, acc[rt][ct], 0, 0, 0);
                    acc[rt][ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[rt], bl, acc[rt][ct], 0, 0, 0);
                    acc[rt][ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[rt], bh, acc[rt][ct], 0, 0, 0);
                }
            }
        }
        if (BIAS) {
#pragma unroll
            for (int r = 0; r < MAXB; ++r) {
                const int idx = r * THREADS + tid;
                if ((r + 1) * THREADS <= NBI || idx < NBI) {
                    u32x4 w = *(const u32x4 *)(sb + idx * 16);
                    if (ragged) {  // LDS chunk (idx & 3) of row idx >> 2 holds source chunk q: samples 4 q .. 4 q + 3 of the slot
                        const int q = (idx & 3) ^ ((idx >> 4) & 3);
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (4 * q + e >= nv) w[e] = 0u;
                    }
                    if constexpr (PAIRS) {  // low halves: the even row, high halves: the odd row
                        bsum[r][0] += (xp_lo16(w.x) + xp_lo16(w.y)) + (xp_lo16(w.z) + xp_lo16(w.w));
                        bsum[r][1] += (xp_hi16(w.x) + xp_hi16(w.y)) + (xp_hi16(w.z) + xp_hi16(w.w));
                    } else {
                        bsum[r][0] += (xp_value(w.x) + xp_value(w.y)) + (xp_value(w.z) + xp_value(w.w));
                    }
                }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the surplus transfers of the last slots

    float *slab = slabs + (size_t)blockIdx.x * NOP * NIN;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                slab[(size_t)((rg * RT + rt) * 32 + tile_row(r, h)) * NIN + (cg * CT + ct) * 32 + i] = acc[rt][ct][r];
    if (BIAS && RAW) {
        // conversion item (row, k-half) is owned by the same thread in every slot; the two halves of a row sit in adjacent lanes
#pragma unroll
        for (int k = 0; k < IPT; ++k) {
            const int item = k * THREADS + tid, row = item >> 1;
            float s = csum[k];
            s += __shfl_xor(s, 1);
            if ((item & 1) == 0 && row < NOPL) bias_slabs[(size_t)blockIdx.x * NOP + row] = s;
        }
    } else if (BIAS) {
        // item (row, chunk) is owned by the same thread in every slot; the 4 chunks of a row sit in 4 adjacent lanes
#pragma unroll
        for (int r = 0; r < MAXB; ++r) {
            const int idx = r * THREADS + tid, row = idx >> 2;
#pragma unroll
            for (int e = 0; e < RPW; ++e) {
                float s = bsum[r][e];
                s += __shfl_xor(s, 1);
                s += __shfl_xor(s, 2);
                if ((idx & 3) == 0 && row < NOPL) bias_slabs[(size_t)blockIdx.x * NOP + RPW * row + e] = s;
            }
        }
    }
}

static int n_out_pad_of(int n_out) { return n_out > 128 ? 256 : (n_out > 32 ? 128 : 32); }

template <bool PAIRS, bool RAW = false>
static int wgrad_packed_impl(const void *deltas, int drow0, int n_out, const void *acts, int arow0, int n_in, int n_in_used, long M, long ld,
                             float *dst, int dst_ld, int dst_col0, float *dst_bias, float *workspace, int max_workgroups, ddnerf_stream_t stream) {
    DDN_REQUIRE(deltas && acts && dst && workspace, DDNERF_E_ARG);
    DDN_REQUIRE(M > 0 && n_out > 0 && n_out <= 256 && n_in_used > 0 && n_in_used <= n_in, DDNERF_E_ARG);
    DDN_REQUIRE(n_in == 32 || n_in == 96 || n_in == 128 || n_in == 256, DDNERF_E_RANGE);
    DDN_REQUIRE(ld % 32 == 0 && ld >= M, DDNERF_E_RANGE);  // the records hold ld / 16 whole blocks
    DDN_REQUIRE(drow0 >= 0 && arow0 >= 0 && drow0 + n_out_pad_of(n_out) <= XP_ROWS && arow0 + n_in <= XP_ROWS, DDNERF_E_RANGE);
    DDN_REQUIRE(!PAIRS || (drow0 % 2 == 0 && arow0 % 2 == 0), DDNERF_E_RANGE);  // a record row holds a row PAIR
    DDN_REQUIRE(ddn_aligned(deltas, 16) && ddn_aligned(acts, 16), DDNERF_E_ALIGN);
    hipStream_t st = (hipStream_t)stream;
    constexpr int RPW = PAIRS ? 2 : 1;
    const int n_out_pad = n_out_pad_of(n_out);
    const long ntiles = (M + XP_TILE - 1) / XP_TILE;
    // split-K width: 256 workgroups fill the chip with ONE job; two jobs side by side on two streams (max_workgroups = 128)
    // write and reduce half the slabs each and overlap each other's prologue / slab epilogue
    const int cap = max_workgroups > 0 && max_workgroups < 256 ? max_workgroups : 256;
    const int nwg = (int)(ntiles < cap ? ntiles : cap);
    const int tiles_per_wg = (int)((ntiles + nwg - 1) / nwg);
    const unsigned *dT = (const unsigned *)deltas + (size_t)(drow0 / RPW) * 16, *aT = (const unsigned *)acts + (size_t)(arow0 / RPW) * 16;
    float *slabs = workspace;
    const size_t slab_stride = (size_t)n_out_pad * n_in;
    float *bias_slabs = dst_bias ? workspace + (size_t)nwg * slab_stride : nullptr;
#define LAUNCH(RT, CT, RG, CG)                                                                                          \
    do {                                                                                                                \
        constexpr int ROWS_ = (32 * RT * RG + 32 * CT * CG) / RPW;                                                      \
        const size_t lds = (size_t)ROWS_ * 64 * xp_nslot<ROWS_>();                                                      \
        if (bias_slabs)                                                                                                 \
            hipLaunchKernelGGL((wgrad_x3p_kernel<RT, CT, RG, CG, true, 0, PAIRS, RAW>), dim3(nwg), dim3(64 * RG * CG), lds, st, dT, aT, aT, M, ld, \
                               tiles_per_wg, slabs, bias_slabs);                                                        \
        else                                                                                                            \
            hipLaunchKernelGGL((wgrad_x3p_kernel<RT, CT, RG, CG, false, 0, PAIRS, RAW>), dim3(nwg), dim3(64 * RG * CG), lds, st, dT, aT, aT, M, ld, \
                               tiles_per_wg, slabs, bias_slabs);                                                        \
    } while (0)
    if (n_out_pad == 256) {
        if (n_in == 256) LAUNCH(2, 4, 4, 2); else if (n_in == 128) LAUNCH(1, 4, 8, 1); else if (n_in == 96) LAUNCH(1, 3, 8, 1); else LAUNCH(1, 1, 8, 1);
    } else if (n_out_pad == 128) {
        if (n_in == 256) LAUNCH(1, 4, 4, 2); else if (n_in == 128) LAUNCH(1, 2, 4, 2); else if (n_in == 96) LAUNCH(1, 3, 4, 1); else LAUNCH(1, 1, 4, 1);
    } else {
        if (n_in == 256) LAUNCH(1, 2, 1, 4); else if (n_in == 128) LAUNCH(1, 1, 1, 4); else if (n_in == 96) LAUNCH(1, 3, 1, 1); else LAUNCH(1, 1, 1, 1);
    }
#undef LAUNCH
    const int total = n_out * n_in_used, nb_w = (total + 63) / 64, nb_b = dst_bias ? (n_out + 63) / 64 : 0;
    const WgradReduceJob jw = {slabs, slab_stride, n_in, n_out, n_in_used, dst_ld, dst_col0, dst};
    const WgradReduceJob jb = {bias_slabs, (size_t)n_out_pad, 1, n_out, 1, 1, 0, dst_bias};
    hipLaunchKernelGGL(wgrad_reduce_pair_kernel, dim3(nb_w + nb_b), dim3(256), 0, st, jw, jb, nb_w, nwg);
    return ddn_launch_status();
}

template <bool PAIRS, bool RAW = false>
static int wgrad_packed_skip_impl(const void *deltas, int drow0, const void *acts, int arow_a, int arow_b, long M, long ld, float *dst,
                                  float *dst_bias, float *workspace, int max_workgroups, ddnerf_stream_t stream) {
    DDN_REQUIRE(deltas && acts && dst && dst_bias && workspace, DDNERF_E_ARG);
    DDN_REQUIRE(M > 0 && ld % 32 == 0 && ld >= M, DDNERF_E_RANGE);
    DDN_REQUIRE(drow0 >= 0 && arow_a >= 0 && arow_b >= 0 && drow0 + 256 <= XP_ROWS && arow_a + 96 <= XP_ROWS && arow_b + 256 <= XP_ROWS,
                DDNERF_E_RANGE);
    DDN_REQUIRE(!PAIRS || (drow0 % 2 == 0 && arow_a % 2 == 0 && arow_b % 2 == 0), DDNERF_E_RANGE);
    DDN_REQUIRE(ddn_aligned(deltas, 16) && ddn_aligned(acts, 16), DDNERF_E_ALIGN);
    hipStream_t st = (hipStream_t)stream;
    constexpr int RPW = PAIRS ? 2 : 1;
    const long ntiles = (M + XP_TILE - 1) / XP_TILE;
    const int cap = max_workgroups > 0 && max_workgroups < 256 ? max_workgroups : 256;
    const int nwg = (int)(ntiles < cap ? ntiles : cap);
    const int tiles_per_wg = (int)((ntiles + nwg - 1) / nwg);
    const unsigned *dT = (const unsigned *)deltas + (size_t)(drow0 / RPW) * 16;
    const unsigned *aA = (const unsigned *)acts + (size_t)(arow_a / RPW) * 16, *aB = (const unsigned *)acts + (size_t)(arow_b / RPW) * 16;
    constexpr int NIN = 352, ROWS = (256 + NIN) / RPW;
    float *slabs = workspace, *bias_slabs = workspace + (size_t)nwg * 256 * NIN;
    hipLaunchKernelGGL((wgrad_x3p_kernel<1, 11, 8, 1, true, 96, PAIRS, RAW>), dim3(nwg), dim3(512), (size_t)ROWS * 64 * xp_nslot<ROWS>(), st, dT, aA, aB, M,
                       ld, tiles_per_wg, slabs, bias_slabs);
    const int total = 256 * NIN, nb_w = (total + 63) / 64, nb_b = (256 + 63) / 64;
    const WgradReduceJob jw = {slabs, (size_t)256 * NIN, NIN, 256, NIN, NIN, 0, dst};
    const WgradReduceJob jb = {bias_slabs, (size_t)256, 1, 256, 1, 1, 0, dst_bias};
    hipLaunchKernelGGL(wgrad_reduce_pair_kernel, dim3(nb_w + nb_b), dim3(256), 0, st, jw, jb, nb_w, nwg);
    return ddn_launch_status();
}

// Same contract as ddnerf_mlp_x3_wgrad (mlp_x3_wgrad.hip) on records: `deltas` and `acts` are blocked hi/lo-word records of 2560 rows
// x ld samples (ld / 16 blocks), as the record-writing build of the fp32 training kernels writes them and as ddnerf_mlp_x3_split
// produces them from fp32 matrices.
DDN_EXPORT int ddnerf_mlp_x3_wgrad_packed(const void *deltas, int drow0, int n_out, const void *acts, int arow0, int n_in,
                                          int n_in_used, long M, long ld, float *dst, int dst_ld, int dst_col0,
                                          float *dst_bias, float *workspace, int max_workgroups, ddnerf_stream_t stream) {
    return wgrad_packed_impl<false>(deltas, drow0, n_out, acts, arow0, n_in, n_in_used, M, ld, dst, dst_ld, dst_col0, dst_bias, workspace,
                                    max_workgroups, stream);
}

// layers_xyz.5 in ONE job: dst[r * 352 + c] for the 256 rows drow0.. of `deltas` against cat(acts rows arow_a .. +96, acts rows
// arow_b .. +256) -- the column order of the reference's cat(xyz, h) input -- and the bias sums.
DDN_EXPORT int ddnerf_mlp_x3_wgrad_packed_skip(const void *deltas, int drow0, const void *acts, int arow_a, int arow_b, long M, long ld,
                                               float *dst, float *dst_bias, float *workspace, int max_workgroups,
                                               ddnerf_stream_t stream) {
    return wgrad_packed_skip_impl<false>(deltas, drow0, acts, arow_a, arow_b, M, ld, dst, dst_bias, workspace, max_workgroups, stream);
}

// The same two jobs on blocked records of fp32 VALUES (ddnerf_mlp_f32_forward_train_recf / _backward_data_recf write them): word index
// ((m >> 4) * 2560 + row) * 16 + (m & 15) holds the value itself; the kernel splits it (hi = bf16(x), lo = bf16(x - hi)) per MFMA fragment.
// Weight gradients bit-identical to ddnerf_mlp_x3_wgrad_packed on the split records of the same values; bias sums add the values.
DDN_EXPORT int ddnerf_mlp_x3_wgrad_blocked(const void *deltas, int drow0, int n_out, const void *acts, int arow0, int n_in, int n_in_used,
                                           long M, long ld, float *dst, int dst_ld, int dst_col0, float *dst_bias, float *workspace,
                                           int max_workgroups, ddnerf_stream_t stream) {
    return wgrad_packed_impl<false, true>(deltas, drow0, n_out, acts, arow0, n_in, n_in_used, M, ld, dst, dst_ld, dst_col0, dst_bias, workspace,
                                          max_workgroups, stream);
}
DDN_EXPORT int ddnerf_mlp_x3_wgrad_blocked_skip(const void *deltas, int drow0, const void *acts, int arow_a, int arow_b, long M, long ld,
                                                float *dst, float *dst_bias, float *workspace, int max_workgroups, ddnerf_stream_t stream) {
    return wgrad_packed_skip_impl<false, true>(deltas, drow0, acts, arow_a, arow_b, M, ld, dst, dst_bias, workspace, max_workgroups, stream);
}

// The x3 training tier's weight gradients: the same jobs on records of bf16 ROW PAIRS (what ddnerf_mlp_x3_forward_train and
// ddnerf_mlp_x3_backward_data write): 1280 pair rows x ld samples, one MFMA per product.  Row offsets are operand rows (even).
DDN_EXPORT int ddnerf_mlp_x3_wgrad_pairs(const void *deltas, int drow0, int n_out, const void *acts, int arow0, int n_in, int n_in_used,
                                         long M, long ld, float *dst, int dst_ld, int dst_col0, float *dst_bias, float *workspace,
                                         int max_workgroups, ddnerf_stream_t stream) {
    return wgrad_packed_impl<true>(deltas, drow0, n_out, acts, arow0, n_in, n_in_used, M, ld, dst, dst_ld, dst_col0, dst_bias, workspace,
                                   max_workgroups, stream);
}
DDN_EXPORT int ddnerf_mlp_x3_wgrad_pairs_skip(const void *deltas, int drow0, const void *acts, int arow_a, int arow_b, long M, long ld,
                                              float *dst, float *dst_bias, float *workspace, int max_workgroups, ddnerf_stream_t stream) {
    return wgrad_packed_skip_impl<true>(deltas, drow0, acts, arow_a, arow_b, M, ld, dst, dst_bias, workspace, max_workgroups, stream);
}

// fp32 [rows][ld] ([feature][sample]) -> rows row0 .. row0 + rows - 1 of a blocked hi/lo-word record (mlp_x3_common.h): the
// operand format of ddnerf_mlp_x3_wgrad_packed, for callers that hold plain fp32 matrices
__global__ __launch_bounds__(256) void x3_split_kernel(const float *__restrict__ x, long ld, int row0, unsigned *__restrict__ rec) {
    const long q = (long)blockIdx.x * 256 + threadIdx.x;  // quad of samples
    const int row = blockIdx.y;
    if (4 * q >= ld) return;
    const f32x4 v = *(const f32x4 *)(x + (size_t)row * ld + 4 * q);
    u32x4 o;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const __bf16 hi = (__bf16)v[c];
        const __bf16 lo = (__bf16)(v[c] - (float)hi);
        o[c] = ((unsigned)__builtin_bit_cast(unsigned short, hi) << 16) | __builtin_bit_cast(unsigned short, lo);
    }
    *(u32x4 *)(rec + (((size_t)(q >> 2) * XP_ROWS + row0 + row) * 16 + 4 * (q & 3))) = o;
}
// fp32 [rows][ld] -> rows row0 .. of a record of bf16 ROW PAIRS: word ((m >> 4) * 1280 + (row >> 1)) * 16 + (m & 15) =
// bf16(x[row even][m]) | bf16(x[row odd][m]) << 16 (round to nearest even).  rows and row0 even.
__global__ __launch_bounds__(256) void x3_split_pairs_kernel(const float *__restrict__ x, long ld, int row0, unsigned *__restrict__ rec) {
    const long q = (long)blockIdx.x * 256 + threadIdx.x;  // quad of samples
    const int pr = blockIdx.y;                            // pair of rows
    if (4 * q >= ld) return;
    const f32x4 a = *(const f32x4 *)(x + (size_t)(2 * pr) * ld + 4 * q), b = *(const f32x4 *)(x + (size_t)(2 * pr + 1) * ld + 4 * q);
    u32x4 o;
#pragma unroll
    for (int c = 0; c < 4; ++c)
        o[c] = (unsigned)__builtin_bit_cast(unsigned short, (__bf16)a[c]) | ((unsigned)__builtin_bit_cast(unsigned short, (__bf16)b[c]) << 16);
    *(u32x4 *)(rec + (((size_t)(q >> 2) * (XP_ROWS / 2) + row0 / 2 + pr) * 16 + 4 * (q & 3))) = o;
}
DDN_EXPORT int ddnerf_mlp_x3_split_pairs(const float *x, int rows, long ld, int row0, void *record, ddnerf_stream_t stream) {
    DDN_REQUIRE(x && record, DDNERF_E_ARG);
    DDN_REQUIRE(rows > 0 && rows % 2 == 0 && row0 >= 0 && row0 % 2 == 0 && row0 + rows <= XP_ROWS && ld > 0 && ld % 16 == 0, DDNERF_E_RANGE);
    DDN_REQUIRE(ddn_aligned(x, 16) && ddn_aligned(record, 16), DDNERF_E_ALIGN);
    hipLaunchKernelGGL(x3_split_pairs_kernel, dim3((unsigned)((ld / 4 + 255) / 256), (unsigned)(rows / 2)), dim3(256), 0, (hipStream_t)stream, x,
                       ld, row0, (unsigned *)record);
    return ddn_launch_status();
}
DDN_EXPORT int ddnerf_mlp_x3_split(const float *x, int rows, long ld, int row0, void *record, ddnerf_stream_t stream) {
    DDN_REQUIRE(x && record, DDNERF_E_ARG);
    DDN_REQUIRE(rows > 0 && row0 >= 0 && row0 + rows <= XP_ROWS && ld > 0 && ld % 16 == 0, DDNERF_E_RANGE);
    DDN_REQUIRE(ddn_aligned(x, 16) && ddn_aligned(record, 16), DDNERF_E_ALIGN);
    hipLaunchKernelGGL(x3_split_kernel, dim3((unsigned)((ld / 4 + 255) / 256), (unsigned)rows), dim3(256), 0, (hipStream_t)stream, x, ld,
                       row0, (unsigned *)record);
    return ddn_launch_status();
}
