// Stage machinery of the x3 backward-data kernel (mlp_x3_train.hip), on v_mfma_f32_32x32x16_bf16: exact hi/lo bf16 operand
// splits, three MFMAs per product, transposed layers, tiles chained through registers, two LDS stage buffers.  (The forward
// kernels moved to the 16x16x32 body of mlp_mfma16.inc: mlp_x3_fwd.hip, mlp_x3_fwd_train.hip; the MODE 0 / MODE 1 paths here are
// what the backward pass shares with the forward formulation it was written beside.)
#pragma once
#include "mlp_bf16_common.h"

#define X3_STAGE_BYTES_MAX (67 * 1024)
#define X3_WG_THREADS 256
#define X3_WG_WAVES 4
#define X3_WG_SAMPLES (X3_WG_WAVES * 32)
#ifndef X3_DEPTH
#define X3_DEPTH 3  // A fragment pairs are read this many k-steps (x 96 matrix-pipe cycles) ahead of their MFMAs
#endif

// one slice = 32 rows of (K + 8) bf16 (hi image), the same 32 rows (lo image), then 32 fp32 biases
__host__ __device__ constexpr int x3_slice_bytes(int K) { return 64 * (2 * K + 16) + 128; }
__host__ __device__ constexpr int x3_round_kib(int b) { return (b + 1023) / 1024 * 1024; }
// A plan PL is a struct of constexpr tables: NL steps; per step K (inputs), NB (32-row output blocks), SPS (slices per
// LDS stage; two stage buffers of <= 67 KiB).
template <class PL>
__host__ __device__ constexpr int x3_stage_bytes(int l, int st) {
    int first = st * PL::SPS[l];
    int ns = PL::NB[l] - first < PL::SPS[l] ? PL::NB[l] - first : PL::SPS[l];
    return x3_round_kib(ns * x3_slice_bytes(PL::K[l]));
}
template <class PL>
__host__ __device__ constexpr int x3_stages_of(int l) {
    return (PL::NB[l] + PL::SPS[l] - 1) / PL::SPS[l];
}
template <class PL>
__host__ __device__ constexpr int x3_total_bytes() {
    int off = 0;
    for (int l = 0; l < PL::NL; ++l)
        for (int st = 0; st < x3_stages_of<PL>(l); ++st) off += x3_stage_bytes<PL>(l, st);
    return off;
}
template <class PL>
__host__ __device__ constexpr int x3_layer_off(int l) {
    int off = 0;
    for (int k = 0; k < l; ++k)
        for (int st = 0; st < x3_stages_of<PL>(k); ++st) off += x3_stage_bytes<PL>(k, st);
    return off;
}

// byte offset of every layer's first stage, as a compile-time table (the pack kernels index it per thread)
template <class PL>
struct X3LayerOffsets {
    int v[PL::NL + 1];
    constexpr X3LayerOffsets() : v() {
        for (int l = 0; l <= PL::NL; ++l) v[l] = x3_layer_off<PL>(l);
    }
};

// Packs one 16-bit word of a plan's image: SRCW(l, out_row, in_col) / SRCB(l, out_row) supply the fp32 values.
template <class PL, class SRCW, class SRCB>
__device__ __forceinline__ unsigned short x3_pack_word(int idx, SRCW &&srcw, SRCB &&srcb) {
    constexpr X3LayerOffsets<PL> offs{};
    int byte = idx * 2, l = PL::NL - 1;
    while (l > 0 && byte < offs.v[l]) --l;
    int rel = byte - offs.v[l], st = 0;
    // all stages of a layer but the last have the same size
    const int full = x3_round_kib(PL::SPS[l] * x3_slice_bytes(PL::K[l]));
    st = rel / full;
    rel -= st * full;
    const int K = PL::K[l], rowb = 2 * K + 16;
    const int sl = rel / x3_slice_bytes(K);
    const int first = st * PL::SPS[l];
    const int nsl = PL::NB[l] - first < PL::SPS[l] ? PL::NB[l] - first : PL::SPS[l];
    if (sl >= nsl) return 0;
    int r2 = rel - sl * x3_slice_bytes(K);
    if (r2 < 64 * rowb) {
        const int part = r2 / (32 * rowb);  // 0: hi image, 1: lo image
        r2 -= part * 32 * rowb;
        const int row = r2 / rowb, col = (r2 % rowb) / 2;
        const float v = col < K ? srcw(l, 32 * (first + sl) + row, korder(col)) : 0.0f;
        const __bf16 hi = (__bf16)v;
        const __bf16 b = part ? (__bf16)(v - (float)hi) : hi;
        return __builtin_bit_cast(unsigned short, b);
    }
    const int bi = (r2 - 64 * rowb) / 4, half = ((r2 - 64 * rowb) % 4) / 2;  // fp32 bias, two 16-bit halves
    const unsigned u = __builtin_bit_cast(unsigned, srcb(l, 32 * (first + sl) + bi));
    return (unsigned short)(half ? (u >> 16) : (u & 0xffffu));
}

__device__ __forceinline__ void x3_dma_stage(const char *__restrict__ src, char *dst, int bytes, int wave, int lane) {
    const unsigned base = lds_addr_of(dst);
    for (int off = wave * 1024; off < bytes; off += X3_WG_WAVES * 1024) dma_piece(src + off + lane * 16, base + off);
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
__device__ __forceinline__ void split_quad(const f32x4 a, const f32x4 b, bf16x8 &hi, bf16x8 &lo) {
    unsigned wh[4], wl[4];
    split_pair(a.x, a.y, wh[0], wl[0]);
    split_pair(a.z, a.w, wh[1], wl[1]);
    split_pair(b.x, b.y, wh[2], wl[2]);
    split_pair(b.z, b.w, wh[3], wl[3]);
    hi = __builtin_bit_cast(bf16x8, u32x4{wh[0], wh[1], wh[2], wh[3]});
    lo = __builtin_bit_cast(bf16x8, u32x4{wl[0], wl[1], wl[2], wl[3]});
}

// B-operand source of k-step ks.  Forward KINDs: 0 first layer (xyz, X[0..5]); 1 hidden (H[ks]); 2 skip layer (X[0..5]
// then H[0..15]); 3 dir layer (H[0..15] then X[6..7] = view dirs); 4 heads (H[0..7]).  Backward KINDs: 10 heads^T (the
// d(raw) tile, X[0..1]); 11 [dir | alpha]^T (d(dir hidden) H[0..7], then the d(raw) tile X[0..1]).
template <int KIND, int KS>
__device__ __forceinline__ const bf16x8 &bsrc(const bf16x8 (&H)[16], const bf16x8 (&X)[8]) {
    if constexpr (KIND == 0 || KIND == 10) return X[KS];
    else if constexpr (KIND == 2) {
        if constexpr (KS < 6) return X[KS];
        else return H[KS - 6];
    } else if constexpr (KIND == 3) {
        if constexpr (KS < 16) return H[KS];
        else return X[6 + (KS - 16)];
    } else if constexpr (KIND == 11) {
        if constexpr (KS < 8) return H[KS];
        else return X[KS - 8];
    } else return H[KS];
}

// What a finished tile does besides feeding the next layer:
//   MODE 0  inference: nothing
//   MODE 1  training forward: the (post-activation) tile is stored transposed into out[row][col] (fp32, non-temporal),
//           and with it one bit per value (value > 0) as a 16-bit word per lane into bits_out[(tile * 2 + h) * ld + col]
//   MODE 2  backward: the tile is multiplied by relu'(recorded activation) taken from bits_in (same layout), then stored
//           transposed into out (the layer's pre-activation gradient)
__device__ __forceinline__ constexpr int x3_tile_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

#ifdef X3_PLAIN_STORES
#define X3_STORE(v, p) (*(p) = (v))
#else
#define X3_STORE(v, p) __builtin_nontemporal_store(v, p)  // streaming data must not evict the weight images from L2
#endif

// Records of the x3 training tier (`acts` of the forward, `deltas` of the backward pass): "blocked hi/lo words".  Element
// (row, sample m) is ONE 32-bit word (bf16 hi << 16) | bf16 lo -- hi = bf16(x), lo = bf16(x - hi), the split the kernels make
// for their own next layer anyway -- at word index ((m >> 4) * X3_ACT_ROWS + row) * 16 + (m & 15): 16-sample blocks, inside a
// block the rows back to back, 64 bytes each.  The weight-gradient kernel (mlp_x3_wgrad_packed.hip) streams a job's rows of a
// block as ONE contiguous run straight into LDS; with the [row][sample] fp32 layout of the fp32 tier every row of a tile
// is in another 2 MiB page, which held that kernel's reads to 2.4 TB/s (4.7 TB/s blocked) -- and the stores here scattered
// the same way.
#define X3_ACT_ROWS 2560
__device__ __forceinline__ size_t x3_rec_index(int row, long m) { return ((size_t)(m >> 4) * X3_ACT_ROWS + row) * 16 + (m & 15); }
__device__ __forceinline__ unsigned x3_word(float x) {
    const __bf16 hi = (__bf16)x;
    const __bf16 lo = (__bf16)(x - (float)hi);
    return ((unsigned)__builtin_bit_cast(unsigned short, hi) << 16) | __builtin_bit_cast(unsigned short, lo);
}
// Addresses are split into a uniform part (the wave's 32 samples = two blocks, and the row of the tile: scalar registers)
// and a per-lane 32-bit element offset fixed for the whole kernel, so that every tile store / sign-word access is one
// saddr-form instruction -- computing the index per lane in 64-bit VALU arithmetic had doubled the kernel's VALU
// instruction count.
struct X3TileIO {
    unsigned *out;      // the record, advanced to the wave's first block
    unsigned short *bits_out;
    const unsigned short *bits_in;
    size_t ld;          // of the sign words ([tile pair rows][sample])
    unsigned lane_w;    // ((m >> 4) & 1) X3_ACT_ROWS 16 + 64 h + (m & 15): word offset of this lane inside a tile's first row group
                        // (tile rows are (r & 3) + 8 (r >> 2) + 4 h: the lane half selects rows +4)
    unsigned lane_u16;  // h ld + col: this lane's sign word inside a tile's pair of word rows
    int row0;           // row of the layer's block 0 in `out` (a multiple of 32)
};
__device__ __forceinline__ X3TileIO x3_tile_io(float *out, unsigned short *bits_out, const unsigned short *bits_in, long ld,
                                               long col, int h, int row0) {
    const int m0 = __builtin_amdgcn_readfirstlane((int)(col & ~31L));  // a wave owns 32 consecutive samples
    return X3TileIO{(unsigned *)out + (size_t)(m0 >> 4) * X3_ACT_ROWS * 16, bits_out, bits_in, (size_t)ld,
                    (unsigned)(((col >> 4) & 1) * X3_ACT_ROWS * 16 + 64 * h + (col & 15)), (unsigned)(h * ld + col), row0};
}

// One stage: NBLK 32-row slices out of LDS buffer `cur`.  The statement order IS the schedule: a 32x32x16 MFMA occupies
// the matrix pipe for 32 cycles but the wave's issue port for 4, so the rest of a k-step is dealt into the three
// 28-cycle gaps behind its three MFMAs (one sched_barrier(0) per gap): A fragment pairs are read DEPTH k-steps ahead
// into a ring; the bias tile of block b+1 is read during block b; the hi/lo re-pack of block b-1's tile is spread over
// block b's k-steps; the next stage's weight pieces travel global -> VGPR -> LDS, PFD pieces in flight.
// Hh/Hl: this layer's input files, Hnh/Hnl: the next layer's (k-steps 2(B0+b), 2(B0+b)+1 per block).
// ACT: apply the activation (forward: ReLU; backward: the recorded ReLU mask).
template <int KIND, int K, int NBLK, int B0, int NCONV, int DMA_BYTES, bool HN_AGPR, int MODE, int PFD, bool ACT>
__device__ __forceinline__ void x3_stage_compute(const char *__restrict__ cur, const bf16x8 (&Hh)[16],
                                                 const bf16x8 (&Hl)[16], const bf16x8 (&Xh)[8], const bf16x8 (&Xl)[8],
                                                 bf16x8 (&Hnh)[16], bf16x8 (&Hnl)[16], f32x16 &keep, int lane, int wave,
                                                 const char *__restrict__ dma_src, char *dma_dst, const X3TileIO &io) {
    constexpr int NKS = K / 16, P = NBLK * NKS, DEPTH = X3_DEPTH, ROWB = 2 * K + 16, SLB = x3_slice_bytes(K);
    constexpr int PIECES = DMA_BYTES / 1024, NP = (PIECES + X3_WG_WAVES - 1) / X3_WG_WAVES;  // pieces of this wave
    constexpr bool BIAS = MODE != 2;  // the backward pass starts its tiles at zero
    const int h = lane >> 5;
    const char *a_lane = cur + (lane & 31) * ROWB + 16 * h;
    const char *b_lane = cur + 64 * ROWB + 16 * h;
    bf16x8 ring_h[DEPTH], ring_l[DEPTH];
    f32x16 acc[2];
    f32x4 pf[PFD];
    unsigned bits[3];  // activation sign words of the tiles in flight (MODE 1: being built; MODE 2: fetched two blocks ahead)
    auto read_a = [&](auto pc) {
        constexpr int p = decltype(pc)::value;
        const char *src = a_lane + (p / NKS) * SLB + 32 * (p % NKS);
        ring_h[p % DEPTH] = *(const bf16x8 *)(src);
        ring_l[p % DEPTH] = *(const bf16x8 *)(src + 32 * ROWB);
    };
    auto read_bias = [&](auto bc, auto gc) {  // rows 8g + 4h + (0..3) of block b -> accumulator registers 4g..4g+3
        constexpr int b = decltype(bc)::value, g = decltype(gc)::value;
        f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
        if constexpr (BIAS) v = *(const f32x4 *)(b_lane + b * SLB + 32 * g);
        acc[b & 1][4 * g + 0] = v.x;
        acc[b & 1][4 * g + 1] = v.y;
        acc[b & 1][4 * g + 2] = v.z;
        acc[b & 1][4 * g + 3] = v.w;
    };
    const unsigned lane_off = wave * 1024 + lane * 16;  // uniform base + 32-bit lane offset: saddr-form global loads
    auto piece_ok = [&](int i) { return (i + 1) * X3_WG_WAVES <= PIECES || wave + X3_WG_WAVES * i < PIECES; };
    auto ld_piece = [&](auto ic) {
        constexpr int i = decltype(ic)::value;
#ifndef X3_NO_STAGE
        if (piece_ok(i)) pf[i % PFD] = *(const f32x4 *)(dma_src + i * (X3_WG_WAVES * 1024) + lane_off);
#endif
    };
    auto st_piece = [&](auto ic) {
        constexpr int i = decltype(ic)::value;
#ifndef X3_NO_STAGE
        if (piece_ok(i)) *(f32x4 *)(dma_dst + i * (X3_WG_WAVES * 1024) + lane_off) = pf[i % PFD];
#endif
    };
    auto tile_index = [&](int pb) { return (io.row0 >> 5) + B0 + pb; };
    auto fetch_bits = [&](auto bc) {  // MODE 2: the sign word of block b's tile (requested a whole block ahead of its use)
        constexpr int b = decltype(bc)::value;
        if constexpr (MODE == 2 && ACT)
            bits[b % 3] = __builtin_nontemporal_load(io.bits_in + (size_t)tile_index(b) * 2 * io.ld + io.lane_u16);
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
            if constexpr (ACT && MODE != 2) {  // ReLU on the bit patterns: one v_max_i32 each
                x0 = __builtin_bit_cast(float, max(__builtin_bit_cast(int, x0), 0));
                x1 = __builtin_bit_cast(float, max(__builtin_bit_cast(int, x1), 0));
            }
            if constexpr (ACT && MODE == 2) {  // x & -(bit): a sign-extended 1-bit field extract and an AND per value
                const int w = (int)bits[pb % 3];
                x0 = __builtin_bit_cast(float, __builtin_bit_cast(int, x0) & ((w << (31 - 2 * u)) >> 31));
                x1 = __builtin_bit_cast(float, __builtin_bit_cast(int, x1) & ((w << (30 - 2 * u)) >> 31));
            }
            if constexpr (MODE == 1) {
                // value > 0 <=> its bit pattern >= 1 (the values are >= +0 after the ReLU; layers without one never read
                // their sign words): min(bits, 1) is the sign bit, one shift-or files it
                const unsigned p0 = min(__builtin_bit_cast(unsigned, x0), 1u), p1 = min(__builtin_bit_cast(unsigned, x1), 1u);
                bits[pb % 3] = u == 0 ? (p0 | (p1 << 1)) : (bits[pb % 3] | (p0 << (2 * u)) | (p1 << (2 * u + 1)));
                if constexpr (u == 7)
                    __builtin_nontemporal_store((unsigned short)bits[pb % 3],
                                                io.bits_out + (size_t)tile_index(pb) * 2 * io.ld + io.lane_u16);
            }
            unsigned hw, lw;
            split_pair(x0, x1, hw, lw);
            if constexpr (MODE != 0) {
                // the record keeps the split: word = (hi << 16) | lo, one v_perm_b32 per value.  Uniform tile base + one of 16
                // uniform row offsets (common subexpressions across all tiles) + the lane part
                unsigned *tile = io.out + (size_t)(io.row0 + 32 * (B0 + pb)) * 16;
                X3_STORE(__builtin_amdgcn_perm(hw, lw, 0x05040100u), tile + x3_tile_row(2 * u, 0) * 16 + io.lane_w);
                X3_STORE(__builtin_amdgcn_perm(hw, lw, 0x07060302u), tile + x3_tile_row(2 * u + 1, 0) * 16 + io.lane_w);
            }
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
    fetch_bits(std::integral_constant<int, 0>{});
    static_for<(DEPTH < P ? DEPTH : P)>([&](auto p) { read_a(p); });
    __builtin_amdgcn_sched_barrier(0);
    static_for<P>([&](auto pc) {
        constexpr int p = decltype(pc)::value, b = p / NKS, ks = p % NKS;
        const bf16x8 &bh = bsrc<KIND, ks>(Hh, Xh), &bl = bsrc<KIND, ks>(Hl, Xl);
        // staging pieces of this step: piece i lives in step (i * PS) / NP (more than one per step when NP > PS), PS = the
        // leading part of the stage the fetches are confined to.  Training kernels: 60 % -- vmcnt retires in order, so the
        // parks at the end of a stage wait for every tile store issued before the last fetch; a last fetch issued early has
        // only long-acknowledged stores in front of it (backward-data 2.07 -> 1.97 ms)
        constexpr int LD_PCT = MODE == 0 ? 100 : 60;
        constexpr int PS = (P * LD_PCT + 99) / 100 > 0 ? (P * LD_PCT + 99) / 100 : 1;
        constexpr int i0 = NP > 0 ? (p < PS ? (p * NP + PS - 1) / PS : NP) : 0;
        constexpr int i1 = NP > 0 ? (p + 1 < PS ? ((p + 1) * NP + PS - 1) / PS : NP) : 0;
        // gap 1: park the staging pieces whose loads have had PFD slots to land
        acc[b & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ring_l[p % DEPTH], bh, acc[b & 1], 0, 0, 0);
        static_for<(i1 - i0 > 0 ? i1 - i0 : 0)>([&](auto dc) {
            constexpr int i = i0 + decltype(dc)::value;
            if constexpr (i < NP && i >= PFD) st_piece(std::integral_constant<int, (i >= PFD ? i - PFD : 0)>{});
        });
        __builtin_amdgcn_sched_barrier(0);
        // gap 2: re-pack of the previous block's tile, its 8 register pairs spread over k-steps 1 .. NKS-2 -- always
        // BEFORE the next block's bias piece: blocks b-1 and b+1 share an accumulator buffer, bias piece g (registers
        // 4g..4g+3) lands in step NKS-5+g, pair u (registers 2u, 2u+1) is re-packed no later than that
        acc[b & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ring_h[p % DEPTH], bl, acc[b & 1], 0, 0, 0);
        if constexpr (b > 0) {
            static_for<8>([&](auto uc) {
                constexpr int u = decltype(uc)::value;
                constexpr int when = NKS >= 6 ? 1 + (u * (NKS - 2)) / 8 : NKS - 1;
                static_assert(NKS < 6 || when <= NKS - 5 + u / 2, "re-pack after the bias overwrite");
                if constexpr (when == ks) repack(std::integral_constant<int, (b > 0 ? b - 1 : 0)>{}, uc);
            });
        }
        __builtin_amdgcn_sched_barrier(0);
        // gap 3: the A fragments DEPTH steps ahead, the next block's start values, the fetch of this step's staging pieces
        acc[b & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ring_h[p % DEPTH], bh, acc[b & 1], 0, 0, 0);
        if constexpr (p + DEPTH < P) read_a(std::integral_constant<int, p + DEPTH>{});
        if constexpr (b + 1 < NBLK) {
            if constexpr (NKS >= 6) {
                if constexpr (ks >= NKS - 5 && ks < NKS - 1)
                    read_bias(std::integral_constant<int, b + 1>{}, std::integral_constant<int, (ks >= NKS - 5 ? ks - (NKS - 5) : 0)>{});
            } else if constexpr (ks == NKS - 1) {  // short blocks (K = 32): everything at the block's last step
                static_for<4>([&](auto g) { read_bias(std::integral_constant<int, b + 1>{}, g); });
            }
            if constexpr (ks == 0) fetch_bits(std::integral_constant<int, b + 1>{});
        }
        static_for<(i1 - i0 > 0 ? i1 - i0 : 0)>([&](auto dc) {
            constexpr int i = i0 + decltype(dc)::value;
            if constexpr (i < NP) ld_piece(std::integral_constant<int, i>{});
        });
        __builtin_amdgcn_sched_barrier(0);
    });
    static_for<(NP < PFD ? NP : PFD)>([&](auto tc) {  // the pieces still in registers
        constexpr int t = decltype(tc)::value, first_left = NP < PFD ? 0 : NP - PFD;
        st_piece(std::integral_constant<int, first_left + t>{});
    });
    constexpr int lb = NBLK - 1;  // the stage's last block is re-packed right away
    static_for<8>([&](auto uc) { repack(std::integral_constant<int, lb>{}, uc); });
#ifndef X3_NO_REPACK
    keep = acc[lb & 1];  // the layer's last tile (heads: rgb / mu,sigma rows; dir layer: alpha row)
#endif
}

// One layer (backward: one step) = its stages.  On entry its first stage sits in LDS buffer PAR (parked by the previous
// stage).  NEXT = layer whose first stage is fetched during this layer's last stage (-1: none).
// FIRST: the kernel's first layer -- its first stage arrives by LDS-DMA, which only an explicit vmcnt(0) can wait for.  No
// other stage may do that: in the training kernels it would also wait for every activation / delta store issued so far to
// be acknowledged, at each of the 44 stage boundaries.
template <class PL, int L, int KIND, int NEXT, int PAR, int NCONV, bool HN_AGPR, int MODE, int PFD, bool ACT, bool FIRST = false>
__device__ __forceinline__ void x3_layer(const char *__restrict__ &wp, char *lds, const bf16x8 (&Hh)[16],
                                         const bf16x8 (&Hl)[16], const bf16x8 (&Xh)[8], const bf16x8 (&Xl)[8],
                                         bf16x8 (&Hnh)[16], bf16x8 (&Hnl)[16], f32x16 &keep, int wave, int lane,
                                         const X3TileIO &io) {
    constexpr int K = PL::K[L], NST = x3_stages_of<PL>(L);
    static_for<NST>([&](auto stc) {
        constexpr int st = decltype(stc)::value;
        constexpr int first = st * PL::SPS[L];
        constexpr int nblk = PL::NB[L] - first < PL::SPS[L] ? PL::NB[L] - first : PL::SPS[L];
        char *cur = lds + ((PAR + st) & 1) * X3_STAGE_BYTES_MAX;
        char *nxt = lds + ((PAR + st + 1) & 1) * X3_STAGE_BYTES_MAX;
        if constexpr (FIRST && st == 0) dma_wait();  // the prologue's LDS-DMA of the very first stage
#ifndef X3_NO_BARRIER
        __syncthreads();  // every wave has parked its pieces of stage `st`; the other buffer is free again
#endif
        wp += x3_stage_bytes<PL>(L, st);
        constexpr int nbytes = st + 1 < NST ? x3_stage_bytes<PL>(L, st + 1)
                                            : (NEXT >= 0 ? x3_stage_bytes<PL>(NEXT >= 0 ? NEXT : 0, 0) : 0);
        x3_stage_compute<KIND, K, nblk, first, NCONV, nbytes, HN_AGPR, MODE, PFD, ACT>(cur, Hh, Hl, Xh, Xl, Hnh, Hnl, keep,
                                                                                        lane, wave, wp, nxt, io);
    });
}

// The forward plan: 11 packed layers as in the fp32 / bf16 kernels.
struct X3FwdPlan {
    static constexpr int NL = 11;
    static constexpr int K[11] = {96, 256, 256, 256, 256, 352, 256, 256, 256, 288, 128};
    static constexpr int NB[11] = {8, 8, 8, 8, 8, 8, 8, 8, 8, 5, 1};
    // slices per stage: K=256 -> 2 (66.3 KiB), K=352 -> 1, K=96 -> 4, K=288 -> 1, K=128 -> 1
    static constexpr int SPS[11] = {4, 2, 2, 2, 2, 1, 2, 2, 2, 1, 1};
};
