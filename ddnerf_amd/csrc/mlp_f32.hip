// K2 (fp32): the whole 8x256 MLP (MipNeRFModel.forward / DepthMipNeRFModel.forward,
// models/base_architectures.py:40-61, 103-126) as ONE kernel on the fp32-input matrix cores
// (v_mfma_f32_32x32x2_f32: bit-for-bit a k-ordered fmaf chain, so the fp32 parity bar holds).
//
// Formulation: every layer is computed TRANSPOSED,  H_out^T[out, sample] = W[out, in] * H_in^T[in, sample].
//   * A operand  = a 32-row block of W (out-features x in-features), staged through LDS and shared by the
//                  four waves of the workgroup;
//   * B operand  = the previous layer's output, which in the MFMA accumulator layout already has the sample
//                  on the lane and the feature on the register -- exactly what the next MFMA wants as B.
// So activations NEVER leave the register file: no LDS round trip, no HBM round trip between the 12 layers.
// A wave owns 32 samples (one 32-wide column block): 8 accumulator tiles (256 features) in, 8 out.
// A workgroup = 4 waves = 128 samples, one workgroup per CU (the kernel wants the full 512-register file).
//
// MFMA lane maps (32x32x2 f32): A[i = lane&31][k = lane>>5], B[k = lane>>5][j = lane&31],
// D[row = (r&3) + 8*(r>>2) + 4*(lane>>5)][col = lane&31], r = 0..15.
// Feeding D register r of block b' back as B means: k-slot h (= lane>>5) carries in-feature
// 32b' + (r&3) + 8(r>>2) + 4h.  For four consecutive r (r = 4g..4g+3) lane half h therefore needs the four
// CONSECUTIVE in-features 32b' + 8g + 4h .. +3 of W's row: one ds_read_b128 of the natural [out][in] layout.
//
// Weights are repacked once per update into the exact LDS image of each 32-row slice (row stride K+4 floats:
// the 16-byte pad makes the b128 fragment reads bank-conflict free), stored in consumption order, so staging
// is a linear copy; the 32 bias values of the slice's rows follow its last row inside the same image (the layers' bias blocks at the
// end of the buffer stay: the training kernels read those).  Slices are double-buffered: slice s+1 is fetched into registers while slice s feeds the
// MFMAs, written to the other LDS buffer behind its later MFMAs, one barrier per slice (in front of its last four MFMAs).
#include "mlp_f32_common.h"

// ---- layer schedule -------------------------------------------------------------------------------
// kind: 0 first layer (K=96 from xyz features)   1 hidden (K=256)   2 skip layer (K=352 = xyz96 + hidden256)
//       3 dir+alpha layer (K=288 = feat256 + dir27 + 5 zero, 160 rows = 128 dir + alpha + 31 zero)
//       4 heads (K=128, 32 rows: rgb 0..2, mu 4, sigma 5)
#define NLAYERS 11
static constexpr int kLayerK[NLAYERS] = {96, 256, 256, 256, 256, 352, 256, 256, 256, 288, 128};
static constexpr int kLayerNB[NLAYERS] = {8, 8, 8, 8, 8, 8, 8, 8, 8, 5, 1};

struct PackPlan {
    int slice_off[NLAYERS];  // float offset of the layer's first slice in the packed buffer
    int bias_off[NLAYERS];   // float offset of the layer's bias block (NB*32 floats)
    int w_src[13];           // float offsets of the 13 weight matrices in the flat parameter buffer
    int b_src[13];           //   "      of the 13 bias vectors
    int total;               // floats in the packed buffer
};

static PackPlan make_plan(int depth_head) {
    PackPlan p;
    static const int nout[13] = {256, 256, 256, 256, 256, 256, 256, 256, 256, 1, 128, 3, 2};
    static const int nin[13] = {96, 256, 256, 256, 256, 352, 256, 256, 256, 256, 283, 128, 128};
    int off = 0;
    for (int l = 0; l < 13; ++l) {
        p.w_src[l] = off;
        off += nout[l] * nin[l];
        p.b_src[l] = off;
        off += nout[l];
        if (l == 11 && !depth_head) {  // no fc_mu_sigma in MipNeRFModel
            p.w_src[12] = p.b_src[12] = -1;
            break;
        }
    }
    off = 0;
    for (int l = 0; l < NLAYERS; ++l) {
        p.slice_off[l] = off;
        off += kLayerNB[l] * slice_floats(kLayerK[l]);
    }
    for (int l = 0; l < NLAYERS; ++l) {
        p.bias_off[l] = off;
        off += kLayerNB[l] * 32;
    }
    p.total = off;
    return p;
}

DDN_EXPORT size_t ddnerf_mlp_f32_packed_floats(int depth_head) { return (size_t)make_plan(depth_head).total; }

// value of packed layer `l`, out-row `o`, in-column `c` (c < K), read from the flat parameter buffer
__device__ __forceinline__ float src_weight(const float *__restrict__ P, const PackPlan &pl, int l, int o, int c) {
    if (l <= 8) return P[pl.w_src[l] + o * kLayerK[l] + c];  // layers_xyz.0..7, fc_feat: K == in_features
    if (l == 9) {
        if (o < 128) return c < 283 ? P[pl.w_src[10] + o * 283 + c] : 0.0f;  // layers_dir.0 on cat(feat, dirs)
        if (o == 128) return c < 256 ? P[pl.w_src[9] + c] : 0.0f;            // fc_alpha on feat
        return 0.0f;
    }
    if (o < 3) return P[pl.w_src[11] + o * 128 + c];                                     // fc_rgb
    if ((o == 4 || o == 5) && pl.w_src[12] >= 0) return P[pl.w_src[12] + (o - 4) * 128 + c];  // fc_mu_sigma
    return 0.0f;
}
__device__ __forceinline__ float src_bias(const float *__restrict__ P, const PackPlan &pl, int l, int o) {
    if (l <= 8) return P[pl.b_src[l] + o];
    if (l == 9) return o < 128 ? P[pl.b_src[10] + o] : (o == 128 ? P[pl.b_src[9]] : 0.0f);
    if (o < 3) return P[pl.b_src[11] + o];
    if ((o == 4 || o == 5) && pl.b_src[12] >= 0) return P[pl.b_src[12] + (o - 4)];
    return 0.0f;
}

__global__ void mlp_f32_pack_kernel(const float *__restrict__ P, PackPlan pl, float *__restrict__ packed) {
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= pl.total) return;
    int l;
    if (idx >= pl.bias_off[0]) {
        for (l = NLAYERS - 1; l > 0; --l)
            if (idx >= pl.bias_off[l]) break;
        packed[idx] = src_bias(P, pl, l, idx - pl.bias_off[l]);
        return;
    }
    for (l = NLAYERS - 1; l > 0; --l)
        if (idx >= pl.slice_off[l]) break;
    const int ld = kLayerK[l] + 4, local = idx - pl.slice_off[l];
    const int sl = local / slice_floats(kLayerK[l]), within = local % slice_floats(kLayerK[l]);
    float v = 0.0f;
    if (within < 32 * ld) {
        const int o = 32 * sl + within / ld, c = within % ld;
        if (c < kLayerK[l]) v = src_weight(P, pl, l, o, c);
    } else if (within < 32 * ld + 32) {
        v = src_bias(P, pl, l, 32 * sl + within - 32 * ld);  // the slice's bias tile rides in its image (round 4): it reaches LDS with the rows
    }
    packed[idx] = v;
}

DDN_EXPORT int ddnerf_mlp_f32_pack(const float *params, int depth_head, float *packed, ddnerf_stream_t stream) {
    DDN_REQUIRE(params && packed, DDNERF_E_ARG);
    DDN_REQUIRE(ddn_aligned(packed, 16), DDNERF_E_ALIGN);
    PackPlan pl = make_plan(depth_head);
    hipLaunchKernelGGL(mlp_f32_pack_kernel, dim3((pl.total + 255) / 256), dim3(256), 0, (hipStream_t)stream, params, pl,
                       packed);
    return ddn_launch_status();
}

#ifdef F32_STAMP_TILE
__device__ unsigned long long g_f32_tile[4096 * 6];
#endif
#ifdef F32_STAMP
// Diagnostic build (tools/f32_clock.py): eight clock stamps per slice, held in scalar registers (one s_memtime each, nothing waits on
// them inside the slice) and written out by thread 0 of every 64th workgroup behind the slice's last MFMA.
#define F32_STAMP_SLOTS 96
__device__ unsigned long long g_f32_stamps[64 * F32_STAMP_SLOTS * 8];
__shared__ unsigned f32_stamp_idx;
#define F32_STAMP_DECL unsigned long long f32_st[8] = {}
#define F32_STAMP_AT(i) asm volatile("s_memtime %0" : "=s"(f32_st[i])::"memory")
#define F32_STAMP_FLUSH()                                                                          \
    do {                                                                                           \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                         \
        if (tid == 0 && blockIdx.x % 64 == 0 && blockIdx.x / 64 < 64) {                            \
            const unsigned i_ = f32_stamp_idx;                                                     \
            f32_stamp_idx = i_ + 1;                                                                \
            if (i_ < F32_STAMP_SLOTS)                                                              \
                for (int k_ = 0; k_ < 8; ++k_) g_f32_stamps[((blockIdx.x / 64) * F32_STAMP_SLOTS + i_) * 8 + k_] = f32_st[k_]; \
        }                                                                                          \
    } while (0)
#else
#define F32_STAMP_DECL
#define F32_STAMP_AT(i)
#define F32_STAMP_FLUSH()
#endif

// ReLU WITHOUT vector-ALU instructions (round 4).  With one wave per SIMD a VALU or vector-memory instruction does not overlap the wave's
// own MFMAs: tools/calib/f32_chain.hip measures 64.0 cycles per v_mfma_f32_32x32x2_f32 with nothing, an LDS instruction or scalar
// instructions between two of them, 80.5 with one VALU instruction, +4 for every further one, 76.3 with one global load.  An activation
// done by v_accvgpr_read + v_max costs 2.8 % of the launch however it is spread.  LDS instructions are free, and the LDS has an ALU:
// a tile is pushed through a per-wave scratch area of zeros with ds_max_f32 (memory = max(memory, x) = relu(x)), read back, and the area
// zeroed again -- 36 LDS instructions per tile, each behind an MFMA of its own, in program order on the wave's own 4 KiB (the LDS
// executes one wave's instructions in order: no barrier).  Scratch layout [register r][lane]: conflict-free for the 32-bit operations.
#define SCR_FLOATS (16 * 64)
__device__ __forceinline__ f32x16 lds_bias_tile(const float *tile32, int h) {   // accumulator layout: register 4g + i = row 8g + 4h + i
    f32x16 v;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const f32x4 b = *(const f32x4 *)(tile32 + 8 * g + 4 * h);
        v[4 * g + 0] = b.x;
        v[4 * g + 1] = b.y;
        v[4 * g + 2] = b.z;
        v[4 * g + 3] = b.w;
    }
    return v;
}
__device__ __forceinline__ void relu_push(float *scr, float x, int r) { __hip_atomic_fetch_max(scr + 64 * r, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ float relu_pull(float *scr, int r) { return __hip_atomic_load(scr + 64 * r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void relu_rezero(float *scr_wave, int lane, int k) { *(f32x4 *)(scr_wave + 256 * k + 4 * lane) = f32x4{0.0f, 0.0f, 0.0f, 0.0f}; }

// The inference kernel's slice step (round 4; the training kernels keep mlp_f32_common.h's slice_step_hooks).  Same work as there --
// multiply the current slice out of LDS while the next one (ROUNDS x 4 KiB at byte `next_off` of the packed buffer; 0 rounds: none)
// is fetched into registers and parked in the other LDS buffer -- with three differences, all of them schedule only:
//   * ONE piece of side work behind EACH MFMA.  The MFMAs of a tile are a dependent chain that issues back to back, 64 cycles apart;
//     whatever sits between two of them in program order must issue inside that window.  A chunk's fragment read, fetch, park and
//     activation instructions all behind its fourth MFMA did not (measured: the ReLU cost 2.8 %, the weight fetches 2.6 % of the launch).
//     mid(q, i) runs behind MFMA i = 0..3 of chunk q.
//   * The barrier sits in front of the slice's LAST chunk instead of behind it -- every park of the next slice and every fragment
//     read of this one has been issued at least a chunk earlier -- and the next slice's first two A fragments are read right behind
//     it, under the last chunk's MFMAs: `carry` hands them to the next call, which starts on its MFMAs at once.
//     The next slice's bias tile (it rides behind the slice's rows) is read there too: `bnext`.
//   * The fetches are buffer loads (scalar offset of the piece + 16 x tid): no 64-bit vector address arithmetic.
template <int KIND, int K, int ROUNDS, int NEXT_K, class Init, class Mid>
__device__ __forceinline__ void slice_step_early(__amdgpu_buffer_rsrc_t wsrc, unsigned next_off, const float *cur, float *nxt,
                                                 const f32x16 (&Breg)[12], f32x16 &acc, f32x4 (&carry)[2], f32x16 &bnext, int tid, int lane,
                                                 Init &&init, Mid &&mid) {
    constexpr int NQ = K / 8, R2 = ROUNDS > 0 ? 2 * ROUNDS : 1;
    static_assert(NQ >= 8, "parks are capped at chunk NQ - 3");
    f32x4 pf[ROUNDS > 0 ? ROUNDS : 1];
    f32x4 a[NQ];
    const float *a_row = cur + (lane & 31) * (K + 4) + 4 * (lane >> 5);
    const unsigned voff = 16u * (unsigned)tid;
    F32_STAMP_DECL;
    F32_STAMP_AT(0);
    a[0] = carry[0];
    a[1] = carry[1];
    init(acc);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int blk = bsel<KIND>(q);
        const int g = (KIND == 2 && q >= 12) ? (q - 12) % 4 : q % 4;
        if (q == NQ - 1) {   // every wave has parked its pieces of the next slice and read its last fragments of this one
#ifndef F32_EXP_NOBARRIER   // (F32_EXP_*: diagnostic builds, timing only)
            __syncthreads();
#endif
            if constexpr (NEXT_K > 0) {
                const float *n_row = nxt + (lane & 31) * (NEXT_K + 4) + 4 * (lane >> 5);
                carry[0] = *(const f32x4 *)(n_row);
                carry[1] = *(const f32x4 *)(n_row + 8);
#ifndef F32_EXP_NOBIAS
                bnext = lds_bias_tile(nxt + 32 * (NEXT_K + 4), lane >> 5);
#endif
            }
            F32_STAMP_AT(6);
            __builtin_amdgcn_sched_barrier(0);
        }
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q].x, Breg[blk][4 * g + 0], acc, 0, 0, 0);
        if (q + 2 < NQ) a[q + 2] = *(const f32x4 *)(a_row + 8 * (q + 2));
        mid(q, 0);
        __builtin_amdgcn_sched_barrier(0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q].y, Breg[blk][4 * g + 1], acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) {
#ifdef F32_EXP_NOLOAD   // (every round re-reads the first KiB of the slice: cache hits)
            if ((r * NQ) / R2 == q) pf[r] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wsrc, voff & 1008u, next_off, 0));
#else
            if ((r * NQ) / R2 == q) pf[r] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wsrc, voff, next_off + 4096u * r, 0));
#endif
        }
        mid(q, 1);
        __builtin_amdgcn_sched_barrier(0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q].z, Breg[blk][4 * g + 2], acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) {
            // (no park later than chunk NQ - 3, so that the barrier in front of chunk NQ - 1 finds every LDS operation a chunk old)
            const int park = NQ / 2 + (r * NQ) / R2;
#ifdef F32_EXP_NOPARK   // (the fetched pieces are consumed by a no-op instead of being written to LDS)
            if ((park > NQ - 3 ? NQ - 3 : park) == q) asm volatile("" ::"v"(pf[r]));
#else
            if ((park > NQ - 3 ? NQ - 3 : park) == q) *(f32x4 *)(nxt + 4 * (r * 256 + tid)) = pf[r];
#endif
        }
        mid(q, 2);
        __builtin_amdgcn_sched_barrier(0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q].w, Breg[blk][4 * g + 3], acc, 0, 0, 0);
        mid(q, 3);
        if (q == 0) F32_STAMP_AT(1);
        if (q == 1) F32_STAMP_AT(2);
        if (q == NQ / 2 - 1) F32_STAMP_AT(3);
        if (q == NQ - 3) F32_STAMP_AT(4);
        if (q == NQ - 2) F32_STAMP_AT(5);
        if (q == NQ - 1) F32_STAMP_AT(7);
        __builtin_amdgcn_sched_barrier(0);
    }
    F32_STAMP_FLUSH();
}

// One layer: NB slices.  `woff` walks the packed buffer in bytes (slices are stored in consumption order); while slice s
// is multiplied, slice s+1 (possibly the next layer's first one, NEXT_K wide; NEXT_K = 0: none) is fetched.
// PAR = parity of the LDS buffer that holds this layer's first slice.
// Nothing but MFMAs sits between two layers (round 4): `bcur` arrives holding this layer's first bias tile and leaves holding the
// next layer's, `carry` likewise the first two A fragments (both read from LDS under the last MFMAs of the slice before), and the ReLU
// of output tile b - 1 (tiles < RELU_NB) rides through the LDS behind the MFMAs of slice b (relu_push in chunks 1-4, relu_pull in
// chunks 5-8, relu_rezero in chunk 9).  The last tile's is left to the next layer (PEND), which applies it to its input tile 7 behind
// the MFMAs of its slice 0 -- tile 7 is the last one a slice reads in every layer kind.
template <int KIND, int K, int NB, int NEXT_K, int PAR, int RELU_NB, bool PEND = false>
__device__ __forceinline__ void layer(__amdgpu_buffer_rsrc_t wsrc, unsigned &woff, float *lds, float *scr_wave, f32x16 (&Breg)[12],
                                      f32x16 (&out)[8], f32x16 &bcur, f32x4 (&carry)[2], int tid, int lane) {
    static_assert(!PEND || bsel<KIND>(9) != 7, "PEND: tile 7 must not feed the first ten chunks");
    static_assert(K / 8 >= 12, "the ReLU schedule uses chunks 1..9");
    static_assert(RELU_NB <= NB - 1 || RELU_NB == NB, "a layer's last tile is either not activated or left to the next layer");
    constexpr int ROUNDS = slice_floats(K) / 1024;  // 4-KiB copy rounds per slice of this layer
    constexpr int NEXT_ROUNDS = NEXT_K > 0 ? slice_floats(NEXT_K) / 1024 : 0;
    float *scr = scr_wave + lane;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const float *cur = lds + ((PAR + b) & 1) * MAX_SLICE_FLOATS;
        float *nxt = lds + ((PAR + b + 1) & 1) * MAX_SLICE_FLOATS;
        woff += 4096u * ROUNDS;  // now the byte offset of the slice after the current one
        f32x16 bnext;
        auto init = [&](f32x16 &a) { a = bcur; };
        auto mid = [&](int q, int i) {   // one LDS instruction behind each MFMA of chunks 1..9
#ifdef F32_EXP_NORELU
            return;
#endif
            const bool mine = b >= 1 && b - 1 < RELU_NB, pend = PEND && b == 0;
            if (!mine && !pend) return;
            f32x16 &t = mine ? out[b >= 1 ? b - 1 : 0] : Breg[7];
            if (q >= 1 && q <= 4) relu_push(scr, t[4 * (q - 1) + i], 4 * (q - 1) + i);
            if (q >= 5 && q <= 8) t[4 * (q - 5) + i] = relu_pull(scr, 4 * (q - 5) + i);
            if (q == 9) relu_rezero(scr_wave, lane, i);
        };
        if (b + 1 < NB) slice_step_early<KIND, K, ROUNDS, K>(wsrc, woff, cur, nxt, Breg, out[b], carry, bnext, tid, lane, init, mid);
        else slice_step_early<KIND, K, NEXT_ROUNDS, NEXT_K>(wsrc, woff, cur, nxt, Breg, out[b], carry, bnext, tid, lane, init, mid);
        if (b + 1 < NB || NEXT_K > 0) bcur = bnext;
    }
}

// load feature columns [32*b0, 32*(b0+nb)) of this lane's sample into Breg[dst..] (B layout, see header)
template <int DST, int B0, int NBLK>
__device__ __forceinline__ void load_features(const float *__restrict__ frow, int h, f32x16 (&Breg)[12]) {
#pragma unroll
    for (int b = 0; b < NBLK; ++b) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 v = *(const f32x4 *)(frow + 32 * (B0 + b) + 8 * g + 4 * h);
            Breg[DST + b][4 * g + 0] = v.x;
            Breg[DST + b][4 * g + 1] = v.y;
            Breg[DST + b][4 * g + 2] = v.z;
            Breg[DST + b][4 * g + 3] = v.w;
        }
    }
}

// Persistent (round 4): one workgroup per CU walks the 128-sample tiles blockIdx.x, blockIdx.x + gridDim.x, ...  The only exposed
// fetches are the first tile's: every later tile finds its slice 0 in LDS (fetched under the head layer of the tile before, whose
// "next slice" wraps to the buffer's start), its first fragments and bias tile in registers, and its xyz features in Breg[8..10]
// (fetched under layers_dir).  A workgroup per tile paid 1.7 % of the launch between workgroups and 0.3 % in the prologue.
template <bool DEPTH>
__global__ __launch_bounds__(256, 1) void mlp_f32_fwd_kernel(const float *__restrict__ feat,
                                                             const float *__restrict__ packed, PackPlan pl,
                                                             float *__restrict__ raw, long M) {
    __shared__ __attribute__((aligned(16))) float lds[2 * MAX_SLICE_FLOATS + 4 * SCR_FLOATS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, h = lane >> 5;
    float *scr_wave = lds + 2 * MAX_SLICE_FLOATS + SCR_FLOATS * wave;  // this wave's ReLU scratch (relu_push / _pull / _rezero)
#pragma unroll
    for (int k = 0; k < 4; ++k) relu_rezero(scr_wave, lane, k);
    const long ntiles = (M + 127) / 128;
    // row of this lane's sample in tile t (clamped: lanes past the end compute on the last sample and store nothing)
    auto feat_row = [&](long t) {
        const long mm = t * 128 + wave * 32 + j;
        return feat + (size_t)(mm < M ? mm : M - 1) * DDNERF_FEAT_LD;
    };

    f32x16 Breg[12];
    f32x16 out[8];
#ifdef F32_STAMP
    if (tid == 0) f32_stamp_idx = 0;
#endif
#ifdef F32_STAMP_TILE   // (diagnostic build: first and last instruction of every workgroup on the shader clock and on the 100 MHz clock)
    const unsigned long long tile_t0 = __builtin_amdgcn_s_memtime(), tile_r0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long tile_t1 = 0;
#endif

    // first tile only: stage slice 0 synchronously, fetch the sample's xyz features into B layout meanwhile
    const __amdgpu_buffer_rsrc_t wsrc = __builtin_amdgcn_make_buffer_rsrc((void *)packed, 0, 4 * pl.bias_off[0], 0x00020000);  // the slices
    {
        constexpr int ROUNDS = slice_floats(96) / 1024;
        f32x4 pf[ROUNDS];
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) pf[r] = *(const f32x4 *)(packed + 4 * (size_t)(r * 256 + tid));
        load_features<8, 0, 3>(feat_row(blockIdx.x), h, Breg);  // xyz features; dead again after layer 0
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) *(f32x4 *)(lds + 4 * (r * 256 + tid)) = pf[r];
    }
    __syncthreads();
#ifdef F32_STAMP_TILE
    tile_t1 = __builtin_amdgcn_s_memtime();
#endif
    f32x16 bcur = lds_bias_tile(lds + 32 * (96 + 4), h);  // the only exposed bias and fragment reads of the workgroup
    f32x4 carry[2];
    carry[0] = *(const f32x4 *)(lds + j * (96 + 4) + 4 * h);
    carry[1] = *(const f32x4 *)(lds + j * (96 + 4) + 4 * h + 8);

    for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long m = tile * 128 + wave * 32 + j;
        const bool valid = m < M;
        const float *frow = feat_row(tile);
        unsigned woff = 0;
#define NEXT_LAYER()                                  \
    _Pragma("unroll") for (int b = 0; b < 8; ++b) Breg[b] = out[b];
        // (straight-line: the output tiles of one layer ARE the B operands of the next -- a loop would copy 128 registers per trip)
        // layer 0: 96 -> 256, ReLU                                             base_architectures.py:42-43
        layer<0, 96, 8, 256, 0, 8, false>(wsrc, woff, lds, scr_wave, Breg, out, bcur, carry, tid, lane);
        NEXT_LAYER();
        // layers 1..4: 256 -> 256, ReLU                                        :44-49
        layer<1, 256, 8, 256, 0, 8, true>(wsrc, woff, lds, scr_wave, Breg, out, bcur, carry, tid, lane);
        NEXT_LAYER();
        layer<1, 256, 8, 256, 0, 8, true>(wsrc, woff, lds, scr_wave, Breg, out, bcur, carry, tid, lane);
        NEXT_LAYER();
        layer<1, 256, 8, 256, 0, 8, true>(wsrc, woff, lds, scr_wave, Breg, out, bcur, carry, tid, lane);
        NEXT_LAYER();
        layer<1, 256, 8, 352, 0, 8, true>(wsrc, woff, lds, scr_wave, Breg, out, bcur, carry, tid, lane);
        NEXT_LAYER();
        // layer 5: cat(xyz, x) 352 -> 256, ReLU                                :45-46
        load_features<8, 0, 3>(frow, h, Breg);  // re-fetched (L2) instead of held in 48 registers across layers 1-4
        __builtin_amdgcn_sched_barrier(0);
        layer<2, 352, 8, 256, 0, 8, true>(wsrc, woff, lds, scr_wave, Breg, out, bcur, carry, tid, lane);
        NEXT_LAYER();
        // layers 6, 7 (ReLU) and fc_feat (no activation)                       :47-50
        layer<1, 256, 8, 256, 0, 8, true>(wsrc, woff, lds, scr_wave, Breg, out, bcur, carry, tid, lane);
        NEXT_LAYER();
        layer<1, 256, 8, 256, 0, 8, true>(wsrc, woff, lds, scr_wave, Breg, out, bcur, carry, tid, lane);
        NEXT_LAYER();
        layer<1, 256, 8, 288, 0, 0, true>(wsrc, woff, lds, scr_wave, Breg, out, bcur, carry, tid, lane);
        NEXT_LAYER();
        // layers_dir.0 on cat(feat, dirs) + fc_alpha on feat: 288 -> 160       :51-56
        load_features<11, 3, 1>(frow, h, Breg);  // view-dir columns 96..127
        load_features<8, 0, 3>(feat_row(tile + gridDim.x < ntiles ? tile + gridDim.x : tile), h, Breg);  // the NEXT tile's xyz features
        __builtin_amdgcn_sched_barrier(0);
        layer<3, 288, 5, 128, 0, 4>(wsrc, woff, lds, scr_wave, Breg, out, bcur, carry, tid, lane);
        const float alpha = out[4][0];  // row 128 = block 4, register 0, lane half 0 (tile 4 carries no ReLU)
#pragma unroll
        for (int b = 0; b < 4; ++b) Breg[b] = out[b];
        // fc_rgb (+ fc_mu_sigma): 128 -> 32 rows; its "next slice" is slice 0 again, for the next tile      :60 / :123-124
        woff = 0u - 4096u * (slice_floats(128) / 1024);
        layer<4, 128, 1, 96, 1, 0>(wsrc, woff, lds, scr_wave, Breg, out, bcur, carry, tid, lane);
#undef NEXT_LAYER

        if (valid) {
            if (DEPTH) {
                float *o = raw + (size_t)m * 6;
                if (h == 0) {
                    *(float2 *)(o) = make_float2(out[0][0], out[0][1]);
                    *(float2 *)(o + 2) = make_float2(out[0][2], alpha);
                } else {
                    *(float2 *)(o + 4) = make_float2(out[0][0], out[0][1]);  // rows 4, 5 = raw mu, raw sigma
                }
            } else if (h == 0) {
                *(f32x4 *)(raw + (size_t)m * 4) = f32x4{out[0][0], out[0][1], out[0][2], alpha};
            }
        }
    }
#ifdef F32_STAMP_TILE
    {
        const unsigned long long t2 = __builtin_amdgcn_s_memtime();   // (behind the last MFMA's issue and the store's issue)
        if (tid == 0 && blockIdx.x < 4096) {
            unsigned long long *o = g_f32_tile + 6 * blockIdx.x;
            o[0] = tile_t0;
            o[1] = tile_t1;
            o[2] = t2;
            o[3] = tile_r0;
            o[4] = __builtin_amdgcn_s_memrealtime();
            o[5] = 0;
        }
    }
#endif
}

#ifdef F32_STAMP_TILE
DDN_EXPORT int ddnerf_debug_f32_tile_stamps(unsigned long long *host_out) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_f32_tile), sizeof(g_f32_tile));
}
#endif
#ifdef F32_STAMP
DDN_EXPORT int ddnerf_debug_f32_stamps(unsigned long long *host_out) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_f32_stamps), sizeof(g_f32_stamps));
}
#endif

DDN_EXPORT int ddnerf_mlp_f32_forward(const float *feat, const float *packed, int depth_head, float *raw, long M,
                                      ddnerf_stream_t stream) {
    DDN_REQUIRE(feat && packed && raw, DDNERF_E_ARG);
    DDN_REQUIRE(M > 0, DDNERF_E_ARG);
    DDN_REQUIRE(ddn_aligned(feat, 16) && ddn_aligned(packed, 16) && ddn_aligned(raw, 16), DDNERF_E_ALIGN);
    PackPlan pl = make_plan(depth_head);
    const long ntiles = (M + 127) / 128, cus = ddn_cu_count();
    dim3 grid((unsigned)(ntiles < cus ? ntiles : cus));   // persistent: the kernel needs a CU's whole register file and most of its LDS
    if (depth_head)
        hipLaunchKernelGGL(mlp_f32_fwd_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, feat, packed, pl, raw, M);
    else
        hipLaunchKernelGGL(mlp_f32_fwd_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, feat, packed, pl, raw, M);
    return ddn_launch_status();
}
