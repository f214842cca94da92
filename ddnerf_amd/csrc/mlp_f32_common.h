// Shared device helpers of the fp32 fused-MLP kernels (forward, training forward, backward-data).
// See mlp_f32.hip for the formulation (transposed layers, activations chained through the accumulator layout).
#pragma once
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// a packed slice = 32 rows of (K + 4) floats, padded to a whole number of 256-thread x float4 copy rounds (1024
// floats): a partial last round would be a predicated load that hipcc sinks BELOW the MFMA block, next to its
// store, exposing a full L2 round trip per slice (measured: MFMA pipe only 80 % busy before this padding).
__host__ __device__ constexpr int slice_floats(int K) { return (32 * (K + 4) + 1023) / 1024 * 1024; }
#define MAX_SLICE_FLOATS (slice_floats(352))

// ---- fused forward ----------------------------------------------------------------------------------
// B-operand register file of a wave: Breg[0..7] hidden activations, Breg[8..10] xyz features (96),
// Breg[11] view-dir features (27 + 5 zeros).  Backward: Breg[0..7] deltas, Breg[8] the d(raw) tile.
template <int KIND>
__device__ __forceinline__ constexpr int bsel(int q) {
    return KIND == 0 ? 8 + q / 4
         : KIND == 1 ? q / 4
         : KIND == 2 ? (q < 12 ? 8 + q / 4 : (q - 12) / 4)
         : KIND == 3 ? (q < 32 ? q / 4 : 11)
         : KIND == 10 ? 8                          // backward D0: the 32-row tile of d(raw)
         : KIND == 11 ? (q < 16 ? q / 4 : 8)       // backward D1: d(dir hidden) blocks 0..3, then d(raw) (alpha row)
                     : q / 4;
}

// One slice step in its round-1 form (round 4: only step d0 of the backward pass, whose K = 32 slices are four MFMA chunks long, still
// runs it; everything else is on slice_step_early, mlp_f32_fwd.inc): fetch the NEXT slice (PF_N4 float4 pieces, 0 = nothing to fetch) into registers, multiply the
// current one out of LDS, then park the fetched slice in the other LDS buffer; one barrier.
//   init(acc)  -- accumulator start value (bias tile / zero) and any loads whose latency should hide under the MFMAs
//   mid(q, NQ) -- runs inside MFMA chunk q of NQ (work to spread behind the MFMAs: stores of the previous tile)
//   post(acc)  -- runs after the slice's MFMAs (activation, stores of activations / deltas)
template <int KIND, int K, int PF_N4, class Init, class Mid, class Post>
__device__ __forceinline__ void slice_step_hooks(const float *__restrict__ next_src, const float *cur, float *nxt,
                                                 const f32x16 (&Breg)[12], f32x16 &acc, int tid, int lane, Init &&init,
                                                 Mid &&mid, Post &&post) {
    static_assert(PF_N4 % 256 == 0, "slices are padded to whole copy rounds");
    constexpr int ROUNDS = PF_N4 / 256, NQ = K / 8, R2 = ROUNDS > 0 ? 2 * ROUNDS : 1;
    f32x4 pf[ROUNDS > 0 ? ROUNDS : 1];
    f32x4 a[NQ];
    // The statement order below IS the schedule (one wave per SIMD: nothing else hides latencies), pinned with a
    // sched_barrier(0) per chunk: the A fragments are read two chunks (8 MFMAs = 512 cycles) ahead; the next slice's
    // copy rounds are fetched (global -> registers) behind the MFMAs of the first half of the slice and parked in the
    // other LDS buffer behind those of the second half, so neither sits in front of the first or after the last MFMA.
    const float *a_row = cur + (lane & 31) * (K + 4) + 4 * (lane >> 5);
    a[0] = *(const f32x4 *)(a_row);
    a[1] = *(const f32x4 *)(a_row + 8);
    init(acc);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int blk = bsel<KIND>(q);
        const int g = (KIND == 2 && q >= 12) ? (q - 12) % 4 : q % 4;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q].x, Breg[blk][4 * g + 0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q].y, Breg[blk][4 * g + 1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q].z, Breg[blk][4 * g + 2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q].w, Breg[blk][4 * g + 3], acc, 0, 0, 0);
        if (q + 2 < NQ) a[q + 2] = *(const f32x4 *)(a_row + 8 * (q + 2));
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) {
            if ((r * NQ) / R2 == q) pf[r] = *(const f32x4 *)(next_src + 4 * (size_t)(r * 256 + tid));
            if (NQ / 2 + (r * NQ) / R2 == q) *(f32x4 *)(nxt + 4 * (r * 256 + tid)) = pf[r];
        }
        mid(q, NQ);
        __builtin_amdgcn_sched_barrier(0);
    }
    post(acc);
    __syncthreads();
}

// row of accumulator register r inside a 32-row tile, for lane half h
__device__ __forceinline__ constexpr int tile_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }
