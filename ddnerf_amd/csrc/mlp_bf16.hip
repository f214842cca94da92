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
#include "common.h"
#define M16_PLANES 1
// (in the library this kernel is ddnerf_mlp_bf16g1_*: ddnerf_mlp_bf16_* picks between it and the two-group kernel by launch size,
// mlp_bf16_g2.hip; a stand-alone build of this file for A/B timing exports it as ddnerf_mlp_bf16_* itself)
// M16_HALF (mlp_f16.hip compiles this file a second time with it): the fp16 tier -- v_mfma_f32_16x16x32_f16 and v_cvt_pk_f16_f32 in place
// of the bf16 forms, weights packed as fp16; everything else is shared.
#ifdef M16_HALF
#define M16_SYM(x) ddnerf_mlp_f16g1_##x
#define M16_KERNEL mlp_f16_fwd_kernel
#define M16_PACK_KERNEL mlp_f16_pack_kernel
#else
#ifdef BF16_DISPATCH
#define M16_SYM(x) ddnerf_mlp_bf16g1_##x
#else
#define M16_SYM(x) ddnerf_mlp_bf16_##x
#endif
#define M16_KERNEL mlp_bf16_fwd_kernel
#define M16_PACK_KERNEL mlp_bf16_pack_kernel
#endif
#define M16_FEAT_T void
#define NSTAGE 40
// stage -> (layer, first block, blocks); the last stage also carries the heads block
static constexpr int kStage[NSTAGE][3] = {
    {0, 0, 6}, {0, 6, 5}, {0, 11, 5},
    {1, 0, 4}, {1, 4, 4}, {1, 8, 4}, {1, 12, 4}, {2, 0, 4}, {2, 4, 4}, {2, 8, 4}, {2, 12, 4},
    {3, 0, 4}, {3, 4, 4}, {3, 8, 4}, {3, 12, 4}, {4, 0, 4}, {4, 4, 4}, {4, 8, 4}, {4, 12, 4},
    {5, 0, 3}, {5, 3, 3}, {5, 6, 3}, {5, 9, 3}, {5, 12, 2}, {5, 14, 2},
    {6, 0, 4}, {6, 4, 4}, {6, 8, 4}, {6, 12, 4}, {7, 0, 4}, {7, 4, 4}, {7, 8, 4}, {7, 12, 4},
    {8, 0, 4}, {8, 4, 4}, {8, 8, 4}, {8, 12, 4},
    {9, 0, 3}, {9, 3, 3}, {9, 6, 3}};

#include "mlp_mfma16.inc"
