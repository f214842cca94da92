// The fp32 training kernels of mlp_f32_train.hip built with `acts` / `deltas` as blocked records of the fp32 VALUES
// (ddnerf_mlp_f32_forward_train_recf, ddnerf_mlp_f32_backward_data_recf; round 5): the layout of the hi/lo-word records
// (mlp_f32_train_rec.hip), the values unsplit.  Their weight gradients run on ddnerf_mlp_x3_wgrad_blocked (mlp_x3_wgrad_packed.hip),
// which splits each value into bf16 hi / lo where it builds its MFMA fragments -- the same split and products, so the same
// gradients bit for bit -- and leaves these kernels' fp32 MFMA chains without the 3.5 vector-ALU instructions per recorded element.
// F32_AHEAD 4: weight fragments are read from LDS four chunks ahead of their MFMAs instead of two.  The backward's sign masks arrive by
// scalar loads the compiler's s_waitcnt insertion does not know of (mlp_f32_train.hip, SignLoader): while one is in flight every lgkmcnt(N)
// the compiler emits for an LDS result waits for one LDS operation more than it meant to, and at distance 2 that is the fragment read issued
// 64 cycles earlier: the backward ran 1.9 % SLOWER with the masks than without; at distance 4 it runs 6.8 % faster (4.27 against 4.58 ms).
#define F32_AHEAD 4
// F32_TRAIN_DMA: the weight slices travel global -> LDS by LDS-DMA as in the inference kernel (no registers, no park).  The slice's wait in
// front of its barrier is vmcnt(4) / vmcnt(16) instead of vmcnt(0): loads and stores retire in one order and a slice's record stores are
// the only vector-memory instructions behind its last piece, so they may stay in flight.  Forward 4.69 -> 4.60, backward 4.28 -> 4.25 ms.
#define F32_TRAIN_DMA 1
#define F32_REC 3
#include "mlp_f32_train.hip"
