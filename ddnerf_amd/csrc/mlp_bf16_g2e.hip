// K1 + K2 (bf16) in ONE kernel: cast_rays + integrated_pos_enc + the view-direction columns + the 8x256 MLP -- the reference's
// run_network (models/models.py:117-142: cast_rays, integrated_pos_enc, positional_encoding of the view directions, concat, the network)
// is one step, and so it is here: the encoded feature rows [M,128] never exist in HBM.  The tile body is the two-group body of
// mlp_bf16_g2.hip generated with its encoder switched on (gen_bf16_g2.py, Gen(fused=True): the design note is in its header): while
// layers 6 - 8 of a tile run, the vector-ALU gaps of the MFMA stream encode the NEXT tile's samples (lane = sample: the Gaussian of the
// sample's interval, then per octave and axis the damped sine / cosine on the hardware transcendentals) into 192-byte rows of a scratch
// area private to the workgroup (96 KiB, rewritten every tile: it lives in the L2), from which the body fetches what the unfused body
// fetches from the feature rows.  The view-direction k-step comes from a per-RAY table (27 values per ray, not per sample).
// Outputs are bit-identical to ddnerf_encode(feat_dtype = 1) followed by ddnerf_mlp_bf16_forward (tests/test_hip_fused_mlp.py).
#include "mlp_bf16_common.h"

#include "mlp_bf16_g2_tables.gen.inc"

// G2E_HALF (mlp_f16_g2e.hip compiles this file a second time with it): the fp16 tier's fused kernel -- the body generated on the f16 forms of the
// MFMA and of the two conversions (re-pack, encoder), the ray table's view-direction row in fp16 (ddnerf_ray_table, feat_dtype 2).
#ifdef G2E_HALF
#define G2E_KERNEL mlp_f16g2e_fwd_kernel
#define G2E_SYM(x) ddnerf_encode_mlp_f16_##x
#define G2E_G1_BYTES ddnerf_mlp_f16g1_packed_bytes
#define G2E_STAMP_VAR g_f16g2e_stamps
#define G2E_STAMP_SET ddnerf_debug_set_stamps_f16g2e
#define G2E_BODY_D0 "mlp_f16_g2e_body_d0.gen.inc"
#define G2E_BODY_D1 "mlp_f16_g2e_body_d1.gen.inc"
#else
#define G2E_KERNEL mlp_bf16g2e_fwd_kernel
#define G2E_SYM(x) ddnerf_encode_mlp_bf16_##x
#define G2E_G1_BYTES ddnerf_mlp_bf16g1_packed_bytes
#define G2E_STAMP_VAR g_bf16g2e_stamps
#define G2E_STAMP_SET ddnerf_debug_set_stamps_g2e
#define G2E_BODY_D0 "mlp_bf16_g2e_body_d0.gen.inc"
#define G2E_BODY_D1 "mlp_bf16_g2e_body_d1.gen.inc"
#endif

#define G2E_SLOT_BYTES (36 * 1024)
#define G2E_LDS_BYTES (4 * G2E_SLOT_BYTES)
#define G2E_TILE 512
#define G2E_SCRATCH_PER_WG (G2E_TILE * G2E_ROW_BYTES)
#define G2E_TABLE_FLOATS 32   // per ray (rays_encode.hip, ddnerf_ray_table): 16 floats, then 32 bf16 view-direction columns in k-order

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// ---- the kernel ------------------------------------------------------------------------------------------------------------------
struct G2EArgs {
    const float *t_vals;     // [n, S + 1]
    const float *table;      // ddnerf_ray_table
    const char *packed;      // the two-group kernel's weight image
    float *raw;
    char *scratch;           // G2E_SCRATCH_PER_WG bytes per workgroup
    long M, ntiles;
    unsigned n, s1, magic, d64;
};

#ifdef BF16_STAMP  // diagnostic build only (mlp_bf16_g2.hip's layout): six values per workgroup -- s_memtime / s_memrealtime around its tile loop, tiles done, entry
__device__ unsigned long long *G2E_STAMP_VAR;
DDN_EXPORT int G2E_STAMP_SET(void *p) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(G2E_STAMP_VAR), &p, sizeof(p)); }
#endif

template <bool DEPTH_HEAD>
__global__ __launch_bounds__(256, 1) void G2E_KERNEL(G2EArgs a) {
    __shared__ __attribute__((aligned(16))) char lds[G2E_LDS_BYTES];
    const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned lds0 = __builtin_amdgcn_readfirstlane(lds_addr_of(lds));
    // raw buffers (stride 0, byte count, the raw-buffer format word): offsets past the end load zero / drop the store
    const char *sc = a.scratch + (size_t)blockIdx.x * G2E_SCRATCH_PER_WG;
    const u32x4 srs = {(unsigned)(size_t)sc, (unsigned)((size_t)sc >> 32) & 0xffffu, (unsigned)G2E_SCRATCH_PER_WG, 0x00020000u};
    const u32x4 rrs = {(unsigned)(size_t)a.raw, (unsigned)((size_t)a.raw >> 32) & 0xffffu, (unsigned)(a.M * (DEPTH_HEAD ? 24 : 16)), 0x00020000u};
    const u32x4 trs = {(unsigned)(size_t)a.t_vals, (unsigned)((size_t)a.t_vals >> 32) & 0xffffu, (unsigned)((size_t)a.n * a.s1 * 4), 0x00020000u};
    const u32x4 yrs = {(unsigned)(size_t)a.table, (unsigned)((size_t)a.table >> 32) & 0xffffu, (unsigned)((size_t)a.n * (G2E_TABLE_FLOATS * 4)), 0x00020000u};
    const unsigned plo = (unsigned)(size_t)a.packed, phi = (unsigned)((size_t)a.packed >> 32);
    const unsigned grid = gridDim.x, tile0 = blockIdx.x;
    const unsigned s1 = a.s1, magic = a.magic, nmax = a.n - 1, d64 = a.d64;
    const unsigned long tab = (unsigned long)(size_t)a.table;
#ifdef BF16_STAMP
    const unsigned long long st_entry = __builtin_amdgcn_s_memrealtime();
    const unsigned long long st_t0 = __builtin_amdgcn_s_memtime(), st_r0 = __builtin_amdgcn_s_memrealtime();
    unsigned done = 0;
#endif
#pragma clang loop unroll(disable)
    for (unsigned tile = tile0; tile < (unsigned)a.ntiles; tile += grid) {
        // (the encoder's stores: this workgroup's scratch, or nothing when there is no next tile to encode for)
        const u32x4 sst = {srs.x, srs.y, (unsigned long)tile + grid < (unsigned long)a.ntiles ? srs.z : 0u, srs.w};
        if constexpr (DEPTH_HEAD) {
            asm volatile(
#include G2E_BODY_D1
                :
                : "s"(srs), "s"(rrs), "s"(plo), "s"(phi), "s"(lds0), "s"(wave), "s"(tile), "s"(grid), "s"(tile0), "s"(trs), "s"(yrs), "s"(s1),
                  "s"(magic), "s"(nmax), "s"(d64), "s"(tab), "s"(sst)
                : G2E_CLOBBERS);
        } else {
            asm volatile(
#include G2E_BODY_D0
                :
                : "s"(srs), "s"(rrs), "s"(plo), "s"(phi), "s"(lds0), "s"(wave), "s"(tile), "s"(grid), "s"(tile0), "s"(trs), "s"(yrs), "s"(s1),
                  "s"(magic), "s"(nmax), "s"(d64), "s"(tab), "s"(sst)
                : G2E_CLOBBERS);
        }
#ifdef BF16_STAMP
        ++done;
#endif
    }
    // (the last tile issued the next one's chunks, rows and inputs: let them land before the wave ends)
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#ifdef BF16_STAMP
    {
        const unsigned long long st_t1 = __builtin_amdgcn_s_memtime(), st_r1 = __builtin_amdgcn_s_memrealtime();
        unsigned long long *sp = G2E_STAMP_VAR;
        if (sp && threadIdx.x == 0) {
            sp[6 * blockIdx.x + 0] = st_t0;
            sp[6 * blockIdx.x + 1] = st_r0;
            sp[6 * blockIdx.x + 2] = st_t1;
            sp[6 * blockIdx.x + 3] = st_r1;
            sp[6 * blockIdx.x + 4] = done;
            sp[6 * blockIdx.x + 5] = st_entry;
        }
    }
#endif
}

// (32-bit byte offsets into t_vals and the outputs, the next tile's included)
#define G2E_MAX_LAUNCH (1L << 22)

// scratch the forward needs: one row area per workgroup of the persistent grid
DDN_EXPORT size_t G2E_SYM(scratch_bytes)(void) { return (size_t)ddn_cu_count() * G2E_SCRATCH_PER_WG; }

extern "C" size_t G2E_G1_BYTES(int depth_head);

// rays are given as their table (ddnerf_ray_table); t_vals [n, S + 1]; packed: the image of ddnerf_mlp_bf16_pack; raw [n * S, 4 | 6].
// Cone rays only, S a multiple of 64 (a group of 64 samples lies on one ray), n * S <= 2^22: DDNERF_E_RANGE otherwise -- the caller
// then runs ddnerf_encode + ddnerf_mlp_bf16_forward, which produce the same bits.
DDN_EXPORT int G2E_SYM(forward)(const void *ray_table, const float *t_vals, const void *packed, int depth_head, float *raw,
                                              int n, int S, void *scratch, ddnerf_stream_t stream) {
    DDN_REQUIRE(ray_table && t_vals && packed && raw && scratch, DDNERF_E_ARG);
    DDN_REQUIRE(n > 0 && S > 0, DDNERF_E_ARG);
    DDN_REQUIRE(S % 64 == 0 && (long)n * S <= G2E_MAX_LAUNCH, DDNERF_E_RANGE);
    DDN_REQUIRE(ddn_aligned(ray_table, 128) && ddn_aligned(packed, 16) && ddn_aligned(raw, 16) && ddn_aligned(scratch, 16) && ddn_aligned(t_vals, 4),
                DDNERF_E_ALIGN);
    const long M = (long)n * S;
    const long ntiles = (M + G2E_TILE - 1) / G2E_TILE;
    const int n_cu = ddn_cu_count();
    const dim3 grid((unsigned)(ntiles < n_cu ? ntiles : n_cu));
    const size_t img = (G2E_G1_BYTES(depth_head) + 255) & ~(size_t)255;   // (the two-group kernel's image inside the common one)
    const unsigned d64 = (unsigned)(S / 64);
    G2EArgs a{t_vals, (const float *)ray_table, (const char *)packed + img, raw, (char *)scratch, M, ntiles, (unsigned)n, (unsigned)(S + 1),
              (unsigned)(((1ull << 31) + d64 - 1) / d64), d64};
    if (depth_head)
        hipLaunchKernelGGL(G2E_KERNEL<true>, grid, dim3(256), 0, (hipStream_t)stream, a);
    else
        hipLaunchKernelGGL(G2E_KERNEL<false>, grid, dim3(256), 0, (hipStream_t)stream, a);
    return ddn_launch_status();
}
