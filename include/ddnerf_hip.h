/*
 * ddnerf_hip.h -- C ABI of libddnerf_hip.so: DDNeRF's ray-march hot path as hand-written
 * HIP kernels for MI355X (gfx950 / CDNA4).
 *
 * The reference has no FFI of its own: its operator API for this path is the set of pure
 * tensor functions that models/models.py calls.  Each entry point below replaces
 * one of those functions (cited as path:line in the reference repository) and is what a
 * binding inside the reference would call instead (INTEGRATION.md shows the ctypes stub).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless marked [host]; tensors are dense, row-major, fp32
 *     unless stated; the library never allocates, never synchronises and never throws
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); work is enqueued on it
 *     and the caller keeps all buffers alive until the stream has passed them
 *   - random / linspace tensors (torch.rand, torch.randn, torch.linspace) are produced by the host
 *     framework and passed in, exactly like the reference draws them -- with ONE exception, a deliberate departure from
 *     ddnerf_composite_forward_keep_rng / ddnerf_dd_coarse_sample_forward draw the density noise of
 *     volume_rendering_utils.py:29-37 IN the kernel (Philox4x32-10 keyed by the seed and offset of torch's CUDA generator, which
 *     the host advances as a torch random kernel would: reproducible under torch.manual_seed, but not torch.randn's values);
 *     it saves a generator launch and a 3-MB tensor per pass on the render path.  A caller that needs the reference's exact
 *     draws (the parity tests do) passes a noise tensor to ddnerf_composite_forward instead
 *   - return value: 0 = success; DDNERF_E_* (<0) = rejected arguments; >0 = hipError_t of the launch
 *   - the library is stateless (thread-safe); weights are passed per call
 *
 * Ray row layout ("rays", [n,12], models/models.py:158):
 *      [origin 0:3 | direction 3:6 | radius 6 | near 7 | far 8 | unit view direction 9:12]
 * Feature row layout ("feat", [M,128], M = n*S):
 *      [IPE sin block 0:48 | IPE cos block 48:96 | view-dir encoding 96:123 | zeros 123:128]
 *      = the reference's `embedded` [M,123] (models/models.py:133) padded to 128 columns.
 * Flat parameter buffer ("params"): the network's parameters in registration order
 *      layers_xyz.0..7, fc_feat, fc_alpha, layers_dir.0, fc_rgb [, fc_mu_sigma], each weight
 *      [out][in] row-major followed by its bias (models/base_architectures.py:22-37, 83-99):
 *      612,740 floats (MipNeRFModel) / 612,998 floats (DepthMipNeRFModel).
 */
#ifndef DDNERF_HIP_H
#define DDNERF_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void *ddnerf_stream_t;

#define DDNERF_OK 0
#define DDNERF_E_ARG (-1)       /* null pointer / non-positive size */
#define DDNERF_E_RANGE (-2)     /* size outside what the kernel was built for */
#define DDNERF_E_ALIGN (-3)     /* pointer not aligned as required */
#define DDNERF_E_WORKSPACE (-4) /* workspace too small */

#define DDNERF_FEAT_LD 128          /* columns of a feature row */
#define DDNERF_PARAMS_FINE 612740   /* floats in MipNeRFModel's flat parameter buffer */
#define DDNERF_PARAMS_COARSE_DD 612998

int ddnerf_abi_version(void);
const char *ddnerf_error_string(int code);
/* How this library was built: ABI version, whether it is the product or the diagnostic (clock-stamp) build, and the experiment
 * switches the generator of the two-group bf16 kernel body ran with ("" for the product body).  Static string, never NULL. */
const char *ddnerf_build_info(void);

/* a1  GeneralMipNerfModel.get_rays_batches   models/models.py:144-162 */
int ddnerf_pack_rays(const float *origins, const float *directions, const float *radii, float near_, float far_,
                     float *rays, int n, ddnerf_stream_t stream);

/* a2  sample_first_cycle   models/samplers.py:30-62
 * t_lin [nc+1] = torch.linspace(0,1,nc+1); t_rand [n,nc+1] = torch.rand(...) or NULL (perturb off).
 * lindisp: 0 linear in depth, 1 linear in disparity, 2 = t_lin already holds the absolute depths of
 * get_combined_samples (models/samplers.py:6-27, dataset.combined_sampling_method). */
int ddnerf_sample_first_cycle(const float *rays, const float *t_lin, const float *t_rand, float *t_vals, int n, int nc,
                              int lindisp, ddnerf_stream_t stream);
/* a1 + a2 in ONE launch for a ray batch that is a single chunk: rays [n,12] and t_vals [n,nc+1], bit for bit the outputs of
 * ddnerf_pack_rays followed by ddnerf_sample_first_cycle. */
int ddnerf_pack_rays_first_cycle(const float *origins, const float *directions, const float *radii, float near_, float far_,
                                 const float *t_lin, const float *t_rand, float *rays, float *t_vals, int n, int nc, int lindisp,
                                 ddnerf_stream_t stream);

/* a3+a4+a5  cast_rays -> integrated_pos_enc, positional_encoding(view dirs), concat
 * general_utils/math_utils.py:7-166, general_utils/nerf_helpers.py:127-171, models/models.py:124-133
 * t_vals [n,S+1] -> feat [n*S,128].  ray_shape: 0 cone, 1 cylinder.  feat_dtype: 0 fp32 (natural column
 * order), 1 bf16, 2 fp16 (both with the columns in MFMA k-order, see ddnerf_mlp_bf16_forward / ddnerf_mlp_f16_forward). */
int ddnerf_encode(const float *rays, const float *t_vals, void *feat, int n, int S, int ray_shape, int feat_dtype,
                  ddnerf_stream_t stream);
/* a1 + a2 + (a3 + a4 + a5) in ONE launch for a ray batch that is a single chunk with no first-cycle jitter: ddnerf_pack_rays_first_cycle
 * (t_rand == NULL) followed by ddnerf_encode of its outputs.  rays [n,12] and t_vals [n,nc+1] are OUTPUTS (bit for bit that entry point's),
 * feat [n*nc,128] is the encoding of exactly those values. */
int ddnerf_encode_first_cycle(const float *origins, const float *directions, const float *radii, float near_, float far_,
                              const float *t_lin, int lindisp, float *rays, float *t_vals, void *feat, int n, int nc, int ray_shape,
                              int feat_dtype, ddnerf_stream_t stream);

/* a3 + a4 + a5 with the view-direction columns ONCE PER RAY, as the reference computes them (models/models.py:128-133: the ray's direction is
 * encoded once and broadcast over its samples): fp32 rows; feat [n*S,128] gets columns 0..95 (96..127 are left untouched), dirs [n,32] one row
 * per ray = what columns 96..127 of every row of that ray hold when ddnerf_encode writes them.  Read by ddnerf_mlp_f32_forward_rays /
 * ddnerf_mlp_x3_forward_rays (same outputs as the plain forwards on full rows; a quarter of the row bytes less, written and read). */
int ddnerf_encode_rays(const float *rays, const float *t_vals, float *feat, float *dirs, int n, int S, int ray_shape, ddnerf_stream_t stream);
int ddnerf_encode_first_cycle_rays(const float *origins, const float *directions, const float *radii, float near_, float far_,
                                   const float *t_lin, int lindisp, float *rays, float *t_vals, float *feat, float *dirs, int n, int nc,
                                   int ray_shape, ddnerf_stream_t stream);

/* a7  MipNeRFModel.forward / DepthMipNeRFModel.forward  models/base_architectures.py:40-61, 103-126
 * as ONE fused kernel (all 12/13 Linear layers, activations never leave the register file).
 *   step 1: repack the flat fp32 parameters into the kernel's LDS-image layout (once per weight update)
 *   step 2: forward: feat [M,128] -> raw [M,4] (rgb, alpha) or [M,6] (+ raw mu, raw sigma) */
size_t ddnerf_mlp_f32_packed_floats(int depth_head);
int ddnerf_mlp_f32_pack(const float *params, int depth_head, float *packed, ddnerf_stream_t stream);
int ddnerf_mlp_f32_forward(const float *feat, const float *packed, int depth_head, float *raw, long M,
                           ddnerf_stream_t stream);
/* ... with the view-direction columns from the per-ray table of ddnerf_encode_rays (S = samples per ray, M = n * S, M * S < 2^32, S > 1) */
int ddnerf_mlp_f32_forward_rays(const float *feat, const float *dirs, int S, const float *packed, int depth_head, float *raw, long M,
                                ddnerf_stream_t stream);

/* bf16-MFMA variant of the same network (bf16 operands, fp32 accumulation, fp32 biases and outputs).
 * feat: bf16 [M,128] as written by ddnerf_encode(feat_dtype=1), i.e. in MFMA "k-order": inside every 32
 * columns, position 8g + e (g = 0..3, e = 0..7) holds column 16(e>>2) + 4g + (e&3) -- the order in which two
 * 16x16 accumulator tiles re-enter the next layer's v_mfma_f32_16x16x32_bf16 as its B operand.
 * Two kernels stand behind these three entry points and produce the same bits for the same sample: the forward picks the one
 * whose tile rounds cost less at this launch size (the two-group kernel from 65,536 samples on whenever its 512-sample tiles do not
 * leave CUs idle that 256-sample tiles would use; DDNERF_BF16_G2_MIN=<samples> makes it a plain threshold), the weight image holds
 * both kernels' layouts. */
size_t ddnerf_mlp_bf16_packed_bytes(int depth_head);
int ddnerf_mlp_bf16_pack(const float *params, int depth_head, void *packed, ddnerf_stream_t stream);
int ddnerf_mlp_bf16_forward(const void *feat, const void *packed, int depth_head, float *raw, long M,
                            ddnerf_stream_t stream);
/* ... the one-group kernel (mlp_bf16.hip): a workgroup owns 256 samples, every wave 64; its own weight image */
size_t ddnerf_mlp_bf16g1_packed_bytes(int depth_head);
int ddnerf_mlp_bf16g1_pack(const float *params, int depth_head, void *packed, ddnerf_stream_t stream);
int ddnerf_mlp_bf16g1_forward(const void *feat, const void *packed, int depth_head, float *raw, long M, ddnerf_stream_t stream);
/* ... the two-group kernel (mlp_bf16_g2.hip, tile body generated as assembly by gen_bf16_g2.py): a workgroup owns 512 samples, every
 * wave two groups of 64; a layer's weights are staged into LDS once per tile and used by both groups (54 % of the L2 -> LDS weight
 * stream per sample); its own weight image */
size_t ddnerf_mlp_bf16g2_packed_bytes(int depth_head);
int ddnerf_mlp_bf16g2_pack(const float *params, int depth_head, void *packed, ddnerf_stream_t stream);
int ddnerf_mlp_bf16g2_forward(const void *feat, const void *packed, int depth_head, float *raw, long M, ddnerf_stream_t stream);

/* a3 + a4 + a5 + a7 in ONE kernel (bf16 tier): run_network, models/models.py:117-142 -- cast_rays, integrated_pos_enc, the view directions'
 * positional encoding, the concat and the network are one step there, and the encoded rows [n*S,128] never exist in memory here: the
 * two-group kernel above carries the encoder in the vector-ALU gaps of its MFMA stream (gen_bf16_g2.py, Gen(fused=True)).
 *   ddnerf_ray_table: rays [n,12] (ddnerf_pack_rays) -> one 128-byte row per RAY: origin, direction, radius^2, d^2, 1 - d^2/|d|^2 as
 *     fp32 (general_utils/math_utils.py:34-54), then the ray's 32 view-direction columns (general_utils/nerf_helpers.py:127-171) as a
 *     16-bit row in k-order (feat_dtype 1: bf16, 2: fp16); table: ddnerf_ray_table_bytes(n) bytes, 128-byte aligned.
 *   ddnerf_encode_mlp_bf16_forward: t_vals [n,S+1] -> raw [n*S, 4|6]; packed = the image of ddnerf_mlp_bf16_pack; scratch =
 *     ddnerf_encode_mlp_bf16_scratch_bytes() bytes the launch may overwrite (its workgroups' private row areas).  Cone rays, S a
 *     multiple of 64, n*S <= 2^22: otherwise DDNERF_E_RANGE, and the caller runs ddnerf_encode(feat_dtype 1) + ddnerf_mlp_bf16_forward,
 *     whose outputs these are BIT FOR BIT. */
size_t ddnerf_ray_table_bytes(int n);
int ddnerf_ray_table(const float *rays, int n, int feat_dtype, void *table, ddnerf_stream_t stream);   /* feat_dtype: 1 bf16, 2 fp16 (the row's 16-bit half) */
/* a1 + a2 + the table in ONE launch (the head of a one-chunk render pass): ddnerf_pack_rays_first_cycle with t_rand = NULL followed by
 * ddnerf_ray_table, bit for bit. */
int ddnerf_pack_rays_first_cycle_table(const float *origins, const float *directions, const float *radii, float near_, float far_,
                                       const float *t_lin, float *rays, float *t_vals, int feat_dtype, void *table, int n, int nc, int lindisp,
                                       ddnerf_stream_t stream);
size_t ddnerf_encode_mlp_bf16_scratch_bytes(void);
int ddnerf_encode_mlp_bf16_forward(const void *ray_table, const float *t_vals, const void *packed, int depth_head, float *raw, int n, int S,
                                   void *scratch, ddnerf_stream_t stream);
/* ... the fp16 tier's twin (table with feat_dtype 2, packed = the image of ddnerf_mlp_f16_pack): bit for bit ddnerf_encode(feat_dtype = 2) +
 * ddnerf_mlp_f16_forward */
size_t ddnerf_encode_mlp_f16_scratch_bytes(void);
int ddnerf_encode_mlp_f16_forward(const void *ray_table, const float *t_vals, const void *packed, int depth_head, float *raw, int n, int S,
                                  void *scratch, ddnerf_stream_t stream);

/* fp16-MFMA variant: the two bf16 kernels above built on v_mfma_f32_16x16x32_f16 / v_cvt_pk_f16_f32 (same rate, same registers, same
 * images, same schedule); feat: fp16 [M,128] in the same k-order, as written by ddnerf_encode(feat_dtype=2).  fp16 keeps 11
 * significant bits against bf16's 8 (operand rounding 8x smaller); biases, accumulation and outputs are fp32 as above.  Same
 * dispatch between the one-group and the two-group kernel, same bits from both. */
size_t ddnerf_mlp_f16_packed_bytes(int depth_head);
int ddnerf_mlp_f16_pack(const float *params, int depth_head, void *packed, ddnerf_stream_t stream);
int ddnerf_mlp_f16_forward(const void *feat, const void *packed, int depth_head, float *raw, long M, ddnerf_stream_t stream);
size_t ddnerf_mlp_f16g1_packed_bytes(int depth_head);
int ddnerf_mlp_f16g1_pack(const float *params, int depth_head, void *packed, ddnerf_stream_t stream);
int ddnerf_mlp_f16g1_forward(const void *feat, const void *packed, int depth_head, float *raw, long M, ddnerf_stream_t stream);
size_t ddnerf_mlp_f16g2_packed_bytes(int depth_head);
int ddnerf_mlp_f16g2_pack(const float *params, int depth_head, void *packed, ddnerf_stream_t stream);
int ddnerf_mlp_f16g2_forward(const void *feat, const void *packed, int depth_head, float *raw, long M, ddnerf_stream_t stream);

/* K2 "x3": the same network on the bf16 matrix cores at fp32-class accuracy -- every fp32 weight and activation is split
 * exactly into hi + lo bf16, three MFMAs per product (hi*hi + hi*lo + lo*hi), fp32 accumulation: outputs within ~1e-6
 * of an fp64 evaluation (exact-fp32 kernel 6e-8, plain bf16 6e-4).  feat is the fp32 [M,128] feature matrix in natural
 * column order (as for ddnerf_mlp_f32_forward); replaces the same reference functions.  Inference kernel (persistent,
 * v_mfma_f32_16x16x32_bf16; mlp_x3_fwd.hip). */
size_t ddnerf_mlp_x3_packed_bytes(int depth_head);
int ddnerf_mlp_x3_pack(const float *params, int depth_head, void *packed, ddnerf_stream_t stream);
int ddnerf_mlp_x3_forward(const float *feat, const void *packed, int depth_head, float *raw, long M,
                          ddnerf_stream_t stream);
/* ... with the view-direction columns from the per-ray table of ddnerf_encode_rays (as ddnerf_mlp_f32_forward_rays; one launch: M < 2^23) */
int ddnerf_mlp_x3_forward_rays(const float *feat, const float *dirs, int S, const void *packed, int depth_head, float *raw, long M,
                               ddnerf_stream_t stream);

/* x3 training kernels (same network; forward and backward-data chains in the same accuracy class): the forward that records what
 * the backward needs, the W^T images, and the fused backward-data pass -- drop-ins for ddnerf_mlp_f32_forward_train / _pack_t /
 * _backward_data.
 *   packed: the inference kernel's weight image (ddnerf_mlp_x3_pack): the training forward is the same kernel plus the records
 *   acts  a RECORD of ddnerf_mlp_act_rows() rows x ld samples (row map as for the fp32 kernels; ld = M rounded up to 128) as
 *         "bf16 row pairs": ddnerf_mlp_act_rows() / 2 * ld 32-bit words -- see ddnerf_mlp_x3_wgrad_pairs, the kernel that reads
 *         it -- NOT an fp32 matrix.  The weight gradients contract these bf16-rounded operands (fp32 accumulation).
 *   bits  [160, ld] uint16 sign words (relu' for the backward pass, 1 bit instead of 4 bytes per value): bit r of word
 *         bits[2 T + h][m] is set iff row 32 T + (r & 3) + 8 (r >> 2) + 4 h of `acts` is > 0 for sample m
 *   deltas a record like `acts`: every layer's pre-activation gradient, rows as in `acts`; rows 2432.. = d(raw) */
int ddnerf_mlp_x3_forward_train(const float *feat, const void *packed, int depth_head, float *raw, float *acts, void *bits,
                                long M, long ld, ddnerf_stream_t stream);
/* ... with the view-direction columns from the per-ray table `dirs` [M / S, 32] of ddnerf_encode_rays (columns 96..127 of `feat` are not
 * read), as ddnerf_mlp_x3_forward_rays takes them: the same outputs, record and sign words bit for bit.  M S < 2^32. */
int ddnerf_mlp_x3_forward_train_rays(const float *feat, const float *dirs, int S, const void *packed, int depth_head, float *raw,
                                     float *acts, void *bits, long M, long ld, ddnerf_stream_t stream);
size_t ddnerf_mlp_x3_packed_t_bytes(int depth_head);
int ddnerf_mlp_x3_pack_t(const float *params, int depth_head, void *packed_t, ddnerf_stream_t stream);
int ddnerf_mlp_x3_backward_data(const float *g_raw, const void *packed_t, const void *bits, int depth_head, float *deltas,
                                long M, long ld, ddnerf_stream_t stream);
/* The same two kernels with EXACT records (the strict mode of the x3 training tier): every layer's output / delta is recorded as blocked
 * hi/lo words -- the exact split of the fp32 value, (bf16 hi << 16) | bf16 lo at word ((m >> 4) * 2560 + row) * 16 + (m & 15), the
 * format of ddnerf_mlp_f32_forward_train's record build -- for ddnerf_mlp_x3_wgrad_packed (three MFMAs per product): fp32-class
 * parameter gradients at twice the record bytes.  Own weight images.  acts / deltas: 2560 x ld words. */
size_t ddnerf_mlp_x3e_packed_bytes(int depth_head);
int ddnerf_mlp_x3e_pack(const float *params, int depth_head, void *packed, ddnerf_stream_t stream);
int ddnerf_mlp_x3e_forward_train(const float *feat, const void *packed, int depth_head, float *raw, float *acts, void *bits,
                                 long M, long ld, ddnerf_stream_t stream);
size_t ddnerf_mlp_x3e_packed_t_bytes(int depth_head);
int ddnerf_mlp_x3e_pack_t(const float *params, int depth_head, void *packed_t, ddnerf_stream_t stream);
int ddnerf_mlp_x3e_backward_data(const float *g_raw, const void *packed_t, const void *bits, int depth_head, float *deltas,
                                 long M, long ld, ddnerf_stream_t stream);

/* a8  DD head, inline in DDNerfModel.predict  models/models.py:242-260, 266-273
 * raw6 [n,nc,6] -> mus, sigmas, left_tail, part_inside (raw sigma) and smoothed sigma/left/part, all [n,nc];
 * scal[4] = {mus_loss, sig_loss, mus_reg, sig_reg}.  workspace: ddnerf_dd_head_workspace_floats() floats. */
size_t ddnerf_dd_head_workspace_floats(int n, int nc);
int ddnerf_dd_head(const float *raw6, int n, int nc, float smooth, float dist_reg, float *mus, float *sigmas,
                   float *left, float *part, float *ssig, float *sleft, float *spart, float *scal, float *workspace,
                   ddnerf_stream_t stream);

/* a10  volume_render_radiance_field  general_utils/volume_rendering_utils.py:6-85 (+ cumprod_exclusive,
 * general_utils/nerf_helpers.py:43-64).  raw [n,S,ldr] (ldr 4 or 6: columns 0:3 rgb, 3 density).
 * noise [n,S] = randn*std or NULL; mus [n,S] or NULL (then cdisp may be NULL).
 * flags: bit0 white background, bit1 blender branch (weights[-1]+=1e-10, depth from normalised pdf).
 * outputs: rgb_map [n,3], disp [n], acc [n], weights [n,S], depth [n], cdisp [n], rgb [n,S,3] (may be NULL). */
#define DDNERF_COMP_WHITE_BKGD 1
#define DDNERF_COMP_BLENDER 2
#define DDNERF_COMP_DP_FILTER 4  /* ddnerf_composite_forward_keep: keep[i] = sum(weights[i]) > 1e-10 (dataset type "blender"), else 1 */
int ddnerf_composite_forward(const float *raw, int ldr, const float *t_vals, const float *rays, const float *noise,
                             const float *mus, int n, int S, int flags, float *rgb_map, float *disp, float *acc,
                             float *weights, float *depth, float *cdisp, float *rgb, ddnerf_stream_t stream);

/* a11  sample_pdf  models/samplers.py:64-121
 * u_base [ns]: perturb off -> torch.linspace(0,1,ns); on -> torch.arange(ns)*(1/ns).
 * rnd [n,ns] = torch.rand(n,ns) or NULL; then u = min(u_base + rnd/fp32(ns+1e-5), 0.9999). */
int ddnerf_sample_pdf(const float *bins, const float *weights, const float *u_base, const float *rnd, float *samples,
                      int n, int nc, int ns, int pdf_padding, ddnerf_stream_t stream);

/* a12  sample_pdf_with_mu_sigma  models/samplers.py:124-215
 * mus/sigmas/part/left: the (smoothed) per-bin quantities the caller passes (models/models.py:227-237).
 * u_base [ns]: perturb off -> torch.linspace(0,0.9999,ns); on -> torch.arange(ns)*(1/(ns-1)).
 * samples [n,ns] sorted; bins_ind [n,ns] int32 (may be NULL) = the bin index of each draw BEFORE the sort. */
int ddnerf_sample_pdf_mu_sigma(const float *bins, const float *weights, const float *mus, const float *sigmas,
                               const float *part, const float *left, const float *u_base, const float *rnd,
                               float near_, float far_, float *samples, int32_t *bins_ind, int n, int nc, int ns,
                               int pdf_padding, ddnerf_stream_t stream);

/* a13  estimate_dp_loss  models/dd_utils.py:6-78 (forward value).
 * t1 [n,nf+1], t0 [n,nc+1], w1 [n,nf] (fine weights), w0 [n,nc] (coarse weights), mus0/sig0/left0/part0 [n,nc].
 * blender != 0 applies the row filter sum(w1) > 1e-10, including the reference's un-filtered left_tails_0
 * gather (row r of the FILTERED set reads left0 row r).  loss[0] = kl_div(..., reduction='mean') over the kept
 * rows, 0 when no row is kept.  workspace: ddnerf_dp_loss_workspace_bytes(n) bytes.
 * reg_scal / loss_total (both or neither, may be NULL): reg_scal = the DD head's scal[4]; loss_total[0] then receives the
 * level-1 `dp_loss` record of models/models.py:287-289, loss * nf + mus_reg + sig_reg, from the same launch. */
size_t ddnerf_dp_loss_workspace_bytes(int n);
int ddnerf_dp_loss_forward(const float *t1, const float *t0, const float *w1, const float *w0, const float *mus0,
                           const float *sig0, const float *left0, const float *part0, int n, int nc, int nf,
                           int blender, float *loss, const float *reg_scal, float *loss_total, void *workspace,
                           ddnerf_stream_t stream);

/* ---- ray generation (the caller immediately upstream of the path) ------------------------------------------
 * get_ray_bundle  general_utils/nerf_helpers.py:67-125: cam2world [host] = 3x4 (or 4x4) row-major pose;
 * outputs origins/directions [H,W,3], radii [H,W] (= [H,W,1]) incl. the 1e-5 zero nudges. */
int ddnerf_ray_bundle(int H, int W, float focal, const float *cam2world_host, float *origins, float *directions,
                      float *radii, ddnerf_stream_t stream);
/* ndc_mipnerf_rays  data_utils/dataset_helpers.py:3-42: NDC warp of a bundle + radii from neighbouring origins */
int ddnerf_ndc_rays(int H, int W, float focal, float near_, const float *origins, const float *directions,
                    float *origins_ndc, float *directions_ndc, float *radii, ddnerf_stream_t stream);
/* switch_t_ndc_to_regular  data_utils/dataset_helpers.py:45-49 (called at train_model.py:227-228): depth[i] =
 * ndc_depth[i] * origins[i][2] / (directions[i][2] - ndc_depth[i] * directions[i][2]) + 1 over n = H*W pixels; origins / directions
 * are the REGULAR (un-warped) bundle [H,W,3] of the same view. */
int ddnerf_ndc_depth_to_regular(long n, const float *ndc_depth, const float *origins, const float *directions, float *depth,
                                ddnerf_stream_t stream);

/* ---- training: backward entry points --------------------------------------------------------------------
 * Gradient structure of the reference's graph: nothing flows through the samplers or the encoder;
 * the MLPs need parameter gradients only; compositing needs d(raw) from d(rgb_map) and (coarse level) d(weights);
 * the DD head from d(mus), d(sigmas) and its regularisers; the dp loss w.r.t. (w0, mus0, sig0). */

/* backward of volume_render_radiance_field w.r.t. raw.  g_weights may be NULL.  g_raw [n,S,ldr]: columns 0..3 are
 * written, columns >= 4 are zeroed (ddnerf_dd_head_backward then ADDS into columns 4,5). */
int ddnerf_composite_backward(const float *raw, int ldr, const float *t_vals, const float *rays, const float *noise,
                              int n, int S, int flags, const float *g_rgb_map, const float *g_weights, float *g_raw,
                              ddnerf_stream_t stream);

/* backward of the DD head: g_mus / g_sigmas [n,nc] (may be NULL), g_scal[4] = upstream gradients of
 * {mus_loss, sig_loss, mus_reg, sig_reg} (may be NULL); adds into g_raw6[..., 4:6]. */
int ddnerf_dd_head_backward(const float *raw6, int n, int nc, float dist_reg, const float *g_mus, const float *g_sigmas,
                            const float *g_scal, float *g_raw6, ddnerf_stream_t stream);

/* The logging records of models/models.py:292-295: mus / sigmas / smoothed sigmas where the level-0 pdf w / sum(w)
 * exceeds 0.1, flat in row-major order (what boolean indexing returns).  Outputs have capacity n*nc; *total receives the
 * number of selected bins (device int).  workspace: ddnerf_dd_records_workspace_bytes(n, nc) bytes, 4-byte aligned. */
size_t ddnerf_dd_records_workspace_bytes(int n, int nc);
int ddnerf_dd_records(const float *weights, const float *mus, const float *sigmas, const float *ssig, int n, int nc,
                      float *out_mus, float *out_sigmas, float *out_ssig, int *total, void *workspace,
                      ddnerf_stream_t stream);
/* The coarse pass of DDNerfModel behind the MLP on the render path (models/models.py:242-295) as TWO launches instead of five: (1) DD head +
 * compositing with the head's mus + flags / counts of the level-0 records, wave per ray; (2) the records' ordered writes + the
 * regularisers' final sums.  Arguments and outputs as for ddnerf_dd_head, ddnerf_composite_forward (flags; cdisp required) and
 * ddnerf_dd_records (rec_*: capacity n * nc; rec_total [1]); bit for bit their outputs at nc = 64 (elsewhere the regulariser
 * partial sums are grouped per 4 rays instead of per 256 elements).  workspace: ddnerf_dd_coarse_workspace_bytes(n, nc). */
size_t ddnerf_dd_coarse_workspace_bytes(int n, int nc);
int ddnerf_dd_coarse_forward(const float *raw6, const float *t_vals, const float *rays, const float *noise, int n, int nc, int flags,
                             float smooth, float dist_reg, float *mus, float *sigmas, float *left, float *part, float *ssig,
                             float *sleft, float *spart, float *scal, float *rgb_map, float *disp, float *acc, float *weights,
                             float *depth, float *cdisp, float *rec_mus, float *rec_sigmas, float *rec_ssig, int *rec_total,
                             void *workspace, ddnerf_stream_t stream);
/* ... and with the fine pass's fenceposts drawn in the SAME first launch (samples != NULL): sample_pdf_with_mu_sigma (models/samplers.py:124-215,
 * called at models/models.py:227-237 with this pass's returned weights, the head's mus and its smoothed sigmas / part-inside / left tails),
 * i.e. ddnerf_sample_pdf_mu_sigma's u_base [ns], rnd [n,ns] or NULL, near_, far_, pdf_padding -> samples [n,ns], bit for bit that entry
 * point's output.  samples == NULL: ddnerf_dd_coarse_forward.
 * noise == NULL and noise_std > 0: the kernel draws the density noise randn * noise_std (volume_rendering_utils.py:29-37) itself -- element
 * (ray, j) = Philox4x32-10 under (noise_seed, noise_offset) at counter noise_base + ray * nc + j, through Box-Muller: a pure function of its
 * arguments (ddnerf_debug_philox_normal materialises it), so no generator launch and no noise tensor precede the launch. */
int ddnerf_dd_coarse_sample_forward(const float *raw6, const float *t_vals, const float *rays, const float *noise, int n, int nc, int flags,
                                    float smooth, float dist_reg, float *mus, float *sigmas, float *left, float *part, float *ssig,
                                    float *sleft, float *spart, float *scal, float *rgb_map, float *disp, float *acc, float *weights,
                                    float *depth, float *cdisp, float *rec_mus, float *rec_sigmas, float *rec_ssig, int *rec_total,
                                    void *workspace, const float *u_base, const float *rnd, float near_, float far_, float *samples, int ns,
                                    int pdf_padding, unsigned long long noise_seed, unsigned long long noise_offset,
                                    unsigned long long noise_base, float noise_std, ddnerf_stream_t stream);
/* The fine pass: compositing + the dp loss's row filter (models/dd_utils.py:16: torch.sum(w1) > 1e-10 IS the compositing's
 * weight sum) in one launch; dp_workspace = a ddnerf_dp_loss_workspace_bytes(n) buffer, then handed to
 * ddnerf_dp_loss_forward_kept, which is ddnerf_dp_loss_forward without its first launch. */
int ddnerf_composite_forward_keep(const float *raw, int ldr, const float *t_vals, const float *rays, const float *noise,
                                  const float *mus, int n, int S, int flags, float *rgb_map, float *disp, float *acc, float *weights,
                                  float *depth, float *cdisp, void *dp_workspace, ddnerf_stream_t stream);
/* ... with the density noise drawn in the kernel when noise == NULL and noise_std > 0 (see ddnerf_dd_coarse_sample_forward) */
int ddnerf_composite_forward_keep_rng(const float *raw, int ldr, const float *t_vals, const float *rays, const float *noise,
                                      const float *mus, int n, int S, int flags, float *rgb_map, float *disp, float *acc,
                                      float *weights, float *depth, float *cdisp, void *dp_workspace, unsigned long long noise_seed,
                                      unsigned long long noise_offset, unsigned long long noise_base, float noise_std,
                                      ddnerf_stream_t stream);
/* out[i] = the noise value of element noise_base + i under (seed, offset, std), i in [0, count): what the two entry points above add to
 * the density of that element. */
int ddnerf_debug_philox_normal(float *out, long count, unsigned long long seed, unsigned long long offset, unsigned long long base,
                               float std, ddnerf_stream_t stream);
int ddnerf_dp_loss_forward_kept(const float *t1, const float *t0, const float *w1, const float *w0, const float *mus0,
                                const float *sig0, const float *left0, const float *part0, int n, int nc, int nf, float *loss,
                                const float *reg_scal, float *loss_total, void *workspace, ddnerf_stream_t stream);
/* backward of estimate_dp_loss w.r.t. (w0, mus0, sig0); g_loss[0] = upstream gradient of the scalar loss.
 * workspace as for the forward. */
int ddnerf_dp_loss_backward(const float *t1, const float *t0, const float *w1, const float *w0, const float *mus0,
                            const float *sig0, const float *left0, const float *part0, int n, int nc, int nf,
                            int blender, const float *g_loss, float *g_w0, float *g_mus, float *g_sig, void *workspace,
                            ddnerf_stream_t stream);

/* fp32 MLP training kernels.  acts / deltas: [ddnerf_mlp_act_rows()][ld] fp32, ld = M rounded up to 128, stored
 * TRANSPOSED (row = feature, column = sample): rows 256*l.. layers_xyz.l (l=0..7), 2048.. fc_feat, 2304.. layers_dir.0.
 *   forward_train : forward + records every layer's output in `acts`
 *   pack_t        : transposed weight images for the backward-data kernel (once per weight update)
 *   backward_data : g_raw [M,4|6] -> every layer's pre-activation gradient in `deltas` (one fused kernel)
 *   wgrad         : one weight-gradient job dW = deltas[drow0:+n_out] x acts[arow0:+n_in]^T over the sample axis on the
 *                   fp32 matrix cores, split over the samples with an order-fixed second-stage reduce (no atomics);
 *                   bias gradient = row sums of the delta rows.  rows 2432.. of `acts` hold the input features
 *                   transposed, rows 2432..2437 of `deltas` hold d(raw) transposed (head layers).
 * Row count of both matrices: ddnerf_mlp_act_rows(). */
size_t ddnerf_mlp_act_rows(void);
size_t ddnerf_mlp_f32_packed_t_floats(int depth_head);
int ddnerf_mlp_f32_pack_t(const float *params, int depth_head, float *packed_t, ddnerf_stream_t stream);
int ddnerf_mlp_f32_forward_train(const float *feat, const float *packed, int depth_head, float *raw, float *acts, long M,
                                 long ld, ddnerf_stream_t stream);
int ddnerf_mlp_f32_backward_data(const float *g_raw, const float *packed_t, const float *acts, int depth_head,
                                 float *deltas, long M, long ld, ddnerf_stream_t stream);
/* The same two kernels with `acts` / `deltas` as records of blocked hi/lo words (see ddnerf_mlp_x3_wgrad_packed below): the
 * exact-fp32 forward / backward arithmetic, recording each value's exact hi/lo split -- the operands of the packed weight-gradient
 * kernel.  (The fp32 [feature][sample] variants above feed ddnerf_mlp_f32_wgrad / ddnerf_mlp_x3_wgrad.) */
int ddnerf_mlp_f32_forward_train_rec(const float *feat, const float *packed, int depth_head, float *raw, float *acts, long M,
                                 long ld, ddnerf_stream_t stream);
int ddnerf_mlp_f32_backward_data_rec(const float *g_raw, const float *packed_t, const float *acts, int depth_head,
                                 float *deltas, long M, long ld, ddnerf_stream_t stream);
/* ... and with records of bf16 ROW PAIRS (the x3 training tier's format, see ddnerf_mlp_x3_wgrad_pairs: 1280 pair rows x ld words): the
 * exact-fp32 forward / backward-data arithmetic again, half the record bytes, weight gradients with one MFMA per product on bf16-rounded
 * operands -- an opt-in speed mode of the fp32 tier (DDNERF_WGRAD=pairs), not fp32-class. */
int ddnerf_mlp_f32_forward_train_recp(const float *feat, const float *packed, int depth_head, float *raw, float *acts, long M,
                                  long ld, ddnerf_stream_t stream);
int ddnerf_mlp_f32_backward_data_recp(const float *g_raw, const float *packed_t, const float *acts, int depth_head,
                                  float *deltas, long M, long ld, ddnerf_stream_t stream);
/* ... and with blocked records of the fp32 VALUES themselves (round 5): the layout of the hi/lo-word records -- element (row, sample m) at
 * word index ((m >> 4) * 2560 + row) * 16 + (m & 15) -- holding the value unsplit.  The weight gradients then run on
 * ddnerf_mlp_x3_wgrad_blocked(_skip), which makes the same hi / lo split once per landed slot: the same weight gradients bit for bit as the
 * hi/lo-word path, and these kernels' fp32 MFMA chains lose the 3.5 vector-ALU instructions per recorded element (each stops the chain).
 * `signs`: the SIGN record the forward also writes and the backward reads for its ReLU masks instead of the activations themselves (1 bit
 * instead of 32 per value): ddnerf_mlp_f32_sign_bytes(ld) bytes, 128-byte aligned; per 128-sample tile 32 KiB = [32-row block of
 * layers_xyz.0-7's outputs, 64][wave = 32 samples, 4][accumulator register r, 16] 64-bit lane masks (bit 32 h + j: row 32 block +
 * (r & 3) + 8 (r >> 2) + 4 h, sample 32 wave + j, is > 0), written by scalar stores and read by scalar loads.
 * `dirs` / `S` (forward; dirs may be NULL): the view-direction columns from the per-ray table [M / S, 32] of ddnerf_encode_rays instead of
 * columns 96..127 of the rows, as in ddnerf_mlp_f32_forward_rays (same outputs and records; M S < 2^32). */
size_t ddnerf_mlp_f32_sign_bytes(long ld);
int ddnerf_mlp_f32_forward_train_recf(const float *feat, const float *packed, int depth_head, float *raw, float *acts, void *signs,
                                  const float *dirs, int S, long M, long ld, ddnerf_stream_t stream);
int ddnerf_mlp_f32_backward_data_recf(const float *g_raw, const float *packed_t, const float *acts, const void *signs, int depth_head,
                                  float *deltas, long M, long ld, ddnerf_stream_t stream);
size_t ddnerf_mlp_f32_wgrad_workspace_floats(long M);
int ddnerf_mlp_f32_wgrad(const float *deltas, int drow0, int n_out, const float *acts, int arow0, int n_in, int n_in_used,
                         long M, long ld, float *dst, int dst_ld, int dst_col0, float *dst_bias, float *workspace,
                         ddnerf_stream_t stream);
/* Same contract on the bf16 matrix cores with every fp32 operand split exactly into hi + lo bf16 (three MFMAs per
 * product, fp32 accumulation): relative product error ~2^-16, HBM-bound instead of MFMA-bound. */
int ddnerf_mlp_x3_wgrad(const float *deltas, int drow0, int n_out, const float *acts, int arow0, int n_in, int n_in_used,
                         long M, long ld, float *dst, int dst_ld, int dst_col0, float *dst_bias, float *workspace,
                         ddnerf_stream_t stream);
/* The record-operand weight-gradient kernel of the fp32 tier: same contract, but `deltas` and `acts` are the RECORDS that
 * ddnerf_mlp_f32_forward_train_rec and ddnerf_mlp_f32_backward_data_rec write: 2560 rows x ld samples of "blocked hi/lo words".  Element
 * (row, sample m) is one 32-bit word (bf16 hi << 16) | bf16 lo, hi = bf16(x), lo = bf16(x - hi) (the value's exact split),
 * at word index ((m >> 4) * 2560 + row) * 16 + (m & 15): 16-sample blocks, the
 * rows of a block back to back.  A job's rows of a block are one contiguous run that travels HBM -> LDS by LDS-DMA and feeds
 * the MFMAs without a split pass (4.7 TB/s where the [row][sample] fp32 operands of ddnerf_mlp_x3_wgrad reach 3.3).  ld must
 * be a multiple of 32.  max_workgroups (0 = 256) caps the split-K width: with 128, two jobs enqueued on two streams (each
 * with its own workspace) share the chip and write half the partial slabs each.  At the default width the weight gradients
 * are bit-identical to ddnerf_mlp_x3_wgrad on the fp32 matrices the words were split from (another width is another -- equally
 * fixed -- summation order); bias sums add hi + lo (2^-17 relative per term).  Only samples 0 .. M - 1 are contracted: the words
 * of the pad columns M .. ld - 1 are never used, whatever they hold (NaN patterns included). */
int ddnerf_mlp_x3_wgrad_packed(const void *deltas, int drow0, int n_out, const void *acts, int arow0, int n_in, int n_in_used,
                                long M, long ld, float *dst, int dst_ld, int dst_col0, float *dst_bias, float *workspace,
                                int max_workgroups, ddnerf_stream_t stream);
/* The x3 training tier's weight gradients: the same jobs on records of "bf16 row pairs" (what ddnerf_mlp_x3_forward_train and
 * ddnerf_mlp_x3_backward_data write).  A record has ddnerf_mlp_act_rows() / 2 PAIR rows; the 32-bit word of (pair row p, sample m),
 * at word index ((m >> 4) * 1280 + p) * 16 + (m & 15), holds bf16(row 2p) in its low and bf16(row 2p + 1) in its high half
 * (round to nearest even) -- the packed conversion the training kernels make for their own next layer.  Half the bytes of the
 * hi/lo-word records and ONE bf16 MFMA per product (fp32 accumulation): the usual mixed-precision weight gradient, relative error
 * ~2^-9 per product, averaging down over the M samples of the contraction.  drow0 / arow0 / arow_a / arow_b are operand rows and
 * must be even; everything else as for ddnerf_mlp_x3_wgrad_packed(_skip), pad columns included.  ddnerf_mlp_x3_split_pairs:
 * fp32 [rows][ld] (rows, row0 even) -> rows row0 .. of such a record. */
int ddnerf_mlp_x3_wgrad_pairs(const void *deltas, int drow0, int n_out, const void *acts, int arow0, int n_in, int n_in_used,
                              long M, long ld, float *dst, int dst_ld, int dst_col0, float *dst_bias, float *workspace,
                              int max_workgroups, ddnerf_stream_t stream);
int ddnerf_mlp_x3_wgrad_pairs_skip(const void *deltas, int drow0, const void *acts, int arow_a, int arow_b, long M, long ld,
                                   float *dst, float *dst_bias, float *workspace, int max_workgroups, ddnerf_stream_t stream);
int ddnerf_mlp_x3_split_pairs(const float *x, int rows, long ld, int row0, void *record, ddnerf_stream_t stream);
/* The same jobs as ddnerf_mlp_x3_wgrad_packed(_skip) on blocked records of fp32 VALUES (what ddnerf_mlp_f32_forward_train_recf and
 * ddnerf_mlp_f32_backward_data_recf write): the kernel splits every value into bf16 hi / lo (round to nearest even, lo = bf16(x - hi)) as it
 * builds its MFMA fragments.  Weight gradients bit-identical to ddnerf_mlp_x3_wgrad_packed on the hi/lo-word records of the same values;
 * bias sums add the values.  Pad columns as there. */
int ddnerf_mlp_x3_wgrad_blocked(const void *deltas, int drow0, int n_out, const void *acts, int arow0, int n_in, int n_in_used,
                                long M, long ld, float *dst, int dst_ld, int dst_col0, float *dst_bias, float *workspace,
                                int max_workgroups, ddnerf_stream_t stream);
int ddnerf_mlp_x3_wgrad_blocked_skip(const void *deltas, int drow0, const void *acts, int arow_a, int arow_b, long M, long ld,
                                     float *dst, float *dst_bias, float *workspace, int max_workgroups, ddnerf_stream_t stream);
/* layers_xyz.5 (input cat(xyz, h4)) as ONE job over the same records: dst [256][352] = rows drow0.. of `deltas` against
 * cat(acts rows arow_a .. +96, acts rows arow_b .. +256); dst_bias [256]. */
int ddnerf_mlp_x3_wgrad_packed_skip(const void *deltas, int drow0, const void *acts, int arow_a, int arow_b, long M, long ld,
                                     float *dst, float *dst_bias, float *workspace, int max_workgroups, ddnerf_stream_t stream);
/* fp32 [rows][ld] ([feature][sample], ld a multiple of 16) -> rows row0 .. row0 + rows - 1 of such a record. */
int ddnerf_mlp_x3_split(const float *x, int rows, long ld, int row0, void *record, ddnerf_stream_t stream);

/* a15  the loss of one training iteration, train_model.py:156-172, as one launch (+ one for its gradient):
 *   loss = c0 * mse(rgb0, target) + c1 * mse(rgb1, target) + c_dp * mean(dp[0..n_dp))      (rgb1 may be NULL: one level; n_dp may be 0)
 * count = number of floats of an rgb tensor (3 n).  out [4] = {loss, mse0, mse1, mean dp}.  Backward: g [1] = the upstream gradient of
 * the loss (NULL = 1): g_rgb0 / g_rgb1 [count], g_dp [n_dp]. */
int ddnerf_train_loss_forward(const float *rgb0, const float *rgb1, const float *target, long count, const float *dp, int n_dp, float c0,
                              float c1, float c_dp, float *out, ddnerf_stream_t stream);
int ddnerf_train_loss_backward(const float *rgb0, const float *rgb1, const float *target, long count, int n_dp, float c0, float c1, float c_dp,
                               const float *g, float *g_rgb0, float *g_rgb1, float *g_dp, ddnerf_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* DDNERF_HIP_H */
