"""MI355X host-side mirror of the reference's `models.models` module surface.

`GeneralMipNerfModel(cfg, backbone="MipNeRFModel")` and `DDNerfModel(cfg)` are selected by name exactly like
the reference (`getattr(models, cfg.nerf.type)(cfg)`, train_model.py:70) and expose the same methods
(`run_iter`, `to`, `train`, `eval`, `load_weights_from_checkpoint`), attributes (`coarse`, `fine`, live `cfg`)
and output schema (reference models/models.py:9-184, 187-322).  The per-ray inner loop --
encode, MLP, DD head, compositing, hierarchical sampling, dp-loss -- runs in the hand-written HIP kernels of
libddnerf_hip.so; this file is control flow only and has no CPU fallback.

Random tensors are drawn through `self.rng` in the reference's order (per ray chunk: rand for the first-cycle
jitter, randn for the coarse compositing noise, rand for the resampling jitter, randn for the fine noise) so
a replaying generator reproduces the reference bit for bit in the parity tests."""
from __future__ import annotations

import torch

from . import base_architectures
from . import depth_analysis
from . import functions as F
from . import ops


class TorchRng:
    """Default random source: torch's generator of the device the rays live on."""

    def rand(self, shape, device):
        return torch.rand(shape, dtype=torch.float32, device=device)

    def randn(self, shape, device):
        return torch.randn(shape, dtype=torch.float32, device=device)

    def randn_scaled(self, shape, device, std):
        """randn * std in ONE kernel (the generator applies the scale: same values as the two-step form)"""
        return torch.empty(shape, dtype=torch.float32, device=device).normal_(0.0, float(std))

    def randn_scaled_pair(self, n, s0, s1, device, std):
        """the compositing noise of BOTH levels of a chunk from ONE generator launch: [n, s0] and [n, s1] (independent normals;
        only the order in which the generator's stream is consumed differs from two calls)"""
        both = torch.empty(n * (s0 + s1), dtype=torch.float32, device=device).normal_(0.0, float(std))
        return both[:n * s0].view(n, s0), both[n * s0:].view(n, s1)

    def kernel_noise(self, count, device, std):
        """`count` normals * std that the compositing kernels draw THEMSELVES (no generator launch, no noise tensor): the Philox (seed,
        offset) of torch's CUDA generator of `device`, whose offset is advanced by what a torch random kernel of that many elements
        would consume -- so torch.manual_seed reproduces the render, and later torch draws do not reuse the stream."""
        gen = torch.cuda.default_generators[device.index if device.index is not None else torch.cuda.current_device()]
        off = gen.get_offset()
        gen.set_offset(off + (count + 3) // 4 * 4)
        return ops.KernelNoise(gen.initial_seed(), off, 0, std)


# render-path launch folding (see DDNerfModel.predict); DDNERF_FUSE_RENDER=0 keeps the one-kernel-per-reference-function path
import os as _os

FUSE_RENDER = _os.environ.get("DDNERF_FUSE_RENDER", "1") != "0"
# bf16 and fp16 tiers: run_network (models/models.py:117-142) as ONE launch, the encoder inside the MLP kernel (ops.encode_mlp_bf16_forward;
# bit-identical to encode + MLP).  "all" both passes, "fine" the fine pass only, "0" never.
FUSE_ENCODER = _os.environ.get("DDNERF_FUSE_ENCODER", "all")
# fp32 / x3 tiers, inference -- and the fp32 tier's training forward --: the view-direction columns once per RAY (ops.encode_rays + mlp_*_forward_rays), as the reference computes them
# (models/models.py:128-133), instead of once per sample in the feature rows; same outputs.  DDNERF_RAY_DIRS=0: per-sample columns.
RAY_DIRS = _os.environ.get("DDNERF_RAY_DIRS", "1") != "0"
KERNEL_NOISE = _os.environ.get("DDNERF_KERNEL_NOISE", "1") != "0"   # (0: the compositing noise comes from a torch generator launch again)

_const_cache = {}


def _host_const(kind, a, b, steps, device):
    """torch.linspace / arange rows built on the CPU (the reference's CPU path) and cached on the device."""
    key = (kind, float(a), float(b), int(steps), str(device))
    t = _const_cache.get(key)
    if t is None:
        if kind == "linspace":
            t = torch.linspace(a, b, steps, dtype=torch.float32)
        else:  # arange(steps) * s, float32 semantics of `torch.arange(n) * python_float`
            t = torch.arange(steps) * a
        t = t.to(torch.float32).to(device)
        _const_cache[key] = t
    return t


def _combined_row(near, split, far, nc, device):
    """get_combined_samples (models/samplers.py:6-27): uniform depths up to `split` for the first half of the bins, then
    log-spaced ones up to far -- one row for every ray, built once on the CPU in fp32 like the reference's CPU path."""
    key = ("combined", near, split, far, nc, str(device))
    row = _const_cache.get(key)
    if row is None:
        t = torch.linspace(0.0, 1.0, nc // 2 + 1, dtype=torch.float32)
        uniform = near * (1.0 - t) + split * t
        max_d = torch.tensor(far, dtype=torch.float32)
        d = split * (1.0 - t) + max_d * t
        frac = torch.sort(1 - (torch.log2(d - split + 1) / torch.log2(max_d - split + 1)))[0]
        log_part = split + frac * (max_d - split)
        row = torch.cat((uniform, log_part[1:])).to(device)
        if row.shape[0] != nc + 1:
            raise ValueError("combined sampling needs an even num_coarse")
        _const_cache[key] = row
    return row


def get_minibatches(inputs, chunksize=1024 * 8):
    """general_utils/nerf_helpers.py:19-24"""
    return [inputs[i:i + chunksize] for i in range(0, inputs.shape[0], chunksize)]


class GeneralMipNerfModel(torch.nn.Module):
    """mip-NeRF: one shared MLP, two passes, `sample_pdf` resampling (models/models.py:9-184)."""

    def __init__(self, cfg, backbone="MipNeRFModel"):
        super().__init__()
        self.coarse = getattr(base_architectures, backbone)(
            hidden_size=cfg.nerf.coarse_hidden_size, max_ipe_deg=16, num_encoding_fn_dir=4, include_input_xyz=False,
            include_input_dir=True, use_viewdirs=True)
        self.fine = self.coarse
        self.cfg = cfg
        self.rng = TorchRng()
        self._set_mlp_dtype()

    def _set_mlp_dtype(self):
        dt = self.cfg.nerf.get("mlp_dtype", "fp32") if hasattr(self.cfg.nerf, "get") else "fp32"
        for net in {id(self.coarse): self.coarse, id(self.fine): self.fine}.values():
            net.mlp_dtype = dt

    # ---- public surface -----------------------------------------------------------------------------
    def run_iter(self, ray_origins, ray_directions, ray_rad, mode="train", depth_analysis_validation=False,
                 rgb_target=None):
        """models/models.py:40-73"""
        shape_rgb = ray_directions.shape
        shape_depth = ray_directions.shape[:-1]
        batches = self._rays_batches(ray_origins, ray_directions, ray_rad, mode, defer=True)
        if rgb_target is not None:
            rgb_targets = get_minibatches(rgb_target.reshape(-1, 3), chunksize=getattr(self.cfg.nerf, mode).chunksize)
        else:
            rgb_targets = [None for _ in batches]
        # Several chunks (a validation image is 40 of them): every chunk is ENQUEUED before the host waits for anything -- the one
        # data-dependent shape per chunk (the DD records' length) is collected after the loop, not inside it, so the GPU never
        # drains between chunks -- and every output key is concatenated ONCE over all chunks (the reference's pairwise torch.cat in
        # a loop, :53-61, copies the growing prefix again for every chunk; the result is the same tensor).
        self._deferred_records = [] if len(batches) > 1 else None
        try:
            pred = [self.predict(b, mode, depth_analysis_validation, t) for b, t in zip(batches, rgb_targets)]
            for finish in (self._deferred_records or []):
                finish()
        finally:
            self._deferred_records = None
            self._ray_table = None         # (the fused bf16 path's per-ray table of the last chunk: it belongs to this call's rays)
            self._flush_first_pending()    # (no-op: predict's first run_network filled the packed rows it was handed)
        output = pred[0]
        if len(pred) > 1:
            for j in range(len(output)):
                for key in list(output[j].keys()):
                    vs = [p[j][key] for p in pred[1:] if key in p[j] and (p[j][key] is not None) and (p[j][key] is not False)]
                    if vs:
                        output[j][key] = torch.cat([output[j][key]] + vs, dim=0)
        if mode == "validation" and not depth_analysis_validation:
            for i in range(len(output)):
                output[i]["rgb"] = output[i]["rgb"].view(shape_rgb)
                for k in ("disp", "acc", "depth"):
                    output[i][k] = output[i][k].view(shape_depth)
                if output[i].get("corrected_disp_map") is not None:
                    output[i]["corrected_disp_map"] = output[i]["corrected_disp_map"].view(shape_depth)
        return output

    def get_rays_batches(self, ray_origins, ray_directions, ray_rad, mode):
        """models/models.py:144-162 (K0 pack kernel, then views of <= chunksize rays).  The rows are filled when this returns."""
        return self._rays_batches(ray_origins, ray_directions, ray_rad, mode, defer=False)

    def _rays_batches(self, ray_origins, ray_directions, ray_rad, mode, defer):
        """get_rays_batches; defer (run_iter ONLY): a one-chunk batch without jitter is handed out EMPTY and filled by the launch that
        encodes its coarse samples -- predict's first step -- or, should anything come between (an exception included), by
        run_iter's `finally: _flush_first_pending()`.  No public caller can obtain unfilled rows."""
        _defer = defer
        self._t0_ready = None
        self._first_pending = None
        mc = self._mode_cfg(mode)
        n = ray_directions.numel() // 3
        if (FUSE_RENDER and n <= mc.chunksize and not mc.perturb and not self.cfg.dataset.get("combined_sampling_method", False)):
            # ONE chunk and no jitter draw: the packed rows and the first-cycle fenceposts are written by the launch that ENCODES the
            # coarse samples (ops.encode_first_cycle, run_network below: the same arithmetic as pack_rays + sample_first_cycle + encode,
            # one launch instead of three).  The two tensors are handed out here and filled there: everything that reads them is
            # enqueued behind that launch (predict's first step is run_network on exactly this batch).
            t_lin = _host_const("linspace", 0.0, 1.0, mc.num_coarse + 1, ray_directions.device)
            if not _defer:   # (a caller other than run_iter gets the rows filled: pack + first cycle in their own launch)
                rays, t0 = ops.pack_rays_first_cycle(ray_origins, ray_directions, ray_rad, self.cfg.dataset.near, self.cfg.dataset.far, t_lin,
                                                     None, bool(mc.lindisp))
                self._t0_ready = (rays.data_ptr(), t0)
                return [rays]
            dev = ray_directions.device
            rays = torch.empty((n, 12), dtype=torch.float32, device=dev)
            t0 = torch.empty((n, mc.num_coarse + 1), dtype=torch.float32, device=dev)
            self._t0_ready = (rays.data_ptr(), t0)
            self._first_pending = (rays, t0, ray_origins, ray_directions, ray_rad, t_lin, bool(mc.lindisp))
            return [rays]
        rays = ops.pack_rays(ray_origins, ray_directions, ray_rad, self.cfg.dataset.near, self.cfg.dataset.far)
        return get_minibatches(rays, chunksize=getattr(self.cfg.nerf, mode).chunksize)

    def to(self, device):
        self.coarse.to(device)
        self.fine.to(device)

    def load_weights_from_checkpoint(self, checkpoint):
        self.coarse.load_state_dict(checkpoint["model_1_state_dict"])
        if self.cfg.nerf.type != "GeneralMipNerfModel":
            self.fine.load_state_dict(checkpoint["model_2_state_dict"])

    def train(self):
        self.coarse.train()
        self.fine.train()

    def eval(self):
        self.coarse.eval()
        self.fine.eval()

    # ---- pieces of predict ---------------------------------------------------------------------------
    def _mode_cfg(self, mode):
        return getattr(self.cfg.nerf, mode)

    def _is_blender(self):
        d = self.cfg.dataset
        return str(d.type).lower() == "blender" or str(d.basedir).endswith("segmented")  # volume_rendering_utils.py:51

    def _first_cycle(self, rays, mode):
        """models/samplers.py:30-62"""
        mc = self._mode_cfg(mode)
        nc = mc.num_coarse
        ready = getattr(self, "_t0_ready", None)
        if ready is not None and ready[0] == rays.data_ptr() and ready[1].shape == (rays.shape[0], nc + 1):
            self._t0_ready = None
            return ready[1]          # (get_rays_batches computed them with the packed rows)
        t_rand = self.rng.rand((rays.shape[0], nc + 1), rays.device) if mc.perturb else None
        if self.cfg.dataset.get("combined_sampling_method", False):   # models/samplers.py:45-49
            # `far[0]` of the reference is the packed rows' far column, i.e. cfg.dataset.far.  Like the reference (a bare try / except
            # around this branch, :44-51) anything that goes wrong here -- no dataset.combined_split, an odd num_coarse -- silently
            # leaves the linear fenceposts in place.
            try:
                row = _combined_row(float(self.cfg.dataset.near), float(self.cfg.dataset.combined_split), float(self.cfg.dataset.far), nc,
                                    rays.device)
            except Exception:
                row = None
            if row is not None:
                return ops.sample_first_cycle(rays, row, t_rand, 2)
        t_lin = _host_const("linspace", 0.0, 1.0, nc + 1, rays.device)
        return ops.sample_first_cycle(rays, t_lin, t_rand, bool(mc.lindisp))

    def _noise(self, n, S, mode, device):
        std = self._mode_cfg(mode).radiance_field_noise_std
        if std > 0.0:  # volume_rendering_utils.py:29-37
            scaled = getattr(self.rng, "randn_scaled", None)
            return scaled((n, S), device, std) if scaled is not None else self.rng.randn((n, S), device) * std
        return None

    def _flush_first_pending(self):
        """fill the ray rows / fenceposts get_rays_batches handed out, without encoding (they were not the next thing encoded)"""
        pend, self._first_pending = getattr(self, "_first_pending", None), None
        if pend is not None:
            rays, t0, ro, rd, rad, t_lin, lindisp = pend
            r2, t2 = ops.pack_rays_first_cycle(ro, rd, rad, self.cfg.dataset.near, self.cfg.dataset.far, t_lin, None, lindisp)
            rays.copy_(r2)
            t0.copy_(t2)

    def run_network(self, ray_batch, t_vals, network, mode):
        """models/models.py:117-142: encode (K1) + fused MLP (K2); [n,S,4|6]"""
        kind = network.mlp_dtype if network.mlp_dtype in ("bf16", "fp16") else "fp32"   # (the 16-bit kernels read 16-bit k-order rows)
        shape = str(self.cfg.nerf.ray_shape)
        if shape not in ("cone", "cylinder"):
            raise AssertionError("ray_shape must be 'cone' or 'cylinder'")  # math_utils.py:28
        pend = getattr(self, "_first_pending", None)
        first = pend is not None and pend[0].data_ptr() == ray_batch.data_ptr() and pend[1].data_ptr() == t_vals.data_ptr()
        n, S = t_vals.shape[0], t_vals.shape[1] - 1
        if (kind in ("bf16", "fp16") and shape == "cone" and FUSE_ENCODER != "0" and (FUSE_ENCODER == "all" or not first)
                and ops.encode_mlp_bf16_supported(S, n * S) and not F.needs_grad(network)):
            # the fused kernel: the encoded rows never exist; a per-RAY table (built once per ray batch, by the launch that packs the rays
            # when this is the coarse pass of a one-chunk batch) carries what the encoder derives from a ray
            if first:
                self._first_pending = None
                rays, t0, ro, rd, rad, t_lin, lindisp = pend
                table = ops.pack_rays_first_cycle_table(ro, rd, rad, self.cfg.dataset.near, self.cfg.dataset.far, t_lin, lindisp, out=(rays, t0), kind=kind)[2]
            else:
                if pend is not None:
                    self._flush_first_pending()
                held = getattr(self, "_ray_table", None)
                table = held[1] if held is not None and held[0] is ray_batch else ops.ray_table(ray_batch, kind)
            self._ray_table = (ray_batch, table)
            raw = F.encode_mlp_bf16(table, t_vals, network)
            return raw.reshape(n, S, raw.shape[-1])
        ray_dirs = (RAY_DIRS and kind == "fp32" and network.mlp_dtype in ("fp32", "x3") and ops.mlp_rays_supported(S, n * S)
                    and (not F.needs_grad(network) or F.mlp_rays_trainable(network)))   # (training: the fp32 tier's values-record kernels, round 5)
        dirs = None
        if first:
            # the coarse pass of a one-chunk batch: this launch also fills ray_batch and t_vals (get_rays_batches handed them out empty)
            self._first_pending = None
            rays, t0, ro, rd, rad, t_lin, lindisp = pend
            if ray_dirs:
                feat, dirs = ops.encode_first_cycle_rays(ro, rd, rad, self.cfg.dataset.near, self.cfg.dataset.far, t_lin, lindisp,
                                                         cylinder=(shape == "cylinder"), out=(rays, t0))[2:]
            else:
                feat = ops.encode_first_cycle(ro, rd, rad, self.cfg.dataset.near, self.cfg.dataset.far, t_lin, lindisp,
                                              cylinder=(shape == "cylinder"), kind=kind, out=(rays, t0))[2]
        else:
            if pend is not None:     # (something else is encoded first: fill the pending tensors the plain way)
                self._flush_first_pending()
            if ray_dirs:
                feat, dirs = ops.encode_rays(ray_batch, t_vals, cylinder=(shape == "cylinder"))
            else:
                feat = ops.encode(ray_batch, t_vals, cylinder=(shape == "cylinder"), kind=kind)
        raw = F.mlp_rays(feat, dirs, S, network) if ray_dirs else F.mlp(feat, network)
        return raw.reshape(t_vals.shape[0], t_vals.shape[1] - 1, raw.shape[-1])

    def predict(self, ray_batch, mode, depth_analysis_validation, rgb_target=None):
        """models/models.py:75-114"""
        mc = self._mode_cfg(mode)
        n = ray_batch.shape[0]
        ret = {}
        weights = t_vals = None
        for i in range(2):
            if i == 0:
                t_vals = self._first_cycle(ray_batch, mode)
            else:
                ns = mc.num_fine + 1
                det = (mc.perturb == 0.0)
                if det:
                    u_base, rnd = _host_const("linspace", 0.0, 1.0, ns, ray_batch.device), None
                else:
                    u_base = _host_const("arange", 1 / ns, 0.0, ns, ray_batch.device)
                    rnd = self.rng.rand((n, ns), ray_batch.device)
                t_vals = ops.sample_pdf(t_vals, weights.detach(), u_base, rnd, bool(self.cfg.train_params.pdf_padding))
            raw = self.run_network(ray_batch, t_vals, self.coarse, mode)
            c = F.composite(raw, t_vals, ray_batch, self._noise(n, t_vals.shape[1] - 1, mode, ray_batch.device), None,
                            bool(mc.white_background), self._is_blender())
            weights = c["weights"]
            ret[i] = {"rgb": c["rgb_map"], "disp": c["disp"], "acc": c["acc"], "weights": weights, "depth": c["depth"]}
            if depth_analysis_validation:  # :108-112 (plots of a few rays)
                ret[i]["uniform_incell_pdf_to_plot"] = depth_analysis.uniform_incell_pdf(t_vals, weights, self.cfg.dataset.near,
                                                                                       self.cfg.dataset.far)
                ret[i]["t_vals_for_plot"] = t_vals
        return ret


class DDNerfModel(GeneralMipNerfModel):
    """DDNeRF: coarse net with a depth-distribution head, separate fine net, truncated-Gaussian resampling and
    the dp (KL) loss (models/models.py:187-322)."""

    def __init__(self, cfg):
        GeneralMipNerfModel.__init__(self, cfg, backbone="DepthMipNeRFModel")
        try:
            hidden_size_fine = cfg.nerf.fine_hidden_size
        except (AttributeError, KeyError):
            print("no nidden size params for fine model, set 256")
            hidden_size_fine = 256
        self.fine = base_architectures.MipNeRFModel(
            hidden_size=hidden_size_fine, max_ipe_deg=16, num_encoding_fn_dir=4, include_input_xyz=False,
            include_input_dir=True, use_viewdirs=True)
        self._set_mlp_dtype()

    def predict(self, ray_batch, mode, depth_analysis_validation, rgb_target=None):
        """models/models.py:207-322"""
        cfg, mc = self.cfg, self._mode_cfg(mode)
        n, dev = ray_batch.shape[0], ray_batch.device
        blender = self._is_blender()
        ret = {}
        model = self.coarse
        # one generator launch for the compositing noise of both levels where the random source offers it (TorchRng does; a replaying
        # source in the parity tests does not and is asked level by level, in the reference's draw order)
        pair = getattr(self.rng, "randn_scaled_pair", None)
        kernel_noise = getattr(self.rng, "kernel_noise", None)
        noise_pair = None
        t_vals_1 = None
        if FUSE_RENDER and mc.radiance_field_noise_std > 0.0 and not torch.is_grad_enabled() and kernel_noise is not None and KERNEL_NOISE:
            # render path with the default random source: the two compositing launches draw their noise themselves
            kn = kernel_noise(n * (mc.num_coarse + mc.num_fine), dev, mc.radiance_field_noise_std)
            noise_pair = (kn.at(0), kn.at(n * mc.num_coarse))
        elif FUSE_RENDER and pair is not None and mc.radiance_field_noise_std > 0.0:
            noise_pair = pair(n, mc.num_coarse, mc.num_fine, dev, mc.radiance_field_noise_std)
        for i in range(2):
            if i == 1:
                model = self.fine
                mus = None
            def resampling_draw():   # models/samplers.py:155-171: the u grid of the fine pass (and, with perturb, its jitter draw)
                ns = mc.num_fine + 1
                if mc.perturb == 0.0:
                    return _host_const("linspace", 0.0, 0.9999, ns, dev), None
                return _host_const("arange", 1 / (ns - 1), 0.0, ns, dev), self.rng.rand((n, ns), dev)

            if i == 0:
                t_vals = self._first_cycle(ray_batch, mode)
            elif t_vals_1 is not None:   # (the coarse pass's fused launch drew them already)
                t_vals = t_vals_1
            else:
                u_base, rnd = resampling_draw()
                # a fresh leaf in the reference (nn.Parameter): nothing flows back through the sampler
                t_vals = ops.sample_pdf_mu_sigma(t_vals_0, weights_0.detach(), mus_0.detach(), head["ssig"].detach(),
                                                 head["spart"].detach(), head["sleft"].detach(), u_base, rnd,
                                                 cfg.dataset.near, cfg.dataset.far, bool(cfg.train_params.pdf_padding))
            raw = self.run_network(ray_batch, t_vals, model, mode)
            if noise_pair is not None:
                noise = noise_pair[i]
            else:
                noise = self._noise(n, t_vals.shape[1] - 1, mode, dev)
            # Render path (nothing to differentiate): the small kernels behind each MLP are folded -- coarse: DD head + compositing +
            # record flags in one launch, the records' writes + the regularisers' sums in a second (five before); fine: compositing +
            # the dp loss's row filter in one.  Same arithmetic, same outputs (ops.dd_coarse_forward / composite_forward_keep).
            fused = FUSE_RENDER and not (torch.is_grad_enabled() and raw.requires_grad)
            dp_ws = None
            if i == 0 and fused:
                # ... and the fine pass's fenceposts in the same launch (the resampling jitter is drawn here, behind the coarse noise and
                # ahead of the fine noise: the reference's draw order)
                u_base, rnd = resampling_draw()
                c, head, records, t_vals_1 = ops.dd_coarse_forward(
                    raw, t_vals, ray_batch, noise, cfg.train_params.gaussian_smooth_factor, cfg.train_params.dist_reg_coeficient,
                    bool(mc.white_background), blender,
                    sample=(u_base, rnd, cfg.dataset.near, cfg.dataset.far, bool(cfg.train_params.pdf_padding)))
            elif i == 0:
                head = F.dd_head(raw, cfg.train_params.gaussian_smooth_factor, cfg.train_params.dist_reg_coeficient)
            if i == 0:
                mus, sigmas = head["mus"], head["sigmas"]
                smoothed_sigmas = head["ssig"]
                scal = head["scal"]
                mus_loss, sig_loss, mus_reg, sig_reg = scal[0], scal[1], scal[2], scal[3]
            dp_blender = str(cfg.dataset.type).lower() == "blender"
            if i == 1 and fused:
                c, dp_ws = ops.composite_forward_keep(raw, t_vals, ray_batch, noise, mus, bool(mc.white_background), blender, dp_blender)
            elif not (i == 0 and fused):
                c = F.composite(raw, t_vals, ray_batch, noise, mus, bool(mc.white_background), blender)
            weights = c["weights"]
            if i == 0:
                t_vals_0, mus_0, sigmas_0, weights_0 = t_vals, mus, sigmas, weights
                # models/models.py:292-295: mus / sigmas / smoothed sigmas where the level-0 pdf exceeds 0.1.  Boolean
                # indexing has a data-dependent size, i.e. a host sync.  ONE stream compaction serves the three records; it
                # is enqueued here, as soon as its inputs exist, with an asynchronous copy of the length: when the host
                # asks for it (end of the chunk) the GPU is still busy with the fine pass, so it never runs dry.
                if not fused:
                    with torch.no_grad():
                        records = ops.dd_records_launch(weights_0, mus_0, sigmas_0, smoothed_sigmas)
            dp_loss = None
            if i == 1:
                dp_args = (t_vals.detach(), t_vals_0.detach(), weights.detach(), weights_0, mus_0, sigmas_0,
                           head["left"].detach(), head["part"].detach(), dp_blender)
                if torch.is_grad_enabled() and (weights_0.requires_grad or mus_0.requires_grad or sigmas_0.requires_grad):
                    dp = F.dp_loss(*dp_args)
                    dp_loss = (dp * (t_vals.shape[1] - 1) + mus_reg + sig_reg).unsqueeze(0)      # :287-289
                elif dp_ws is not None:  # the row filter is in the workspace already (the fine compositing launch wrote it)
                    dp_loss = ops.dp_loss_forward_kept(*dp_args[:-1], dp_ws, scal)[1]
                else:  # nothing to differentiate: the record comes out of the loss kernel's last launch
                    dp_loss = ops.dp_loss_forward(*dp_args, reg_scal=scal)[1]
            # level 1 records the stale level-0 tensors under the level-0 mask, as the reference does (:297-300); the
            # three logging-only records are filled in below, after the fine pass has been enqueued
            ret[i] = {"rgb": c["rgb_map"], "disp": c["disp"], "acc": c["acc"], "weights": weights, "depth": c["depth"],
                      "mus": None, "sigmas": None, "dp_loss": dp_loss,
                      "corrected_disp_map": c["cdisp"], "smoothed_sigmas": None}
            if i == 0:
                ret[i]["mus_loss"] = mus_loss.unsqueeze(0)
                ret[i]["sig_loss"] = sig_loss.unsqueeze(0)
                ret[i]["mus_reg"] = mus_reg.unsqueeze(0)
                ret[i]["sig_reg"] = sig_reg.unsqueeze(0)
            if depth_analysis_validation:  # :307-319 (plots of a few rays)
                near, far = cfg.dataset.near, cfg.dataset.far
                ret[i]["uniform_incell_pdf_to_plot"] = depth_analysis.uniform_incell_pdf(t_vals, weights, near, far)
                ret[i]["t_vals_for_plot"] = t_vals
                if i == 1:
                    ret[i]["gaussian_incell_pdf_to_plot"] = depth_analysis.gaussian_incell_pdf(
                        t_vals_0, weights_0, mus_0, sigmas_0, head["part"], near, far)
                    ret[i]["smoothed_gaussian_incell_pdf_to_plot"] = depth_analysis.gaussian_incell_pdf(
                        t_vals_0, weights_0, mus_0, smoothed_sigmas, head["spart"], near, far)
        def finish_records():
            r_mus, r_sig, r_ssig = ops.dd_records_finish(records)
            rec = {"mus": r_mus, "sigmas": r_sig, "smoothed_sigmas": r_ssig}
            for i in range(2):
                ret[i].update(rec)

        deferred = getattr(self, "_deferred_records", None)
        if deferred is not None:   # run_iter over several chunks: it collects the lengths when every chunk is in the queue
            deferred.append(finish_records)
        else:
            finish_records()
        return ret
