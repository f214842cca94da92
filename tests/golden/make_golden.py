#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by IMPORTING the reference.

Run only in the build container (``/root/reference`` does not exist on the GPU
box):  ``cd /tmp && PYTHONDONTWRITEBYTECODE=1 python /root/repo/tests/golden/make_golden.py``

Nothing of the reference is copied: the script imports its modules from where
they lie, feeds them seeded synthetic inputs (``ddnerf_amd/synthetic.py``) and
stores inputs + outputs (+ captured internals) as small ``.npz`` files.
Oracle here = reference source + torch 2.10.0 CPU.  Every random tensor the
reference draws is reproduced by re-seeding torch and replaying the same draw,
so the fixtures carry them explicitly.
"""
import importlib.util
import os
import sys

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
# the reference's top-level packages are called `models`, `general_utils`; make sure
# they win over this repo's drop-in `models` alias
sys.path = [REF] + [p for p in sys.path if os.path.abspath(p or ".") != REPO]
sys.dont_write_bytecode = True

import numpy as np  # noqa: E402
import torch  # noqa: E402
import yaml  # noqa: E402

spec = importlib.util.spec_from_file_location("synthetic", os.path.join(REPO, "ddnerf_amd", "synthetic.py"))
synthetic = importlib.util.module_from_spec(spec)
spec.loader.exec_module(synthetic)

from general_utils.cfgnode import CfgNode  # noqa: E402
from general_utils import math_utils as ref_math  # noqa: E402
from general_utils import nerf_helpers as ref_helpers  # noqa: E402
from general_utils import volume_rendering_utils as ref_vr  # noqa: E402
from models import models as ref_models  # noqa: E402
from models import samplers as ref_samplers  # noqa: E402
from models import dd_utils as ref_dd  # noqa: E402

import warnings  # noqa: E402

warnings.filterwarnings("ignore")
torch.set_num_threads(8)

T = torch.from_numpy


def npy(x):
    if x is None:
        return None
    if isinstance(x, torch.Tensor):
        return x.detach().cpu().numpy()
    return np.asarray(x)


def save(name, **arrs):
    arrs = {k: npy(v) for k, v in arrs.items() if v is not None}
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrs)
    print("%-40s %8.1f KB" % (name, os.path.getsize(path) / 1024))


def load_cfg(fname, nc, nf, kind=None, **over):
    cfg = CfgNode(yaml.load(open(os.path.join(REF, "configs", fname)), Loader=yaml.FullLoader))
    for mode in ("train", "validation"):
        cfg.nerf[mode]["num_coarse"] = nc
        cfg.nerf[mode]["num_fine"] = nf
    if kind == "real360":  # normalize_poses rescaling done by the data loader (data_utils/data_utils.py:67-74)
        cfg.dataset.near = cfg.dataset.near / cfg.dataset.normalize_factor
        cfg.dataset.far = cfg.dataset.far / cfg.dataset.normalize_factor
    for k, v in over.items():
        sec, key = k.split("__")
        setattr(getattr(cfg, sec), key, v)
    return cfg


def load_weights(module, depth_head, seed, sharpen=1.0):
    sd = synthetic.make_state_dict(depth_head, seed, sharpen)
    module.load_state_dict({k: T(v) for k, v in sd.items()})


CFG_OF = {"blender": "config_blender.yml", "llff": "config_ff.yml", "real360": "config_360.yml"}
CFG_MIP_OF = {"blender": "config_blender_mipnerf.yml", "llff": "config_ff_mipnerf.yml", "real360": "config_360_mipnerf.yml"}


# ----------------------------------------------------------------------------------------------
# 1. first-cycle sampler (a2)
# ----------------------------------------------------------------------------------------------
def gen_first_cycle():
    out = {}
    for tag, (near, far, nc, lindisp) in {
        "lin": (2.0, 6.0, 64, False),
        "disp": (0.2, 2.8, 32, True),
        "ndc": (0.0, 1.0, 16, False),
    }.items():
        cfg = load_cfg("config_blender.yml", nc, nc)
        n = 19
        nr = torch.full((n, 1), near)
        fr = torch.full((n, 1), far)
        for mode, perturb in (("train", True), ("validation", False)):
            cfg.nerf[mode]["perturb"] = perturb
            cfg.nerf[mode]["lindisp"] = lindisp
            torch.manual_seed(7)
            t = ref_samplers.sample_first_cycle(cfg, nr, fr, mode)
            torch.manual_seed(7)
            t_rand = torch.rand((n, nc + 1))
            if not perturb:
                t = t.expand(n, nc + 1) if t.shape[0] != n else t
            out["%s_%s_t" % (tag, mode)] = t.contiguous()
            out["%s_%s_rand" % (tag, mode)] = t_rand
        out["%s_lin" % tag] = torch.linspace(0.0, 1.0, nc + 1)
        out["%s_meta" % tag] = np.array([near, far, nc, int(lindisp)], dtype=np.float64)
    save("first_cycle", **out)


# ----------------------------------------------------------------------------------------------
# 2. encode (a1, a3, a4, a5) + MLP (a7)
# ----------------------------------------------------------------------------------------------
def gen_encode():
    for kind, n, s, shape in (("blender", 6, 64, "cone"), ("llff", 5, 16, "cone"), ("real360", 5, 32, "cone"),
                              ("blender", 3, 8, "cylinder")):
        cfg = load_cfg(CFG_OF[kind], s, s, kind)
        cfg.nerf.ray_shape = shape
        model = ref_models.DDNerfModel(cfg)
        load_weights(model.coarse, True, 11)
        load_weights(model.fine, False, 12)
        ro, rd, rad, _ = synthetic.make_rays(kind, n, seed=3)
        rays = model.get_rays_batches(T(ro), T(rd), T(rad), "train")[0]
        near, far = rays[:, 7:8], rays[:, 8:9]
        torch.manual_seed(5)
        t_vals = ref_samplers.sample_first_cycle(cfg, near, far, "train")
        means, covs = ref_math.cast_rays(t_vals, rays[:, :3], rays[:, 3:6], rays[:, 6:7], shape)
        ipe = ref_math.integrated_pos_enc((means, covs))
        dirs = model.encode_direction_fn(rays[:, -3:])
        with torch.no_grad():
            raw6 = model.run_network(rays, t_vals, model.coarse, "train")
            raw4 = model.run_network(rays, t_vals, model.fine, "train")
        save("encode_%s_%s" % (kind, shape), ro=ro, rd=rd, rad=rad, near=float(cfg.dataset.near),
             far=float(cfg.dataset.far), rays=rays, t_vals=t_vals, means=means, covs=covs, ipe=ipe, dirs=dirs,
             raw6=raw6, raw4=raw4, seeds=np.array([11, 12]))


# ----------------------------------------------------------------------------------------------
# 3. compositing (a10)
# ----------------------------------------------------------------------------------------------
def gen_composite():
    rng = np.random.Generator(np.random.PCG64(21))
    for tag, kind, n, s, white, use_mus, noise_std, scale in (
        ("blender_mus_noise", "blender", 40, 64, False, True, 1.0, 3.0),
        ("blender_plain", "blender", 33, 128, False, False, 0.0, 6.0),
        ("blender_white", "blender", 17, 32, True, True, 0.0, 4.0),
        ("llff_white", "llff", 21, 16, True, False, 0.0, 8.0),
        ("real360_mus", "real360", 25, 64, False, True, 0.5, 5.0),
        ("blender_empty", "blender", 9, 64, False, True, 0.0, 0.0),
    ):
        cfg = load_cfg(CFG_OF[kind], s, s, kind)
        near, far = float(cfg.dataset.near), float(cfg.dataset.far)
        _, rd, _, _ = synthetic.make_rays(kind, n, seed=4)
        t = np.sort(rng.uniform(near, far, (n, s + 1)).astype(np.float32), axis=1)
        t[:, 0] = near
        t[:, -1] = far
        raw = (rng.standard_normal((n, s, 4)) * scale).astype(np.float32)
        if tag == "blender_empty":
            raw[..., 3] = -60.0  # softplus underflows: all-zero weights row (0/0 paths)
        mus = rng.random((n, s)).astype(np.float32) if use_mus else None
        torch.manual_seed(9)
        outs = ref_vr.volume_render_radiance_field(T(raw), T(t), T(rd), radiance_field_noise_std=noise_std,
                                                   white_background=white, mus=None if mus is None else T(mus),
                                                   cfg=cfg)
        torch.manual_seed(9)
        noise = torch.randn((n, s)) * noise_std if noise_std > 0 else None
        rgb_map, disp, acc, weights, depth, cdisp, rgb = outs
        save("composite_" + tag, raw=raw, t_vals=t, rd=rd, mus=mus, noise=noise,
             flags=np.array([int(white), int(kind == "blender")]),
             rgb_map=rgb_map, disp=disp, acc=acc, weights=weights, depth=depth, cdisp=cdisp, rgb=rgb)


# ----------------------------------------------------------------------------------------------
# 4. hierarchical samplers (a11, a12) with captured bin indices
# ----------------------------------------------------------------------------------------------
def weight_profiles(rng, n, nc):
    """Rows that exercise the sampler: smooth, peaky, one-hot, zero, tied, tiny."""
    w = np.zeros((n, nc), np.float32)
    for i in range(n):
        m = i % 8
        if m == 0:
            w[i] = rng.random(nc)
        elif m == 1:
            c = rng.integers(0, nc)
            w[i] = np.exp(-0.5 * ((np.arange(nc) - c) / 0.7) ** 2)
        elif m == 2:
            w[i, rng.integers(0, nc)] = 0.9
        elif m == 3:
            pass  # all-zero row
        elif m == 4:
            w[i] = 0.25  # ties everywhere
        elif m == 5:
            w[i] = rng.random(nc) * 1e-12
        elif m == 6:
            w[i] = rng.random(nc) ** 8
        else:
            w[i, -1] = 1.0
            w[i, 0] = 0.5
    return w.astype(np.float32)


def capture_gather_index(fn, *args, **kw):
    """Run fn while recording the index tensor of the first torch.gather call (= bins_ind)."""
    seen = []
    orig = torch.gather

    def spy(*a, **k):
        if not seen:
            seen.append(k["index"].clone() if "index" in k else a[2].clone())
        return orig(*a, **k)

    torch.gather = spy
    try:
        out = fn(*args, **kw)
    finally:
        torch.gather = orig
    return out, (seen[0] if seen else None)


def gen_samplers():
    rng = np.random.Generator(np.random.PCG64(33))
    for tag, nc, ns, n, near, far in (("c64f129", 64, 129, 96, 2.0, 6.0), ("c16f17", 16, 17, 40, 0.0, 1.0),
                                      ("c33f70", 33, 70, 24, 0.2, 2.8), ("c1f9", 1, 9, 8, 2.0, 6.0)):
        cfg = load_cfg("config_blender.yml", nc, ns - 1)
        cfg.dataset.near, cfg.dataset.far = near, far
        bins = np.sort(rng.uniform(near, far, (n, nc + 1)).astype(np.float32), axis=1)
        bins[:, 0], bins[:, -1] = near, far
        w = weight_profiles(rng, n, nc)
        mus = rng.random((n, nc)).astype(np.float32)
        sig = (rng.random((n, nc)) * 1.2 + 0.001).astype(np.float32)
        sig[::3] *= 0.02  # narrow in-cell gaussians: z clamp 0.999 and t clip are hit
        sqrt2 = np.float32(np.sqrt(2.0))
        left = ref_math.approximate_cdf((0 - T(mus)) / T(sig))
        part = ref_math.approximate_cdf((1 - T(mus)) / T(sig)) - left
        out = dict(bins=bins, weights=w, mus=mus, sigmas=sig, left=left, part=part,
                   meta=np.array([near, far, nc, ns], np.float64))
        for pad in (True, False):
            cfg.train_params.pdf_padding = pad
            for det in (True, False):
                key = "pad%d_det%d" % (pad, det)
                torch.manual_seed(13)
                (s_dd, ind) = capture_gather_index(
                    ref_samplers.sample_pdf_with_mu_sigma, T(bins), T(w), T(mus), T(sig), part.clone(), left.clone(),
                    ns, cfg, det=det)
                torch.manual_seed(13)
                rnd = torch.rand(n, ns)
                out["dd_" + key] = s_dd
                if ind is not None:
                    out["ddind_" + key] = ind.to(torch.int32)
                if nc > 1:  # the reference's sample_pdf raises on a single coarse cell (empty cumsum)
                    torch.manual_seed(13)
                    s_mip = ref_samplers.sample_pdf(T(bins), T(w), ns, cfg, det=det)
                    out["mip_" + key] = s_mip
                if not det:
                    out["rand"] = rnd
        out["u_dd_det"] = torch.linspace(0.0, 0.9999, ns)
        out["u_mip_det"] = torch.linspace(0.0, 1.0, ns)
        out["arange_dd"] = torch.arange(ns) * (1 / (ns - 1))
        out["arange_mip"] = torch.arange(ns) * (1 / ns)
        save("sampler_" + tag, **out)


# ----------------------------------------------------------------------------------------------
# 5. DD head (a8) + dp loss (a13) + whole run_iter, captured from DDNerfModel.predict
# ----------------------------------------------------------------------------------------------
class Spy:
    """Wrap the three free functions DDNerfModel.predict calls so the inline DD-head
    tensors (models/models.py:242-273) become observable without touching the reference."""

    def __init__(self):
        self.rec = {}

    def __enter__(self):
        self.o_s, self.o_d, self.o_v = (ref_models.sample_pdf_with_mu_sigma, ref_models.estimate_dp_loss,
                                         ref_models.volume_render_radiance_field)
        rec = self.rec

        def s(bins, weights, mus, sigmas, part, left, ns, cfg, det=True):
            rec.update(s_bins=bins, s_weights=weights, s_mus=mus, s_ssig=sigmas, s_spart=part, s_sleft=left)
            out = self.o_s(bins, weights, mus, sigmas, part, left, ns, cfg, det=det)
            rec["s_out"] = out
            return out

        def d(t1, t0, w1, w0, mus0, sig0, left0, part0, cfg):
            rec.update(d_t1=t1, d_t0=t0, d_w1=w1, d_w0=w0, d_mus0=mus0, d_sig0=sig0, d_left0=left0, d_part0=part0)
            out = self.o_d(t1, t0, w1, w0, mus0, sig0, left0, part0, cfg)
            rec["d_out"] = out
            return out

        def v(raw, t, rd, **kw):
            i = rec.get("v_n", 0)
            rec["v_n"] = i + 1
            rec["v%d_raw" % i] = raw
            rec["v%d_t" % i] = t
            return self.o_v(raw, t, rd, **kw)

        ref_models.sample_pdf_with_mu_sigma, ref_models.estimate_dp_loss, ref_models.volume_render_radiance_field = s, d, v
        return self

    def __exit__(self, *a):
        ref_models.sample_pdf_with_mu_sigma, ref_models.estimate_dp_loss, ref_models.volume_render_radiance_field = (
            self.o_s, self.o_d, self.o_v)


def flat_out(output):
    res = {}
    for lvl in output:
        for k, v in output[lvl].items():
            if v is None or v is False:
                continue
            res["o%d_%s" % (lvl, k)] = v
    return res


def replay_randoms(seed, shapes):
    """The reference draws, per ray chunk and in this order: rand(first cycle, train only),
    randn(coarse composite), rand(sampler, non-det only), randn(fine composite)."""
    torch.manual_seed(seed)
    out = []
    for kind, shp in shapes:
        out.append(torch.rand(shp) if kind == "rand" else torch.randn(shp))
    return out


def gen_runiter():
    for model_type, kind, n, nc, nf, sharpen, noise, *opt in (
        # N = 64 gradient fixture with the dp-loss term switched off: the coarse net then receives ONLY the MSE
        # gradient through composite_bwd -> MLP backward, which pins that chain tightly (the dp-loss gradient is ill-conditioned
        # in the reference itself)
        ("DDNerfModel", "blender", 64, 64, 64, 8.0, 0.0, "dp0"),
        ("DDNerfModel", "blender", 32, 64, 128, 20.0, 1.0),
        ("DDNerfModel", "blender", 24, 32, 32, 1.0, 0.0),
        ("DDNerfModel", "llff", 24, 16, 16, 20.0, 1.0),
        ("DDNerfModel", "real360", 24, 32, 48, 8.0, 0.0),
        ("GeneralMipNerfModel", "blender", 24, 64, 128, 20.0, 1.0),
        ("GeneralMipNerfModel", "llff", 24, 16, 16, 20.0, 0.0),
    ):
        cfgname = (CFG_OF if model_type == "DDNerfModel" else CFG_MIP_OF)[kind]
        cfg = load_cfg(cfgname, nc, nf, kind)
        dp0 = "dp0" in opt
        if dp0:
            cfg.train_params.dp_coeficient = 0.0
        for mode in ("train", "validation"):
            cfg.nerf[mode]["radiance_field_noise_std"] = noise
        if cfg.train_params.set_automatic_dist_reg_coeficient:  # train_model.py:124-125
            cfg.train_params.dist_reg_coeficient = min(max(1 / cfg.nerf.train.num_coarse, 0.01), 0.12)
        model = getattr(ref_models, cfg.nerf.type)(cfg)
        dd = model_type == "DDNerfModel"
        load_weights(model.coarse, dd, 11, sharpen)
        if dd:
            load_weights(model.fine, False, 12, sharpen)
        ro, rd, rad, tgt = synthetic.make_rays(kind, n, seed=6)
        for mode in ("train",) if dp0 else ("train", "validation"):
            tag = "runiter_%s_%s_%dx%d%s_%s" % ("dd" if dd else "mip", kind, nc, nf, "dp0" if dp0 else "", mode)
            perturb = bool(cfg.nerf[mode]["perturb"])
            shapes = []
            if perturb:
                shapes.append(("rand", (n, nc + 1)))
            if noise > 0:
                shapes.append(("randn", (n, nc)))
            if perturb:
                shapes.append(("rand", (n, nf + 1)))
            if noise > 0:
                shapes.append(("randn", (n, nf)))
            rnd = replay_randoms(17, shapes)
            torch.manual_seed(17)
            extra = {}
            with Spy() as spy:
                if mode == "train":
                    model.train()
                    for p in list(model.coarse.parameters()) + list(model.fine.parameters()):
                        p.grad = None
                    out = model.run_iter(T(ro), T(rd), T(rad), mode="train", rgb_target=T(tgt))
                    loss = 0
                    for j in range(len(out)):
                        loss = loss + cfg.train_params.loss_coeficients[j] * torch.nn.functional.mse_loss(out[j]["rgb"], T(tgt))
                    if dd:
                        loss = loss + cfg.train_params.dp_coeficient * out[1]["dp_loss"].mean()
                    loss.backward()
                    extra["loss"] = loss
                    nets = [("c", model.coarse)] + ([("f", model.fine)] if dd else [])
                    for pfx, net in nets:
                        for name, p in net.named_parameters():
                            g = p.grad.reshape(-1)
                            extra["g%s_%s_sub" % (pfx, name)] = g[::61].clone()
                            extra["g%s_%s_stat" % (pfx, name)] = torch.stack([g.double().norm(), g.double().sum()])
                else:
                    model.eval()
                    with torch.no_grad():
                        out = model.run_iter(T(ro), T(rd), T(rad), mode="validation", rgb_target=T(tgt))
            rec = {k: v for k, v in spy.rec.items() if isinstance(v, torch.Tensor)}
            rnd_named = {"rnd%d" % i: r for i, r in enumerate(rnd)}
            save(tag, ro=ro, rd=rd, rad=rad, tgt=tgt,
                 meta=np.array([nc, nf, sharpen, noise, float(cfg.dataset.near), float(cfg.dataset.far),
                                float(cfg.train_params.dist_reg_coeficient),
                                float(cfg.train_params.gaussian_smooth_factor), int(cfg.train_params.pdf_padding)]),
                 dp_coef=np.array(float(cfg.train_params.dp_coeficient)),
                 **flat_out(out), **rec, **rnd_named, **extra)


def gen_fullsize():
    """BASELINE configs at their stated sizes (validation pass, noise off): only every STRIDE-th ray's outputs are stored,
    the inputs come from the synthetic seeds.  cfg1: 256 rays x 64 x 64; cfg2: 4096 x (64 + 128) blender; cfg3: config_ff
    NDC rays; cfg4: config_360, 8192 rays; cfg5: config_blender_mipnerf (one shared MLP)."""
    for tag, model_type, kind, n, nc, nf in (
        ("cfg1", "DDNerfModel", "blender", 256, 64, 64),
        ("cfg2", "DDNerfModel", "blender", 4096, 64, 128),
        ("cfg3", "DDNerfModel", "llff", 4096, 64, 128),
        ("cfg4", "DDNerfModel", "real360", 8192, 64, 128),
        ("cfg5", "GeneralMipNerfModel", "blender", 4096, 64, 128),
    ):
        dd = model_type == "DDNerfModel"
        cfg = load_cfg((CFG_OF if dd else CFG_MIP_OF)[kind], nc, nf, kind)
        for mode in ("train", "validation"):
            cfg.nerf[mode]["radiance_field_noise_std"] = 0.0
        if cfg.train_params.set_automatic_dist_reg_coeficient:
            cfg.train_params.dist_reg_coeficient = min(max(1 / cfg.nerf.train.num_coarse, 0.01), 0.12)
        model = getattr(ref_models, cfg.nerf.type)(cfg)
        load_weights(model.coarse, dd, 11, 20.0)
        if dd:
            load_weights(model.fine, False, 12, 20.0)
        ro, rd, rad, tgt = synthetic.make_rays(kind, n, seed=1)
        model.eval()
        with torch.no_grad():
            out = model.run_iter(T(ro), T(rd), T(rad), mode="validation", rgb_target=T(tgt))
        stride = max(1, n // 64 - 3)  # 61 at 4096 rays, 125 at 8192, 1 at 256
        keep = {}
        for lvl in out:
            for k in ("rgb", "depth", "acc", "disp", "weights"):
                keep["o%d_%s" % (lvl, k)] = out[lvl][k][::stride]
            for k in ("dp_loss", "mus_reg", "sig_reg", "mus_loss", "sig_loss"):
                if out[lvl].get(k, None) is not None:
                    keep["o%d_%s" % (lvl, k)] = out[lvl][k]
        save("fullsize_%s_%s_%s_%d_%dx%d" % (tag, "dd" if dd else "mip", kind, n, nc, nf),
             meta=np.array([n, nc, nf, 20.0, stride, float(cfg.dataset.near), float(cfg.dataset.far),
                            float(cfg.train_params.dist_reg_coeficient), float(cfg.train_params.gaussian_smooth_factor),
                            int(cfg.train_params.pdf_padding)]), **keep)


def gen_depthanalysis():
    """run_iter(depth_analysis_validation=True) (models/models.py:108-112, 307-319) and get_combined_samples
    (models/samplers.py:6-27, dataset.combined_sampling_method) on a handful of rays."""
    for model_type, kind, n, nc, nf in (("DDNerfModel", "blender", 6, 16, 16), ("GeneralMipNerfModel", "blender", 6, 16, 16)):
        dd = model_type == "DDNerfModel"
        cfg = load_cfg((CFG_OF if dd else CFG_MIP_OF)[kind], nc, nf, kind)
        for mode in ("train", "validation"):
            cfg.nerf[mode]["radiance_field_noise_std"] = 0.0
        if cfg.train_params.set_automatic_dist_reg_coeficient:
            cfg.train_params.dist_reg_coeficient = min(max(1 / cfg.nerf.train.num_coarse, 0.01), 0.12)
        model = getattr(ref_models, cfg.nerf.type)(cfg)
        load_weights(model.coarse, dd, 11, 8.0)
        if dd:
            load_weights(model.fine, False, 12, 8.0)
        ro, rd, rad, tgt = synthetic.make_rays(kind, n, seed=4)
        model.eval()
        with torch.no_grad():
            out = model.run_iter(T(ro), T(rd), T(rad), mode="validation", depth_analysis_validation=True, rgb_target=T(tgt))
        keep = {}
        for lvl in out:
            for k, v in out[lvl].items():
                if isinstance(v, torch.Tensor) and ("to_plot" in k or "for_plot" in k or k in ("rgb", "depth")):
                    keep["o%d_%s" % (lvl, k)] = v
        save("depthanalysis_%s_%s" % ("dd" if dd else "mip", kind), ro=ro, rd=rd, rad=rad,
             meta=np.array([nc, nf, 8.0, 0.0, float(cfg.dataset.near), float(cfg.dataset.far), float(cfg.train_params.dist_reg_coeficient),
                            float(cfg.train_params.gaussian_smooth_factor), int(cfg.train_params.pdf_padding)]), **keep)
    # combined sampling (config_360 with the switch on), perturb off and on
    cfg = load_cfg(CFG_OF["real360"], 16, 16, "real360")
    cfg.dataset.combined_sampling_method = True
    near = torch.full((5, 1), float(cfg.dataset.near))
    far = torch.full((5, 1), float(cfg.dataset.far))
    rec = {}
    for mode, perturb in (("validation", False), ("train", True)):
        cfg.nerf[mode]["perturb"] = perturb
        torch.manual_seed(9)
        rnd = torch.rand(5, 17) if perturb else None
        torch.manual_seed(9)
        rec["t_%s" % mode] = ref_samplers.sample_first_cycle(cfg, near, far, mode)
        if rnd is not None:
            rec["rnd_%s" % mode] = rnd
    save("combined_first_cycle", meta=np.array([16, float(cfg.dataset.near), float(cfg.dataset.far), float(cfg.dataset.combined_split)]), **rec)


def gen_manifest():
    """Checkpoint interchange: the reference models' state_dict manifests (name, shape, dtype in order), the
    Adam state_dict layout after one step, and the key set of the checkpoint dict the reference's training loop saves
    (train_model.py:248-263; that script cannot be imported here -- tensorboard / imageio are absent -- so the keys are read
    from its source text as data)."""
    import json
    import re

    out = {}
    for model_type, kind in (("DDNerfModel", "blender"), ("GeneralMipNerfModel", "blender")):
        dd = model_type == "DDNerfModel"
        cfg = load_cfg((CFG_OF if dd else CFG_MIP_OF)[kind], 8, 8, kind)
        model = getattr(ref_models, cfg.nerf.type)(cfg)
        for tag, net in (("coarse", model.coarse), ("fine", model.fine)):
            out["%s.%s" % (model_type, tag)] = [[k, list(v.shape), str(v.dtype)] for k, v in net.state_dict().items()]
        opt = torch.optim.Adam(model.coarse.parameters(), lr=1e-3)
        sum(p.sum() for p in model.coarse.parameters()).backward()
        opt.step()
        osd = opt.state_dict()
        out["%s.optimizer" % model_type] = {"top": sorted(osd.keys()), "param_group": sorted(osd["param_groups"][0].keys()),
                                           "n_params": len(osd["param_groups"][0]["params"]),
                                           "state_entry": sorted(osd["state"][0].keys())}
    src = open(os.path.join(REF, "train_model.py")).read()
    blk = src[src.index("checkpoint_dict = {"):src.index("torch.save(", src.index("checkpoint_dict = {"))]
    out["checkpoint_keys_always"] = re.findall(r'^\s*"(\w+)":', blk, flags=re.M)
    out["checkpoint_keys_two_networks"] = re.findall(r'checkpoint_dict\["(\w+)"\]', blk)
    out["checkpoint_file"] = re.search(r'os\.path\.join\(logdir, "([\w.]+)"\)', src[src.index("checkpoint_dict = {"):]).group(1)
    path = os.path.join(HERE, "checkpoint_manifest.json")
    json.dump(out, open(path, "w"), indent=1)
    print("%-40s %8.1f KB" % ("checkpoint_manifest.json", os.path.getsize(path) / 1024))


def gen_trainsteps():
    """A few whole optimiser steps of the reference's loop (train_model.py:144-177: run_iter, loss assembly, backward,
    one Adam per network) on fixed rays with replayed randoms: pins that every step evaluates the UPDATED weights."""
    n, nc, nf, steps, lr = 32, 32, 32, 5, 1e-3
    for model_type, kind in (("DDNerfModel", "blender"), ("GeneralMipNerfModel", "blender")):
        dd = model_type == "DDNerfModel"
        cfg = load_cfg((CFG_OF if dd else CFG_MIP_OF)[kind], nc, nf, kind)
        cfg.nerf.train["radiance_field_noise_std"] = 1.0
        if cfg.train_params.set_automatic_dist_reg_coeficient:
            cfg.train_params.dist_reg_coeficient = min(max(1 / cfg.nerf.train.num_coarse, 0.01), 0.12)
        model = getattr(ref_models, cfg.nerf.type)(cfg)
        load_weights(model.coarse, dd, 11, 4.0)
        if dd:
            load_weights(model.fine, False, 12, 4.0)
        optims = [torch.optim.Adam(model.coarse.parameters(), lr=lr)]
        if dd:
            optims.append(torch.optim.Adam(model.fine.parameters(), lr=lr))
        ro, rd, rad, tgt = synthetic.make_rays(kind, n, seed=8)
        shapes = [("rand", (n, nc + 1)), ("randn", (n, nc)), ("rand", (n, nf + 1)), ("randn", (n, nf))]
        model.train()
        rec = {}
        for it in range(steps):
            rnd = replay_randoms(100 + it, shapes)
            torch.manual_seed(100 + it)
            out = model.run_iter(T(ro), T(rd), T(rad), mode="train", rgb_target=T(tgt))
            mses = [torch.nn.functional.mse_loss(out[j]["rgb"], T(tgt)) for j in range(len(out))]
            loss = sum(cfg.train_params.loss_coeficients[j] * mses[j] for j in range(len(out)))
            if dd:
                loss = loss + cfg.train_params.dp_coeficient * out[1]["dp_loss"].mean()
            loss.backward()
            for o in optims:
                o.step()
                o.zero_grad()
            rec["loss%d" % it] = loss.detach()
            rec["mse%d" % it] = torch.stack([m.detach() for m in mses])
            for i, r in enumerate(rnd):
                rec["rnd%d_%d" % (it, i)] = r
        nets = [("c", model.coarse)] + ([("f", model.fine)] if dd else [])
        for pfx, net in nets:
            for name, p in net.named_parameters():
                rec["p%s_%s_sub" % (pfx, name)] = p.detach().reshape(-1)[::61].clone()
        save("trainsteps_%s_%s" % ("dd" if dd else "mip", kind), ro=ro, rd=rd, rad=rad, tgt=tgt,
             meta=np.array([nc, nf, 4.0, 1.0, float(cfg.dataset.near), float(cfg.dataset.far),
                            float(cfg.train_params.dist_reg_coeficient), float(cfg.train_params.gaussian_smooth_factor),
                            int(cfg.train_params.pdf_padding), steps, lr]), **rec)


def gen_dploss():
    """estimate_dp_loss in isolation, incl. the row-filter misalignment and the all-filtered return."""
    rng = np.random.Generator(np.random.PCG64(55))
    for tag, kind, n, nc, nf, drop in (("blender_drop", "blender", 24, 32, 48, True),
                                       ("blender_full", "blender", 16, 64, 128, False),
                                       ("llff", "llff", 16, 16, 16, True),
                                       ("blender_allzero", "blender", 6, 16, 16, "all")):
        cfg = load_cfg(CFG_OF[kind], nc, nf, kind)
        near, far = float(cfg.dataset.near), float(cfg.dataset.far)
        t0 = np.sort(rng.uniform(near, far, (n, nc + 1)).astype(np.float32), 1)
        t0[:, 0], t0[:, -1] = near, far
        t1 = np.sort(rng.uniform(near, far, (n, nf + 1)).astype(np.float32), 1)
        t1[:, 0], t1[:, -1] = near, far
        w0 = weight_profiles(rng, n, nc) + np.float32(1e-3) * rng.random((n, nc)).astype(np.float32)
        w1 = (rng.random((n, nf)) ** 4).astype(np.float32)
        if drop is True:
            w1[::5] = 0.0
        elif drop == "all":
            w1[:] = 0.0
        mus = rng.random((n, nc)).astype(np.float32)
        sig = (rng.random((n, nc)) + 0.001).astype(np.float32)
        left = ref_math.approximate_cdf((0 - T(mus)) / T(sig))
        part = ref_math.approximate_cdf((1 - T(mus)) / T(sig)) - left
        w0t, must, sigt = T(w0).requires_grad_(), T(mus).requires_grad_(), T(sig).requires_grad_()
        loss = ref_dd.estimate_dp_loss(T(t1), T(t0), T(w1), w0t, must, sigt, left, part, cfg)
        g = {}
        if loss.requires_grad:
            loss.backward()
            g = dict(g_w0=w0t.grad, g_mus=must.grad, g_sig=sigt.grad)
        save("dploss_" + tag, t1=t1, t0=t0, w1=w1, w0=w0, mus=mus, sig=sig, left=left, part=part,
             loss=loss.detach().to(torch.float64), is_blender=np.array(int(kind == "blender")), **g)


def gen_aten_orders():
    """Pin the two ATen CPU reduction orders the samplers depend on."""
    rng = np.random.Generator(np.random.PCG64(77))
    out = {}
    for n in (1, 5, 16, 17, 31, 32, 33, 64, 65, 128, 129, 130):
        x = (rng.random((32, n)) ** 3).astype(np.float32)
        out["x%d" % n] = x
        out["sum%d" % n] = torch.sum(T(x), dim=-1)
        out["cumsum%d" % n] = torch.cumsum(T(x), dim=-1)
        out["cumprod%d" % n] = torch.cumprod(T(1 - 0.5 * x), dim=-1)
    save("aten_orders", **out)


def gen_raygen():
    """get_ray_bundle / ndc_mipnerf_rays (the callers immediately upstream of the path)"""
    from data_utils.dataset_helpers import ndc_mipnerf_rays

    rng = np.random.Generator(np.random.PCG64(91))
    out = {}
    for tag, H, W, focal in (("a", 9, 7, 11.5), ("b", 6, 12, 20.0)):
        q, _ = np.linalg.qr(rng.standard_normal((3, 3)))
        pose = np.concatenate([q, rng.standard_normal((3, 1)) * 3], 1).astype(np.float32)
        if tag == "b":
            pose[1, 3] = 0.0  # exercises the zero nudge of the origins
            pose[:3, :3] = np.eye(3, dtype=np.float32)  # ... and of the directions (centre pixel)
        o, d, r = ref_helpers.get_ray_bundle(H, W, focal, T(pose))
        o, d = o.clone(), d.clone()
        on, dn, rn = ndc_mipnerf_rays(H, W, focal, o, d, 1)
        out.update({tag + "_pose": pose, tag + "_hwf": np.array([H, W, focal]), tag + "_o": o, tag + "_d": d, tag + "_r": r,
                    tag + "_on": on, tag + "_dn": dn, tag + "_rn": rn})
    save("raygen", **out)


def gen_sampler4096():
    """The sampler boundary at BASELINE size: the reference's own cfg2 coarse pass (4096 blender rays x 64 bins, validation, noise
    off) feeds sample_pdf_with_mu_sigma; inputs exactly as the reference hands them over, the bin indices of all 4096 x 129
    samples (uint8) and the samples."""
    n, nc, nf = 4096, 64, 128
    cfg = load_cfg(CFG_OF["blender"], nc, nf, "blender")
    for mode in ("train", "validation"):
        cfg.nerf[mode]["radiance_field_noise_std"] = 0.0
    if cfg.train_params.set_automatic_dist_reg_coeficient:
        cfg.train_params.dist_reg_coeficient = min(max(1 / cfg.nerf.train.num_coarse, 0.01), 0.12)
    model = ref_models.DDNerfModel(cfg)
    load_weights(model.coarse, True, 11, 20.0)
    load_weights(model.fine, False, 12, 20.0)
    ro, rd, rad, tgt = synthetic.make_rays("blender", n, seed=1)
    model.eval()
    with Spy() as spy:
        with torch.no_grad():
            (_, ind) = capture_gather_index(model.run_iter, T(ro), T(rd), T(rad), mode="validation", rgb_target=T(tgt))
    r = spy.rec
    assert ind is not None and tuple(ind.shape) == (n, nf + 1) and int(ind.max()) < 256
    assert bool((r["s_bins"] == r["s_bins"][0:1]).all())  # validation: one first-cycle row for every ray
    save("sampler4096_cfg2", bins_row=r["s_bins"][0], weights=r["s_weights"], mus=r["s_mus"], ssig=r["s_ssig"], spart=r["s_spart"],
         sleft=r["s_sleft"], bins_ind=ind.to(torch.uint8), samples=r["s_out"],
         meta=np.array([n, nc, nf + 1, float(cfg.dataset.near), float(cfg.dataset.far), int(cfg.train_params.pdf_padding)], np.float64))


def gen_grad4096():
    """Parameter gradients at BASELINE size (cfg2: 4096 rays x (64 + 128), train mode with perturb / noise off so that no random
    tensor is involved), every 61st entry of every parameter's gradient plus its norm and sum; with the dp term on and off."""
    n, nc, nf = 4096, 64, 128
    for tag, dp_on in (("dp1", True), ("dp0", False)):
        cfg = load_cfg(CFG_OF["blender"], nc, nf, "blender")
        for mode in ("train", "validation"):
            cfg.nerf[mode]["radiance_field_noise_std"] = 0.0
            cfg.nerf[mode]["perturb"] = False
        if not dp_on:
            cfg.train_params.dp_coeficient = 0.0
        if cfg.train_params.set_automatic_dist_reg_coeficient:
            cfg.train_params.dist_reg_coeficient = min(max(1 / cfg.nerf.train.num_coarse, 0.01), 0.12)
        model = ref_models.DDNerfModel(cfg)
        load_weights(model.coarse, True, 11, 8.0)
        load_weights(model.fine, False, 12, 8.0)
        ro, rd, rad, tgt = synthetic.make_rays("blender", n, seed=6)
        model.train()
        out = model.run_iter(T(ro), T(rd), T(rad), mode="train", rgb_target=T(tgt))
        mses = [torch.nn.functional.mse_loss(out[j]["rgb"], T(tgt)) for j in range(2)]
        loss = sum(cfg.train_params.loss_coeficients[j] * mses[j] for j in range(2))
        loss = loss + cfg.train_params.dp_coeficient * out[1]["dp_loss"].mean()
        loss.backward()
        rec = {"loss": loss.detach(), "mse": torch.stack([m.detach() for m in mses]), "dp_loss": out[1]["dp_loss"].detach()}
        for pfx, net in (("c", model.coarse), ("f", model.fine)):
            for name, p in net.named_parameters():
                g = p.grad.reshape(-1)
                rec["g%s_%s_sub" % (pfx, name)] = g[::61].clone()
                rec["g%s_%s_stat" % (pfx, name)] = torch.stack([g.double().norm(), g.double().sum()])
        save("grad4096_cfg2_" + tag, meta=np.array([n, nc, nf, 8.0, 0.0, float(cfg.dataset.near), float(cfg.dataset.far),
                                                     float(cfg.train_params.dist_reg_coeficient), float(cfg.train_params.gaussian_smooth_factor),
                                                     int(cfg.train_params.pdf_padding), float(cfg.train_params.dp_coeficient)]), **rec)
        del out, loss, model


def gen_grad4096_trained():
    """gen_grad4096 at the TRAINED state: the reference's parameter gradients of one 4096-ray training pass (perturb / noise off) with
    the networks of its own 3000-iteration run (trained_weights_dd_blender.npz) and the procedural targets they were trained on, dp
    term on (config value) and off -- the backward kernels on trained activations (sparse ReLU patterns, peaked weights)."""
    n, nc, nf = 4096, 64, 128
    w = np.load(os.path.join(HERE, "trained_weights_dd_blender.npz"))
    for tag, dp_on in (("dp1", True), ("dp0", False)):
        cfg = load_cfg(CFG_OF["blender"], nc, nf, "blender")
        for mode in ("train", "validation"):
            cfg.nerf[mode]["radiance_field_noise_std"] = 0.0
            cfg.nerf[mode]["perturb"] = False
        if not dp_on:
            cfg.train_params.dp_coeficient = 0.0
        if cfg.train_params.set_automatic_dist_reg_coeficient:
            cfg.train_params.dist_reg_coeficient = min(max(1 / cfg.nerf.train.num_coarse, 0.01), 0.12)
        cfg.train_params.gaussian_smooth_factor = float(w["meta"][1])       # (the smoothing schedule's value at iteration 2999)
        model = ref_models.DDNerfModel(cfg)
        model.coarse.load_state_dict({k[2:]: T(w[k]) for k in w.files if k.startswith("c.")})
        model.fine.load_state_dict({k[2:]: T(w[k]) for k in w.files if k.startswith("f.")})
        ro, rd, rad, _ = synthetic.make_rays("blender", n, seed=6)
        tgt = synthetic.procedural_targets(ro, rd)
        model.train()
        out = model.run_iter(T(ro), T(rd), T(rad), mode="train", rgb_target=T(tgt))
        mses = [torch.nn.functional.mse_loss(out[j]["rgb"], T(tgt)) for j in range(2)]
        loss = sum(cfg.train_params.loss_coeficients[j] * mses[j] for j in range(2))
        loss = loss + cfg.train_params.dp_coeficient * out[1]["dp_loss"].mean()
        loss.backward()
        rec = {"loss": loss.detach(), "mse": torch.stack([m.detach() for m in mses]), "dp_loss": out[1]["dp_loss"].detach()}
        for pfx, net in (("c", model.coarse), ("f", model.fine)):
            for name, p in net.named_parameters():
                g = p.grad.reshape(-1)
                rec["g%s_%s_sub" % (pfx, name)] = g[::61].clone()
                rec["g%s_%s_stat" % (pfx, name)] = torch.stack([g.double().norm(), g.double().sum()])
        print("grad4096_trained", tag, float(loss), [float(m) for m in mses], flush=True)
        save("grad4096_trained_" + tag, meta=np.array([n, nc, nf, 1.0, 0.0, float(cfg.dataset.near), float(cfg.dataset.far),
                                                        float(cfg.train_params.dist_reg_coeficient), float(cfg.train_params.gaussian_smooth_factor),
                                                        int(cfg.train_params.pdf_padding), float(cfg.train_params.dp_coeficient)]), **rec)
        del out, loss, model


def gen_train1500():
    """1500 iterations of the reference's training loop with the schedule train_model.py hard-wires (:101-107: log-lerp 5e-4 -> 5e-6 over
    cfg.experiment.train_iters with the 2500-step x0.01 warm-up), i.e. the FIRST 1500 steps of a real run, DDNerfModel, 256 fresh rays per
    iteration; loss / MSE / dp every 25 iterations."""
    _gen_train(1500, 2500, None, 25, ("DDNerfModel",), "train1500")


def gen_train300():
    _gen_train(300, 50, 300, 10, ("DDNerfModel", "GeneralMipNerfModel"), "train300")


def _gen_train(iters, delay, max_steps, every, model_types, tag, nudge=None, keep=False):
    """Training parity (SURVEY.md 8d "PSNR vs ref (2)"): 300 iterations of the reference's loop (train_model.py:132-177: smoothing
    schedule, lr schedule, run_iter, loss assembly, backward, one Adam per network) on a procedural scene -- a fresh seeded batch of
    256 rays per iteration with analytic colour targets -- with perturb / noise off (no random tensors).  The learning-rate
    function is the reference's own (general_utils/nerf_helpers.py:211-245) with a 50-step warm-up instead of the 2500 the script
    hard-wires (a 300-step run would otherwise never leave the warm-up).  Loss / MSE every 10 iterations."""
    import functools

    n, nc, nf = 256, 64, 128
    for model_type, kind in [(m, "blender") for m in model_types]:
        dd = model_type == "DDNerfModel"
        cfg = load_cfg((CFG_OF if dd else CFG_MIP_OF)[kind], nc, nf, kind)
        for mode in ("train", "validation"):
            cfg.nerf[mode]["radiance_field_noise_std"] = 0.0
            cfg.nerf[mode]["perturb"] = False
        model = getattr(ref_models, cfg.nerf.type)(cfg)
        load_weights(model.coarse, dd, 11, 1.0)
        if dd:
            load_weights(model.fine, False, 12, 1.0)
        if nudge == "ulp":       # reference-vs-itself drift (gen_drift1500): every initial weight moved ONE fp32 ulp away from zero
            with torch.no_grad():
                for net in ([model.coarse, model.fine] if dd else [model.coarse]):
                    for p_ in net.parameters():
                        p_.copy_(torch.nextafter(p_, p_ * 2))
        optims = [torch.optim.Adam(model.coarse.parameters(), lr=cfg.optimizer.lr)]
        if dd:
            optims.append(torch.optim.Adam(model.fine.parameters(), lr=cfg.optimizer.lr))
        steps = max_steps if max_steps is not None else int(cfg.experiment.train_iters)       # train_model.py:102 (the config's run length)
        lr_function = functools.partial(ref_helpers.learning_rate_decay, lr_init=0.0005, lr_final=5e-6, max_steps=steps,
                                        lr_delay_steps=delay, lr_delay_mult=0.01)
        dsmooth = (cfg.train_params.gaussian_smooth_factor - cfg.train_params.final_smooth) / cfg.train_params.finnish_smooth
        initial_smooth = cfg.train_params.gaussian_smooth_factor
        if cfg.train_params.set_automatic_dist_reg_coeficient:
            cfg.train_params.dist_reg_coeficient = min(max(1 / cfg.nerf.train.num_coarse, 0.01), 0.12)
        rec = {"loss": [], "mse": [], "dp": [], "lr": [], "it": []}
        for i in range(iters):
            model.cfg.train_params.gaussian_smooth_factor = initial_smooth - dsmooth * i
            model.train()
            lr_new = lr_function(i)
            for o in optims:
                for g in o.param_groups:
                    g["lr"] = lr_new
            ro, rd, rad, _ = synthetic.make_rays(kind, n, seed=5000 + i)
            tgt = synthetic.procedural_targets(ro, rd)
            out = model.run_iter(T(ro), T(rd), T(rad), mode="train", rgb_target=T(tgt))
            mses = [torch.nn.functional.mse_loss(out[j]["rgb"], T(tgt)) for j in range(len(out))]
            loss = sum(cfg.train_params.loss_coeficients[j] * mses[j] for j in range(len(out)))
            dp = torch.zeros(())
            if dd:
                dp = out[1]["dp_loss"].mean()
                loss = loss + cfg.train_params.dp_coeficient * dp
            loss.backward()
            for o in optims:
                o.step()
                o.zero_grad()
            if i % every == 0 or i == iters - 1:
                rec["it"].append(i)
                rec["loss"].append(float(loss))
                rec["mse"].append([float(m) for m in mses])
                rec["dp"].append(float(dp))
                rec["lr"].append(float(lr_new))
                print(model_type, i, float(loss), [float(m) for m in mses], flush=True)
        if keep:                 # gen_trained / gen_drift1500 use the curve (and the trained model) themselves
            return model, cfg, {k: np.array(v) for k, v in rec.items()}
        save("%s_%s_%s" % (tag, "dd" if dd else "mip", kind),
             meta=np.array([n, nc, nf, iters, delay, float(cfg.dataset.near), float(cfg.dataset.far), steps]),
             it=np.array(rec["it"]), loss=np.array(rec["loss"]), mse=np.array(rec["mse"]), dp=np.array(rec["dp"]), lr=np.array(rec["lr"]))


TRAINED_ITERS = 3000


def gen_trained():
    """Parity on TRAINED weights (train_model.py:132-177 produces them; every other fixture uses seeded-uniform ones): the reference's
    loop of gen_train1500 carried on to TRAINED_ITERS iterations; its final coarse / fine state_dicts (trained_weights_dd_blender.npz),
    the curve (its first 1500 iterations must reproduce train1500_dd_blender.npz -- asserted here), and the reference's
    run_iter(validation) with those weights on 4096 blender rays and on 4096 NDC rays (config_ff.yml), every 61st ray's outputs kept,
    plus -- blender -- the sampler boundary of that pass: the inputs the reference hands sample_pdf_with_mu_sigma and its bin indices
    for all 4096 x 129 samples."""
    model, cfg_t, rec = _gen_train(TRAINED_ITERS, 2500, None, 25, ("DDNerfModel",), "train%d" % TRAINED_ITERS, keep=True)
    old = np.load(os.path.join(HERE, "train1500_dd_blender.npz"))
    k = len(old["it"]) - 1         # (the last record of the 1500-step run is iteration 1499, off the 25-grid)
    assert np.array_equal(rec["it"][:k], old["it"][:k]) and np.array_equal(rec["loss"][:k], old["loss"][:k]), "train1500 not reproduced"
    sd = {"c." + k_: v for k_, v in model.coarse.state_dict().items()}
    sd.update({"f." + k_: v for k_, v in model.fine.state_dict().items()})
    save("trained_weights_dd_blender", meta=np.array([TRAINED_ITERS, float(model.cfg.train_params.gaussian_smooth_factor)]),
         it=rec["it"], loss=rec["loss"], mse=rec["mse"], dp=rec["dp"], **sd)
    n, nc, nf = 4096, 64, 128
    for kind in ("blender", "llff"):
        cfg = load_cfg(CFG_OF[kind], nc, nf, kind)
        for mode in ("train", "validation"):
            cfg.nerf[mode]["radiance_field_noise_std"] = 0.0
        if cfg.train_params.set_automatic_dist_reg_coeficient:
            cfg.train_params.dist_reg_coeficient = min(max(1 / cfg.nerf.train.num_coarse, 0.01), 0.12)
        m2 = ref_models.DDNerfModel(cfg)
        m2.coarse.load_state_dict(model.coarse.state_dict())
        m2.fine.load_state_dict(model.fine.state_dict())
        ro, rd, rad, tgt = synthetic.make_rays(kind, n, seed=1)
        if kind == "blender":
            tgt = synthetic.procedural_targets(ro, rd)
        m2.eval()
        with Spy() as spy:
            with torch.no_grad():
                (out, ind) = capture_gather_index(m2.run_iter, T(ro), T(rd), T(rad), mode="validation", rgb_target=T(tgt))
        stride = 61
        keep_ = {}
        for lvl in out:
            for k_ in ("rgb", "depth", "acc", "disp", "weights"):
                keep_["o%d_%s" % (lvl, k_)] = out[lvl][k_][::stride]
            for k_ in ("dp_loss", "mus_reg", "sig_reg", "mus_loss", "sig_loss"):
                if out[lvl].get(k_, None) is not None:
                    keep_["o%d_%s" % (lvl, k_)] = out[lvl][k_]
        psnr = [float(-10 * torch.log10(torch.nn.functional.mse_loss(out[j]["rgb"], T(tgt)))) for j in range(2)]
        print("trained", kind, "psnr vs target", psnr, flush=True)
        save("fullsize_trained_dd_%s_%d_%dx%d" % (kind, n, nc, nf),
             meta=np.array([n, nc, nf, 1.0, stride, float(cfg.dataset.near), float(cfg.dataset.far),
                            float(cfg.train_params.dist_reg_coeficient), float(cfg.train_params.gaussian_smooth_factor),
                            int(cfg.train_params.pdf_padding)]), psnr=np.array(psnr), **keep_)
        if kind == "blender":
            r = spy.rec
            assert ind is not None and tuple(ind.shape) == (n, nf + 1) and int(ind.max()) < 256
            save("sampler4096_trained", bins_row=r["s_bins"][0], weights=r["s_weights"], mus=r["s_mus"], ssig=r["s_ssig"],
                 spart=r["s_spart"], sleft=r["s_sleft"], bins_ind=ind.to(torch.uint8), samples=r["s_out"],
                 meta=np.array([n, nc, nf + 1, float(cfg.dataset.near), float(cfg.dataset.far), int(cfg.train_params.pdf_padding)], np.float64))


def gen_drift1500():
    """How far does the REFERENCE drift from ITSELF over the 1500 iterations of gen_train1500 under a perturbation of fp32 round-off
    size?  (Unperturbed it reproduces train1500_dd_blender.npz bit for bit.)  Two nudges: every initial weight moved one ulp away from
    zero ("ulp"), and the same arithmetic on 4 ATen threads instead of 8 ("thr4": other reduction splits).  The curves calibrate the
    bars of tests/test_hip_baseline_size.py::test_training_curve_1500_iterations_of_the_real_schedule."""
    old = np.load(os.path.join(HERE, "train1500_dd_blender.npz"))
    out = {}
    for tag in ("ulp", "thr4"):
        torch.set_num_threads(4 if tag == "thr4" else 8)
        _, _, rec = _gen_train(1500, 2500, None, 25, ("DDNerfModel",), "drift", nudge="ulp" if tag == "ulp" else None, keep=True)
        torch.set_num_threads(8)
        assert np.array_equal(rec["it"], old["it"])
        out["loss_" + tag], out["mse_" + tag], out["dp_" + tag] = rec["loss"], rec["mse"], rec["dp"]
        ps = lambda m: -10 * np.log10(m)  # noqa: E731
        d = np.abs(ps(rec["mse"]) - ps(old["mse"]))
        print("drift", tag, "max |dPSNR|", d.max(axis=0), "at", d.argmax(axis=0), "mean", d.mean(axis=0),
              "max rel loss", np.max(np.abs(rec["loss"] - old["loss"]) / np.abs(old["loss"])), flush=True)
    save("train1500_drift_dd_blender", it=old["it"], **out)


def gen_ndcswitch():
    """switch_t_ndc_to_regular (data_utils/dataset_helpers.py:45-49; called at train_model.py:227-228)"""
    from data_utils.dataset_helpers import switch_t_ndc_to_regular

    rng = np.random.Generator(np.random.PCG64(93))
    H, W, focal = 7, 9, 13.0
    q, _ = np.linalg.qr(rng.standard_normal((3, 3)))
    pose = np.concatenate([q, rng.standard_normal((3, 1))], 1).astype(np.float32)
    o, d, _ = ref_helpers.get_ray_bundle(H, W, focal, T(pose))
    depth = T(rng.uniform(0.0, 0.98, (H, W)).astype(np.float32))
    save("ndcswitch", ro=o, rd=d, ndc_depth=depth, regular=switch_t_ndc_to_regular(depth, o, d))


def _placeholder_modules(*names):
    """EMPTY modules under the names the reference's loader files import at their top (imageio, cv2, skimage.transform: absent
    from this image) so that the import statements succeed.  They define nothing: only pure numpy functions of those files
    are called below; whatever would touch an image library is replaced by synthetic arrays (see gen_loaders)."""
    import types

    for nm in names:
        if nm not in sys.modules:
            sys.modules[nm] = types.ModuleType(nm)
            parent, _, child = nm.rpartition(".")
            if parent:
                setattr(sys.modules[parent], child, sys.modules[nm])


def gen_colmap():
    """COLMAP binary model readers (data_utils/poses/colmap_read_model.py:108-260, numpy + struct only) on a tiny synthetic model
    that THIS build's writer (ddnerf_amd/colmap.py) produces; committed: the three .bin files and what the reference parsed.
    Plus the poses / depth bounds the reference derives from the model (data_utils/poses/pose_utils.py:10-90, pure numpy behind
    its imageio / skimage imports)."""
    import importlib.util as iu
    import json

    spec = iu.spec_from_file_location("ddn_colmap", os.path.join(REPO, "ddnerf_amd", "colmap.py"))
    colmap = iu.module_from_spec(spec)
    spec.loader.exec_module(colmap)
    from data_utils.poses import colmap_read_model as ref_cm

    rng = np.random.Generator(np.random.PCG64(97))
    root = os.path.join(HERE, "colmap_model")
    sparse = os.path.join(root, "sparse", "0")
    os.makedirs(sparse, exist_ok=True)
    cams, imgs, pts = colmap.synthetic_model(rng, n_images=5, n_points=40)
    colmap.write_cameras_binary(cams, os.path.join(sparse, "cameras.bin"))
    colmap.write_images_binary(imgs, os.path.join(sparse, "images.bin"))
    colmap.write_points3d_binary(pts, os.path.join(sparse, "points3D.bin"))
    rc = ref_cm.read_cameras_binary(os.path.join(sparse, "cameras.bin"))
    ri = ref_cm.read_images_binary(os.path.join(sparse, "images.bin"))
    rp = ref_cm.read_points3d_binary(os.path.join(sparse, "points3D.bin"))
    parsed = {
        "cameras": {str(k): {"model": c.model, "width": int(c.width), "height": int(c.height), "params": [float(x) for x in c.params]}
                    for k, c in rc.items()},
        "images": {str(k): {"qvec": [float(x) for x in im.qvec], "tvec": [float(x) for x in im.tvec], "camera_id": int(im.camera_id),
                            "name": im.name, "xys": np.asarray(im.xys).tolist(), "point3D_ids": [int(x) for x in im.point3D_ids],
                            "rotmat": ref_cm.qvec2rotmat(im.qvec).tolist()} for k, im in ri.items()},
        "points3D": {str(k): {"xyz": [float(x) for x in pt.xyz], "rgb": [int(x) for x in pt.rgb], "error": float(pt.error),
                              "image_ids": [int(x) for x in pt.image_ids], "point2D_idxs": [int(x) for x in pt.point2D_idxs]}
                     for k, pt in rp.items()},
        "order": {"cameras": [int(k) for k in rc], "images": [int(k) for k in ri], "points3D": [int(k) for k in rp]},
    }
    _placeholder_modules("imageio", "skimage", "skimage.transform")
    from data_utils.poses import pose_utils as ref_pu

    poses, pts3d, perm = ref_pu.load_colmap_data(root)
    out_dir = os.path.join("/tmp", "ddn_colmap_out")
    os.makedirs(out_dir, exist_ok=True)
    save_arr = ref_pu.save_poses(out_dir, poses, pts3d, perm)
    parsed["poses"] = np.asarray(poses).tolist()
    parsed["perm"] = [int(x) for x in perm]
    parsed["poses_bounds"] = np.asarray(save_arr).tolist()
    json.dump(parsed, open(os.path.join(root, "reference_parse.json"), "w"))
    print("%-40s %8.1f KB" % ("colmap_model/", sum(os.path.getsize(os.path.join(dp, f)) for dp, _, fs in os.walk(root) for f in fs) / 1024))


def gen_loaders():
    """The pose pipelines of the scene loaders (data_utils/load_llff.py:138-368, data_utils/load_blender.py:9-65): pure numpy behind
    imports of imageio / cv2, which this image lacks.  The import statements are satisfied by EMPTY placeholder modules; the one
    function that reads image files (`_load_data`) is replaced by synthetic arrays for the call, so only the reference's own
    numpy code runs: axis re-ordering, bd_factor rescale, recentring, spherify, spiral / circle render paths, hold-out view."""
    _placeholder_modules("imageio", "cv2", "skimage", "skimage.transform")
    from data_utils import load_llff as ref_llff
    from data_utils import load_blender as ref_blender

    rng = np.random.Generator(np.random.PCG64(99))
    out = {}
    N, H, W = 11, 6, 8
    for tag, cfgname, kind, spherify, bd_factor in (("llff", "config_ff.yml", "llff", False, 0.75), ("real360", "config_360.yml", "real360", False, False),
                                                    ("real360_sph", "config_360.yml", "real360", True, 0.75)):
        cfg = load_cfg(cfgname, 8, 8)
        cfg.dataset.spherify = spherify
        cfg.dataset.bd_factor = bd_factor
        cfg.dataset.basedir = "/nowhere/scene_" + tag
        # LLFF raw convention: [3,5,N] with rotation columns [down, right, back], translation, (H, W, focal)
        ang = rng.uniform(-0.4, 0.4, (N, 3)) if kind == "llff" else rng.uniform(-3.1, 3.1, (N, 3))
        poses = np.zeros((3, 5, N))
        for i in range(N):
            q, _ = np.linalg.qr(np.eye(3) + 0.5 * rng.standard_normal((3, 3)) * (0.3 if kind == "llff" else 2.0))
            if np.linalg.det(q) < 0:
                q[:, 0] = -q[:, 0]
            poses[:, :3, i] = q
            poses[:, 3, i] = rng.standard_normal(3) * (0.5 if kind == "llff" else 2.0)
            poses[:, 4, i] = (H, W, 9.5)
        bds = np.sort(rng.uniform(0.8, 9.0, (2, N)), 0)
        imgs = rng.random((H, W, 3, N))
        ref_llff._load_data = lambda basedir, factor=None, _p=poses, _b=bds, _i=imgs: (_p.copy(), _b.copy(), _i.copy())
        images, p_out, b_out, render_poses, i_test = ref_llff.load_data_after_colmap(cfg)
        out.update({tag + "_in_poses": poses, tag + "_in_bds": bds, tag + "_in_imgs": imgs.astype(np.float32), tag + "_images": images, tag + "_poses": p_out,
                    tag + "_bds": b_out, tag + "_render": np.asarray(render_poses), tag + "_itest": np.array(int(i_test)),
                    tag + "_flags": np.array([int(spherify), float(bd_factor) if bd_factor else 0.0])})
    # free-standing pose helpers
    out["pose_spherical"] = np.stack([npy(ref_blender.pose_spherical(a, -30.0, 4.0)) for a in np.linspace(-180, 180, 7)[:-1]])
    out["pose_360_beta"] = np.stack([np.asarray(ref_blender.pose_spherical_for_real_world_360(a, -10, 0.89, "beta")) for a in (0.0, 45.0, 200.0)])
    out["pose_360_other"] = np.stack([np.asarray(ref_blender.pose_spherical_for_real_world_360(a, -10, 0.89, "garden")) for a in (0.0, 45.0, 200.0)])
    save("loaders", **out)


if __name__ == "__main__":
    which = sys.argv[1:] or ["first_cycle", "encode", "composite", "samplers", "runiter", "dploss", "aten_orders", "raygen", "trainsteps", "fullsize", "manifest", "depthanalysis", "sampler4096", "grad4096", "train300", "ndcswitch",
                             "colmap", "loaders"]
    for w in which:
        globals()["gen_" + w]()
