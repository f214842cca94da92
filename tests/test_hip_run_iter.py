"""GPU parity of the whole path through the drop-in `models.models` surface: `run_iter` of the HIP-backed
models against (a) the golden outputs produced by the reference on the same rays / weights / random
tensors and (b) the CPU oracle.  north_star bar: RGB / depth within 1e-4 in fp32."""
import numpy as np
import pytest
import torch

from _cases import load_runiter, maxerr, relerr, runiter_names
from ddnerf_amd.cfgnode import CfgNode

pytestmark = pytest.mark.gpu

CFG_OF = {("dd", "blender"): "config_blender.yml", ("dd", "llff"): "config_ff.yml", ("dd", "real360"): "config_360.yml",
          ("mip", "blender"): "config_blender_mipnerf.yml", ("mip", "llff"): "config_ff_mipnerf.yml"}


class ReplayRng:
    """Hands out the reference's random tensors (from the fixture) in draw order, checking kind and shape."""

    def __init__(self, draws):
        self.draws = list(draws)

    def _next(self, kind, shape, device):
        k, t = self.draws.pop(0)
        assert k == kind and tuple(t.shape) == tuple(shape), (k, kind, t.shape, shape)
        return torch.from_numpy(t).to(device)

    def rand(self, shape, device):
        return self._next("rand", shape, device)

    def randn(self, shape, device):
        return self._next("randn", shape, device)


def build_model(c):
    import os
    from models import models  # the drop-in alias

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = CfgNode.load(os.path.join(root, "configs", CFG_OF[("dd" if c["dd"] else "mip", c["kind"])]))
    for mode in ("train", "validation"):
        cfg.nerf[mode]["num_coarse"] = c["nc"]
        cfg.nerf[mode]["num_fine"] = c["nf"]
        cfg.nerf[mode]["radiance_field_noise_std"] = c["noise"]
    cfg.dataset.near, cfg.dataset.far = c["near"], c["far"]
    cfg.train_params.dist_reg_coeficient = c["dist_reg"]
    cfg.train_params.gaussian_smooth_factor = c["smooth"]
    cfg.train_params.pdf_padding = c["pdf_padding"]
    if c.get("dp_coef") is not None:
        cfg.train_params.dp_coeficient = c["dp_coef"]
    model = getattr(models, cfg.nerf.type)(cfg)
    model.coarse.load_state_dict({k: torch.from_numpy(v) for k, v in c["sd_coarse"].items()})
    if c["dd"]:
        model.fine.load_state_dict({k: torch.from_numpy(v) for k, v in c["sd_fine"].items()})
    model.to("cuda")
    g = c["g"]
    if not any(k.startswith("rnd") for k in g) and not c["train"] and c["noise"] == 0:
        model.rng = ReplayRng([])  # nothing random in this pass
        return model
    draws = []
    rnd = [g[k] for k in sorted((k for k in g if k.startswith("rnd")), key=lambda s: int(s[3:]))]
    it = iter(rnd)
    if c["train"]:
        draws.append(("rand", next(it)))
    if c["noise"] > 0:
        draws.append(("randn", next(it)))
    if c["train"]:
        draws.append(("rand", next(it)))
    if c["noise"] > 0:
        draws.append(("randn", next(it)))
    model.rng = ReplayRng(draws)
    return model


@pytest.mark.parametrize("name", runiter_names())
def test_run_iter_forward_matches_reference(name):
    c = load_runiter(name)
    g = c["g"]
    model = build_model(c)
    d = lambda x: torch.from_numpy(x).cuda()
    if c["train"]:
        model.train()
    else:
        model.eval()
    with torch.no_grad():
        out = model.run_iter(d(g["ro"]), d(g["rd"]), d(g["rad"]), mode=c["mode"], rgb_target=d(g["tgt"]))
    assert model.rng.draws == []
    assert len(out) == 2
    for lvl in (0, 1):
        for k in ("rgb", "depth", "disp", "acc", "weights"):
            # ABSOLUTE error for everything but the disparity (1e10 on empty rays): depth reaches far = 6
            e = (relerr if k == "disp" else maxerr)(out[lvl][k].cpu().numpy(), g["o%d_%s" % (lvl, k)])
            assert e <= 1e-4, (lvl, k, e)          # north_star bar
            if lvl == 0:
                assert e <= 5e-6, (lvl, k, e)      # the coarse pass has no sampler in front of it: near-ulp
    if c["dd"]:
        assert set(out[0].keys()) == {"rgb", "disp", "acc", "weights", "depth", "mus", "sigmas", "dp_loss",
                                      "corrected_disp_map", "smoothed_sigmas", "mus_loss", "sig_loss", "mus_reg",
                                      "sig_reg"}
        assert out[0]["dp_loss"] is None and out[1]["corrected_disp_map"] is None
        for k in ("mus", "sigmas", "smoothed_sigmas"):
            a, b = out[0][k].cpu().numpy(), g["o0_" + k]
            assert a.shape == b.shape and (a.size == 0 or np.abs(a - b).max() <= 5e-6), k
            assert np.array_equal(out[1][k].cpu().numpy(), a)        # stale level-0 record, like the reference
        assert relerr(out[0]["corrected_disp_map"].cpu().numpy(), g["o0_corrected_disp_map"]) <= 5e-6
        for k in ("mus_loss", "sig_loss", "mus_reg", "sig_reg"):
            assert relerr(out[0][k].cpu().numpy(), g["o0_" + k]) <= 5e-6, k
        ref = float(g["o1_dp_loss"][0])
        assert abs(float(out[1]["dp_loss"][0]) - ref) <= 1e-4 * max(abs(ref), 1e-2)
    else:
        assert set(out[0].keys()) == {"rgb", "disp", "acc", "weights", "depth"}


@pytest.mark.parametrize("name", runiter_names())
def test_run_iter_x3_meets_the_fp32_bar(name):
    """`nerf.mlp_dtype: x3` (bf16 matrix cores, exact hi/lo operand splits) is held to the SAME bar as the exact fp32
    kernel: RGB / depth / weights within 1e-4 of the reference on every fixture."""
    c = load_runiter(name)
    g = c["g"]
    model = build_model(c)
    model.cfg.nerf["mlp_dtype"] = "x3"
    model._set_mlp_dtype()
    d = lambda x: torch.from_numpy(x).cuda()
    model.train() if c["train"] else model.eval()
    with torch.no_grad():
        out = model.run_iter(d(g["ro"]), d(g["rd"]), d(g["rad"]), mode=c["mode"], rgb_target=d(g["tgt"]))
    for lvl in (0, 1):
        for k in ("rgb", "depth", "disp", "acc", "weights"):
            e = (relerr if k == "disp" else maxerr)(out[lvl][k].cpu().numpy(), g["o%d_%s" % (lvl, k)])
            assert e <= 1e-4, (lvl, k, e)


@pytest.mark.parametrize("name", ["runiter_dd_blender_64x128_validation", "runiter_dd_llff_16x16_validation", "runiter_mip_blender_64x128_validation"])
def test_run_iter_bf16_tier(name):
    """The bf16-MFMA MLP is its own tolerance tier ('expect ~1e-2 / >= 40 dB; a tolerance tier to be
    fixed empirically' -- fixed here at 75 dB / 1e-3, three times the measured error): same rays / weights as the fp32 fixtures, RGB PSNR against the
    reference's fp32 output."""
    if name not in runiter_names():
        pytest.skip("fixture not generated")
    c = load_runiter(name)
    g = c["g"]
    model = build_model(c)
    model.cfg.nerf["mlp_dtype"] = "bf16"
    model._set_mlp_dtype()
    model.eval()
    d = lambda x: torch.from_numpy(x).cuda()
    with torch.no_grad():
        out = model.run_iter(d(g["ro"]), d(g["rd"]), d(g["rad"]), mode=c["mode"], rgb_target=d(g["tgt"]))
    for lvl in (0, 1):
        rgb, ref = out[lvl]["rgb"].cpu().numpy(), g["o%d_rgb" % lvl]
        psnr = -10.0 * np.log10(max(float(np.mean((rgb - ref) ** 2)), 1e-20))
        assert psnr >= 75.0, (lvl, psnr)                      # measured 78-93 dB
        assert np.abs(rgb - ref).max() <= 1e-3, (lvl, np.abs(rgb - ref).max())   # measured <= 2.7e-4: the bar is ~3x that
        depth, dref = out[lvl]["depth"].cpu().numpy(), g["o%d_depth" % lvl]
        assert np.abs(depth - dref).max() <= 2.5e-3 * max(1.0, np.abs(dref).max()), lvl   # measured <= 6.9e-4


def test_validation_reshape_and_chunking():
    """image-shaped validation input, ray chunks smaller than the image: outputs are reshaped / concatenated
    like models/models.py:53-72, and equal the unchunked result."""
    c = load_runiter("runiter_dd_blender_32x32_validation")
    g = c["g"]
    model = build_model(c)
    model.rng = __import__("ddnerf_amd.models", fromlist=["TorchRng"]).TorchRng()
    model.eval()
    d = lambda x: torch.from_numpy(x).cuda()
    ro, rd, rad = d(g["ro"]).view(4, 6, 3), d(g["rd"]).view(4, 6, 3), d(g["rad"]).view(4, 6, 1)
    with torch.no_grad():
        full = model.run_iter(ro, rd, rad, mode="validation")
        model.cfg.nerf.validation.chunksize = 7
        chunked = model.run_iter(ro, rd, rad, mode="validation")
    assert full[1]["rgb"].shape == (4, 6, 3) and full[1]["depth"].shape == (4, 6)
    assert full[0]["corrected_disp_map"].shape == (4, 6)
    assert chunked[1]["dp_loss"].shape == (4,) and chunked[0]["mus_reg"].shape == (4,)
    for k in ("rgb", "depth", "weights", "acc", "disp"):
        assert torch.equal(full[1][k], chunked[1][k]), k


def test_missing_library_fails_loudly(monkeypatch):
    from ddnerf_amd import _lib

    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "SO_PATH", "/nonexistent/libddnerf_hip.so")
    with pytest.raises(_lib.DDNerfHipError):
        _lib.lib()


@pytest.mark.parametrize("mlp_dtype", ["fp32", "x3", "bf16"])
def test_full_size_run_iter_properties(mlp_dtype):
    """BASELINE config 2 (4096 rays x (64 + 128)) through the drop-in model on each MLP kernel: size-independent properties.
    Rays are independent, so (a) permuting them permutes the outputs bit for bit, (b) rendering in chunks of 1000 rays equals
    rendering at once, bit for bit, (c) the run is reproducible (no atomics / races), and (d) the x3 / bf16 kernels agree with
    the exact one within their tiers at full size."""
    import os
    from models import models
    from ddnerf_amd import synthetic

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = CfgNode.load(os.path.join(root, "configs", "config_blender.yml"))
    for m in ("train", "validation"):
        cfg.nerf[m].update(num_coarse=64, num_fine=128, radiance_field_noise_std=0.0, perturb=False)
    cfg.train_params.dist_reg_coeficient = 1 / 64

    def build(dt):
        cfg.nerf["mlp_dtype"] = dt
        model = getattr(models, cfg.nerf.type)(cfg)
        model.coarse.load_state_dict({k: torch.from_numpy(v) for k, v in synthetic.make_state_dict(True, 11, 20.0).items()})
        model.fine.load_state_dict({k: torch.from_numpy(v) for k, v in synthetic.make_state_dict(False, 12, 20.0).items()})
        model.to("cuda")
        model.eval()
        return model

    model = build(mlp_dtype)
    ro, rd, rad, _ = (torch.from_numpy(x).cuda() for x in synthetic.make_rays("blender", 4096, 1))
    with torch.no_grad():
        a = model.run_iter(ro, rd, rad, mode="validation")
        b = model.run_iter(ro, rd, rad, mode="validation")
        perm = torch.randperm(4096, device="cuda")
        p = model.run_iter(ro[perm], rd[perm], rad[perm], mode="validation")
        model.cfg.nerf.validation.chunksize = 1000
        c = model.run_iter(ro, rd, rad, mode="validation")
        model.cfg.nerf.validation.chunksize = 16384
    for lvl in (0, 1):
        for k in ("rgb", "depth", "weights", "acc", "disp"):
            assert torch.isfinite(a[lvl][k]).all(), (lvl, k)
            assert torch.equal(a[lvl][k], b[lvl][k]), ("reproducible", lvl, k)
            assert torch.equal(a[lvl][k][perm], p[lvl][k]), ("permutation", lvl, k)
            assert torch.equal(a[lvl][k], c[lvl][k]), ("chunking", lvl, k)
    assert float(a[1]["acc"].max()) <= 1.0 + 1e-5 and float(a[1]["weights"].min()) >= 0.0
    if mlp_dtype != "fp32":
        ref = build("fp32")
        with torch.no_grad():
            r = ref.run_iter(ro, rd, rad, mode="validation")
        tol = 1e-4 if mlp_dtype == "x3" else 3e-3
        for k in ("rgb", "depth"):
            assert float((a[1][k] - r[1][k]).abs().max()) <= tol * max(1.0, float(r[1][k].abs().max())), (k, mlp_dtype)
