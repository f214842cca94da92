"""GPU parity at BASELINE.json's stated sizes against outputs of the REFERENCE itself (tests/golden/fullsize_*.npz, generated
by importing the reference on the same seeded rays and weights; every 61st / 125th ray is stored):
  cfg1 config_blender.yml 256 rays x 64 x 64, cfg2 config_blender.yml 4096 x (64 + 128), cfg3 config_ff.yml NDC rays,
  cfg4 config_360.yml 8192 rays, cfg5 config_blender_mipnerf.yml (one shared MLP); "trained": 4096 blender / NDC rays through the
  weights of the reference's own 3000-iteration training run (PSNR 35 dB on its scene) instead of seeded-uniform ones.
fp32 and x3 kernels: RGB / depth / acc / weights within 1e-4 (north_star); the bf16 kernel on its tier (cfg3 is BASELINE's
bf16 configuration)."""
import numpy as np
import pytest
import torch

from _cases import fullsize_names, load_fullsize, maxerr, relerr
from ddnerf_amd import synthetic

pytestmark = pytest.mark.gpu


def _run(c, mlp_dtype):
    from test_hip_run_iter import build_model

    model = build_model(c)
    model.cfg.nerf["mlp_dtype"] = mlp_dtype
    model._set_mlp_dtype()
    model.eval()
    ro, rd, rad, tgt = (torch.from_numpy(x).cuda() for x in synthetic.make_rays(c["kind"], c["n"], 1))
    with torch.no_grad():
        return model.run_iter(ro, rd, rad, mode="validation", rgb_target=tgt)


@pytest.mark.parametrize("mlp_dtype", ["fp32", "x3"])
@pytest.mark.parametrize("name", fullsize_names())
def test_full_size_matches_reference(name, mlp_dtype):
    c = load_fullsize(name)
    g, st = c["g"], c["stride"]
    out = _run(c, mlp_dtype)
    assert len(out) == 2
    for lvl in (0, 1):
        for k in ("rgb", "depth", "acc", "disp", "weights"):
            # north_star: RGB / depth within 1e-4 -- ABSOLUTE (depth reaches far = 6); only the disparity, which reaches 1e10 on
            # empty rays, is held relatively
            e = (relerr if k == "disp" else maxerr)(out[lvl][k][::st].cpu().numpy(), g["o%d_%s" % (lvl, k)])
            assert e <= 1e-4, (name, lvl, k, e)
    if c["dd"]:
        ref = float(g["o1_dp_loss"][0])
        assert abs(float(out[1]["dp_loss"][0]) - ref) <= 2e-4 * max(abs(ref), 1e-2)
        for k in ("mus_reg", "sig_reg"):
            assert relerr(out[0][k].cpu().numpy(), g["o0_" + k]) <= 2e-5, k


@pytest.mark.parametrize("name", [n for n in fullsize_names() if "cfg3" in n or "cfg2" in n or "trained" in n])
def test_full_size_bf16_tier(name):
    """bf16-MFMA MLP at full size against the reference's fp32 outputs.  Seeded-uniform weights (cfg2, cfg3): measured <= 3e-4 RGB /
    78-93 dB; the bar is 3x that.  TRAINED weights (the reference's own 3000-iteration run; round 5, tools/tier_errors.py): the errors
    are 3-6x larger -- RGB <= 1.9e-3, depth <= 2.1e-3 absolute, weights <= 1.5e-3 -- because a trained network's activations are large
    where the scene is (bf16 keeps 8 bits of each); held to 3e-3 / 4e-3 absolute and 60 dB."""
    c = load_fullsize(name)
    g, st = c["g"], c["stride"]
    out = _run(c, "bf16")
    trained = c["tag"] == "trained"
    for lvl in (0, 1):
        rgb, ref = out[lvl]["rgb"][::st].cpu().numpy(), g["o%d_rgb" % lvl]
        psnr = -10.0 * np.log10(max(float(np.mean((rgb - ref) ** 2)), 1e-20))
        depth, dref = out[lvl]["depth"][::st].cpu().numpy(), g["o%d_depth" % lvl]
        got = (psnr, float(np.abs(rgb - ref).max()), float(np.abs(depth - dref).max()))
        if trained:
            assert psnr >= 60.0 and got[1] <= 3e-3 and got[2] <= 4e-3, (lvl, got)
        else:
            assert psnr >= 75.0, (lvl, got)
            assert got[1] <= 1e-3, (lvl, got)
            assert got[2] <= 2.5e-3 * max(1.0, np.abs(dref).max()), (lvl, got)
