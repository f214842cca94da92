"""GPU parity at BASELINE.json's stated sizes against outputs of the REFERENCE itself (tests/golden/fullsize_*.npz, generated
by importing the reference on the same seeded rays and weights; every 61st / 125th ray is stored):
  cfg1 config_blender.yml 256 rays x 64 x 64, cfg2 config_blender.yml 4096 x (64 + 128), cfg3 config_ff.yml NDC rays,
  cfg4 config_360.yml 8192 rays, cfg5 config_blender_mipnerf.yml (one shared MLP).
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


@pytest.mark.parametrize("name", [n for n in fullsize_names() if "cfg3" in n or "cfg2" in n])
def test_full_size_bf16_tier(name):
    """bf16-MFMA MLP at full size against the reference's fp32 outputs: measured <= 3e-4 RGB / 78-93 dB on the small fixtures;
    the bar is 3x that."""
    c = load_fullsize(name)
    g, st = c["g"], c["stride"]
    out = _run(c, "bf16")
    for lvl in (0, 1):
        rgb, ref = out[lvl]["rgb"][::st].cpu().numpy(), g["o%d_rgb" % lvl]
        psnr = -10.0 * np.log10(max(float(np.mean((rgb - ref) ** 2)), 1e-20))
        assert psnr >= 75.0, (lvl, psnr)
        assert np.abs(rgb - ref).max() <= 1e-3, (lvl, np.abs(rgb - ref).max())
        depth, dref = out[lvl]["depth"][::st].cpu().numpy(), g["o%d_depth" % lvl]
        assert np.abs(depth - dref).max() <= 2.5e-3 * max(1.0, np.abs(dref).max()), lvl
