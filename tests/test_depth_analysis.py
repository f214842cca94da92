"""The two reference features that sit beside the hot path and were missing: `run_iter(depth_analysis_validation=True)`
(models/models.py:108-112, 307-319: the per-ray density histograms the plots draw) and `dataset.combined_sampling_method`
(models/samplers.py:6-27, 45-49), against outputs of the reference itself (tests/golden/depthanalysis_*.npz,
combined_first_cycle.npz)."""
import os

import numpy as np
import pytest
import torch

from _cases import GOLDEN, relerr
from ddnerf_amd import synthetic
from ddnerf_amd.cfgnode import CfgNode

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_incell_pdfs_cpu_math():
    """the histogram builders alone (plain torch, no GPU): mass of the uniform histogram, NaN row for an empty interval"""
    from ddnerf_amd import depth_analysis as DA

    t = torch.tensor([[2.0, 3.0, 4.5, 6.0], [2.001, 2.0015, 4.0, 6.0]])   # second ray: an interval between two cell centres
    w = torch.tensor([[0.2, 0.5, 0.3], [0.1, 0.6, 0.3]])
    u = DA.uniform_incell_pdf(t, w, 2.0, 6.0)
    assert u.shape == (2, 1000) and abs(float(u[0].sum()) - 1.0) < 1e-5
    assert torch.isnan(u[1]).all()          # an interval without a cell centre poisons its row, as upstream
    g = DA.gaussian_incell_pdf(t[:1], w[:1], torch.full((1, 3), 0.5), torch.full((1, 3), 0.2), torch.full((1, 3), 0.98), 2.0, 6.0)
    assert g.shape == (1, 1000) and torch.isfinite(g).all() and 0.9 < float(g.sum()) < 1.1


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["depthanalysis_dd_blender", "depthanalysis_mip_blender"])
def test_depth_analysis_outputs_match_reference(name):
    from models import models

    g = dict(np.load(os.path.join(GOLDEN, name + ".npz")))
    dd = "_dd_" in name
    nc, nf, sharpen, _noise, near, far, dist_reg, smooth, pad = g["meta"]
    cfg = CfgNode.load(os.path.join(ROOT, "configs", "config_blender.yml" if dd else "config_blender_mipnerf.yml"))
    for mode in ("train", "validation"):
        cfg.nerf[mode].update(num_coarse=int(nc), num_fine=int(nf), radiance_field_noise_std=0.0)
    cfg.train_params.dist_reg_coeficient = float(dist_reg)
    model = getattr(models, cfg.nerf.type)(cfg)
    model.coarse.load_state_dict({k: torch.from_numpy(v) for k, v in synthetic.make_state_dict(dd, 11, float(sharpen)).items()})
    if dd:
        model.fine.load_state_dict({k: torch.from_numpy(v) for k, v in synthetic.make_state_dict(False, 12, float(sharpen)).items()})
    model.to("cuda")
    model.eval()
    d = lambda x: torch.from_numpy(x).cuda()
    with torch.no_grad():
        out = model.run_iter(d(g["ro"]), d(g["rd"]), d(g["rad"]), mode="validation", depth_analysis_validation=True)
    want = {k for k in g if k.startswith("o")}
    assert {"o%d_%s" % (lvl, k) for lvl in out for k in out[lvl] if "plot" in k} == {k for k in want if "plot" in k}
    assert out[1]["rgb"].shape == (6, 3)                     # no image reshape in this mode (models/models.py:64)
    for k in sorted(want):
        lvl, key = int(k[1]), k[3:]
        a, b = out[lvl][key].cpu().numpy(), g[k]
        if key.endswith("incell_pdf_to_plot"):
            assert a.shape == b.shape and np.array_equal(np.isnan(a), np.isnan(b)), k
            assert np.nanmax(np.abs(a - b)) <= 2e-4 * max(1.0, float(np.nanmax(np.abs(b)))), (k, np.nanmax(np.abs(a - b)))
        else:
            assert relerr(a, b) <= 1e-4, k


@pytest.mark.gpu
def test_combined_sampling_matches_reference():
    from ddnerf_amd import models as M, ops

    g = dict(np.load(os.path.join(GOLDEN, "combined_first_cycle.npz")))
    nc, near, far, split = g["meta"]
    row = M._combined_row(float(near), float(split), float(far), int(nc), "cuda")
    rays = torch.zeros(5, 12, device="cuda")
    rays[:, 7], rays[:, 8] = float(near), float(far)
    t = ops.sample_first_cycle(rays, row, None, 2).cpu().numpy()
    assert np.array_equal(t, g["t_validation"])                                   # bit for bit
    t = ops.sample_first_cycle(rays, row, torch.from_numpy(g["rnd_train"]).cuda(), 2).cpu().numpy()
    assert np.array_equal(t, g["t_train"])
