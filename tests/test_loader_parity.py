"""The scene loaders' pose pipelines against the REFERENCE's own outputs (tests/golden/loaders.npz, written by
tests/golden/make_golden.py gen_loaders from data_utils/load_llff.py:277-368 and data_utils/load_blender.py:36-65): the same
synthetic poses_bounds arrays go through ddnerf_amd.llff.load_llff -- axis re-ordering, bd_factor rescale, recentring, spherify,
spiral / circle render paths, hold-out view.  (Image decoding is the one step replaced on both sides: property tests in
tests/test_llff_loader.py / tests/test_blender_loader.py cover it.)  Also switch_t_ndc_to_regular on the GPU."""
import os

import numpy as np
import pytest

from ddnerf_amd import llff
from ddnerf_amd.cfgnode import CfgNode

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("tag,cfgname", [("llff", "config_ff.yml"), ("real360", "config_360.yml"), ("real360_sph", "config_360.yml")])
def test_llff_pipeline_matches_the_reference(golden, monkeypatch, tag, cfgname):
    g = golden("loaders")
    cfg = CfgNode.load(os.path.join(ROOT, "configs", cfgname))
    spherify, bd = g[tag + "_flags"]
    cfg.dataset.spherify = bool(spherify)
    cfg.dataset.bd_factor = float(bd) if bd else False
    cfg.dataset.basedir = "/nowhere/scene_" + tag
    monkeypatch.setattr(llff, "read_scene", lambda basedir, factor=None: (g[tag + "_in_poses"].copy(), g[tag + "_in_bds"].copy(),
                                                                          g[tag + "_in_imgs"].astype(np.float64)))
    images, poses, bds, render_poses, i_test = llff.load_llff(cfg)
    assert i_test == int(g[tag + "_itest"])
    assert images.dtype == np.float32 and np.array_equal(images, g[tag + "_images"])
    assert poses.dtype == np.float32 and poses.shape == g[tag + "_poses"].shape
    assert np.allclose(poses, g[tag + "_poses"], rtol=0, atol=2e-6)
    assert np.allclose(bds, g[tag + "_bds"], rtol=2e-6, atol=0)
    ref_render = g[tag + "_render"]
    assert render_poses.shape == ref_render.shape
    assert np.allclose(render_poses, ref_render, rtol=0, atol=3e-6)


def test_pose_helpers_match_the_reference(golden):
    from ddnerf_amd import data

    g = golden("loaders")
    got = np.stack([data._pose_spherical(a, -30.0, 4.0) for a in np.linspace(-180, 180, 7)[:-1]])
    assert np.allclose(got, g["pose_spherical"], rtol=0, atol=1e-6)
    for name, key in (("beta", "pose_360_beta"), ("garden", "pose_360_other")):
        got = np.stack([llff.real360_pose(a, -10, 0.89, name) for a in (0.0, 45.0, 200.0)])
        assert np.allclose(got, g[key], rtol=0, atol=1e-6), name


@pytest.mark.gpu
def test_ndc_depth_to_regular(golden):
    import torch

    from ddnerf_amd import ops

    g = golden("ndcswitch")
    d = lambda k: torch.from_numpy(g[k]).cuda()
    got = ops.ndc_depth_to_regular(d("ndc_depth"), d("ro"), d("rd")).cpu().numpy()
    ref = g["regular"]
    assert got.shape == ref.shape
    assert np.all(np.abs(got - ref) <= 4e-7 * np.maximum(1.0, np.abs(ref)))   # one fp32 rounding of a quotient near a pole
