"""Helpers shared by the CPU (oracle) and GPU (HIP) parity tests: rebuild the exact inputs of a
golden `runiter_*` case (weights from seeds, random tensors from the fixture, torch linspace rows)."""
import glob
import os

import numpy as np
import torch

from ddnerf_amd import synthetic

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def runiter_names():
    return sorted(os.path.basename(f)[:-4] for f in glob.glob(os.path.join(GOLDEN, "runiter_*.npz")))


def load_runiter(name):
    g = dict(np.load(os.path.join(GOLDEN, name + ".npz")))
    _, mt, kind, _, mode = name.split("_")
    nc, nf, sharpen, noise, near, far, dist_reg, smooth, pad = g["meta"]
    dd = mt == "dd"
    train = mode == "train"
    rnd = [g[k] for k in sorted((k for k in g if k.startswith("rnd")), key=lambda s: int(s[3:]))]
    it = iter(rnd)
    # draw order of the reference, per chunk: rand(first cycle) / randn(coarse) / rand(sampler) / randn(fine)
    t_rand = next(it) if train else None
    noise0 = next(it) * np.float32(noise) if noise > 0 else None
    u_rand = next(it) if train else None
    noise1 = next(it) * np.float32(noise) if noise > 0 else None
    case = dict(
        g=g, dd=dd, kind=kind, mode=mode, train=train, nc=int(nc), nf=int(nf), sharpen=float(sharpen), noise=float(noise),
        near=float(near), far=float(far), dist_reg=float(dist_reg), smooth=float(smooth), pdf_padding=bool(pad),
        blender=(kind == "blender"), t_rand=t_rand, noise0=noise0, u_rand=u_rand, noise1=noise1,
        t_lin=torch.linspace(0.0, 1.0, int(nc) + 1).numpy(),
        u_det=torch.linspace(0.0, 0.9999 if dd else 1.0, int(nf) + 1).numpy(),
        sd_coarse=synthetic.make_state_dict(dd, 11, sharpen),
        sd_fine=synthetic.make_state_dict(False, 12, sharpen) if dd else None,
        dp_coef=float(g["dp_coef"]) if "dp_coef" in g else None,
    )
    return case


def fullsize_names():
    return sorted(os.path.basename(f)[:-4] for f in glob.glob(os.path.join(GOLDEN, "fullsize_*.npz")))


def load_fullsize(name):
    """BASELINE-size validation fixtures (tests/golden/make_golden.py gen_fullsize): every `stride`-th ray's outputs of the
    reference; rays and weights come from the synthetic seeds."""
    g = dict(np.load(os.path.join(GOLDEN, name + ".npz")))
    _, tag, mt, kind, n, _ = name.split("_")
    n_, nc, nf, sharpen, stride, near, far, dist_reg, smooth, pad = g["meta"]
    assert int(n_) == int(n)
    dd = mt == "dd"
    if tag == "trained":     # TRAINED weights: the reference's own 3000-iteration run (make_golden.py gen_trained), not the seeded ones
        sd_coarse, sd_fine = trained_state_dicts()
    else:
        sd_coarse = synthetic.make_state_dict(dd, 11, float(sharpen))
        sd_fine = synthetic.make_state_dict(False, 12, float(sharpen)) if dd else None
    return dict(g=g, tag=tag, dd=dd, kind=kind, n=int(n), nc=int(nc), nf=int(nf), stride=int(stride), near=float(near), far=float(far),
                dist_reg=float(dist_reg), smooth=float(smooth), pdf_padding=bool(pad), noise=0.0, train=False, mode="validation",
                sd_coarse=sd_coarse, sd_fine=sd_fine, dp_coef=None)


def trained_state_dicts():
    """(coarse, fine) state_dicts after 3000 iterations of the reference's training loop (tests/golden/trained_weights_dd_blender.npz)"""
    w = np.load(os.path.join(GOLDEN, "trained_weights_dd_blender.npz"))
    return ({k[2:]: w[k] for k in w.files if k.startswith("c.")}, {k[2:]: w[k] for k in w.files if k.startswith("f.")})


def maxerr(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    if a.size == 0:
        return 0.0
    assert np.array_equal(np.isnan(a), np.isnan(b)), "NaN pattern differs"
    return float(np.nanmax(np.abs(a - b))) if np.isfinite(a).any() else 0.0


def relerr(a, b):
    """max |a-b| / max(1, |b|): absolute below 1, relative above (disparities reach 1e10)."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    if a.size == 0:
        return 0.0
    assert np.array_equal(np.isnan(a), np.isnan(b)), "NaN pattern differs"
    return float(np.nanmax(np.abs(a - b) / np.maximum(1.0, np.abs(b))))
