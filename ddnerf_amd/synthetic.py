"""Seeded synthetic rays and network weights (numpy only, no torch import).

There are no datasets or checkpoints in the build/bench environment, so every
parity test, golden vector and benchmark runs on rays and weights produced
here from a seed.  The same generators feed (a) the golden-vector script that
imports the reference, (b) the C oracle, (c) the HIP path — so fixtures never
have to store the 2 x 2.4 MB of weights.

Shapes follow the reference:
  * rays: origins [N,3], directions [N,3] (NOT normalised by the callers that
    matter: ``data_utils/dataset.py`` hands un-normalised pinhole directions to
    ``run_iter``), radii [N,1]   (reference ``models/models.py:40``)
  * weights: ``MipNeRFModel`` / ``DepthMipNeRFModel`` state-dict names and
    shapes (reference ``models/base_architectures.py:22-37,83-99``)
"""
from __future__ import annotations

import numpy as np

# (name, out_features, in_features) in reference registration order
# reference models/base_architectures.py:24-37 (fine) and :85-99 (coarse DD)
_LAYERS_COMMON = (
    [("layers_xyz.0", 256, 96)]
    + [("layers_xyz.%d" % i, 256, 352 if i == 5 else 256) for i in range(1, 8)]
    + [("fc_feat", 256, 256), ("fc_alpha", 1, 256), ("layers_dir.0", 128, 283), ("fc_rgb", 3, 128)]
)
LAYERS_FINE = tuple(_LAYERS_COMMON)
LAYERS_COARSE_DD = tuple(_LAYERS_COMMON + [("fc_mu_sigma", 2, 128)])


def layer_table(depth_head: bool):
    """(name, out, in) triples of MipNeRFModel (False) or DepthMipNeRFModel (True)."""
    return LAYERS_COARSE_DD if depth_head else LAYERS_FINE


def make_state_dict(depth_head: bool, seed: int, sharpen: float = 1.0) -> dict:
    """Closed-form seeded weights with torch.nn.Linear's default init *semantics*
    (uniform +-1/sqrt(fan_in) for weight and bias).  ``sharpen`` multiplies
    ``fc_alpha.weight`` so compositing weights become peaky and the sampler's
    tails and clamps are exercised (SURVEY.md 8d weight set B uses 20)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    sd = {}
    for name, n_out, n_in in layer_table(depth_head):
        bound = 1.0 / np.sqrt(n_in)
        sd[name + ".weight"] = rng.uniform(-bound, bound, size=(n_out, n_in)).astype(np.float32)
        sd[name + ".bias"] = rng.uniform(-bound, bound, size=(n_out,)).astype(np.float32)
    if sharpen != 1.0:
        sd["fc_alpha.weight"] = (sd["fc_alpha.weight"] * np.float32(sharpen)).astype(np.float32)
    return sd


def _unit(v):
    return v / np.linalg.norm(v, axis=-1, keepdims=True)


def make_rays(kind: str, n: int, seed: int):
    """Synthetic ray bundle (origins, directions, radii, rgb targets), float32.

    kind:
      'blender'  cameras on the z>0 hemisphere of radius 4 looking at the origin,
                 radius 2/(sqrt(12)*1111.11) (800-px Lego focal, pixel-footprint
                 rule of reference general_utils/nerf_helpers.py:117-123); near 2, far 6
      'llff'     NDC-like forward-facing rays, near 0, far 1
      'real360'  as blender with |o| ~ 0.8 (poses normalised by 5), near 0.2, far 2.8
    """
    rng = np.random.Generator(np.random.PCG64(seed))
    if kind in ("blender", "real360"):
        o = _unit(rng.standard_normal((n, 3)))
        o[:, 2] = np.abs(o[:, 2])
        o = o * (4.0 if kind == "blender" else 0.8)
        d = -o / np.linalg.norm(o, axis=-1, keepdims=True) + 0.25 * rng.standard_normal((n, 3))
        d = _unit(d)
        # pinhole bundles are not unit length: scale like 1/cos of the off-axis angle
        d = d * (1.0 + 0.15 * rng.random((n, 1)))
        rad = np.full((n, 1), 2.0 / (np.sqrt(12.0) * 1111.11))
    elif kind == "llff":
        o = np.concatenate([rng.uniform(-1, 1, (n, 2)), -np.ones((n, 1))], -1)
        d = np.concatenate([rng.uniform(-0.3, 0.3, (n, 2)), 2.0 * np.ones((n, 1))], -1)
        rad = np.full((n, 1), 1.0e-3)
    else:
        raise ValueError(kind)
    o = o.astype(np.float32)
    d = d.astype(np.float32)
    # the reference nudges exact zeros by 1e-5 (general_utils/nerf_helpers.py:114-115)
    o[o == 0] += np.float32(1e-5)
    d[d == 0] += np.float32(1e-5)
    tgt = rng.random((n, 3)).astype(np.float32)
    return o, d, rad.astype(np.float32), tgt


NEAR_FAR = {"blender": (2.0, 6.0), "llff": (0.0, 1.0), "real360": (0.2, 2.8)}


def procedural_targets(o, d):
    """A smooth analytic colour per ray (a function of the camera position and the unit viewing direction): targets a network can
    actually fit, for the training-parity trajectory (tests/golden/make_golden.py gen_train300; SURVEY.md 8d "PSNR vs ref (2)")."""
    o = np.asarray(o, np.float64)
    u = np.asarray(d, np.float64)
    u = u / np.linalg.norm(u, axis=-1, keepdims=True)
    ph = 2.0 * u * np.array([1.0, 2.0, 3.0]) + o * np.array([0.7, 0.5, 0.3])
    rgb = 0.5 + 0.4 * np.sin(ph) * np.cos(ph[:, [1, 2, 0]] * 0.5)
    return rgb.astype(np.float32)
