"""Pins the CPU oracle (oracle/) against golden vectors produced by importing the reference
(tests/golden/make_golden.py).  Integer/index results must be bit-exact; float results are held to a
few fp32 ulps (the reference's torch CPU kernels use SLEEF sin/exp/erf, the oracle uses libm)."""
import numpy as np
import pytest

import oracle as O
from _cases import load_runiter, maxerr, runiter_names
from ddnerf_amd import synthetic

SIZES = (1, 5, 16, 17, 31, 32, 33, 64, 65, 128, 129, 130)


def test_aten_reduction_orders_bit_exact(golden):
    g = golden("aten_orders")
    for n in SIZES:
        x = g["x%d" % n]
        assert np.array_equal(O.aten_sum(x), g["sum%d" % n]), n
        assert np.array_equal(O.aten_cumsum(x), g["cumsum%d" % n]), n
        assert np.array_equal(O.aten_cumsum(1 - 0.5 * x, prod=True), g["cumprod%d" % n]), n


@pytest.mark.parametrize("tag", ["lin", "disp", "ndc"])
def test_first_cycle_bit_exact(golden, tag):
    g = golden("first_cycle")
    near, far, nc, lind = g[tag + "_meta"]
    rays = np.zeros((19, 12), np.float32)
    rays[:, 7], rays[:, 8] = near, far
    for mode, pert in (("train", True), ("validation", False)):
        t = O.sample_first_cycle(rays, g[tag + "_lin"], g[tag + "_train_rand"] if pert else None, bool(lind))
        assert np.array_equal(t, g["%s_%s_t" % (tag, mode)])


@pytest.mark.parametrize("name", ["blender_cone", "llff_cone", "real360_cone", "blender_cylinder"])
def test_encode_and_mlp(golden, name):
    g = golden("encode_" + name)
    cyl = name.endswith("cylinder")
    rays = O.pack_rays(g["ro"], g["rd"], g["rad"], float(g["near"]), float(g["far"]))
    assert maxerr(rays, g["rays"]) <= 1.2e-7
    means, covs = O.cast_rays(g["rays"], g["t_vals"], cyl)
    assert np.array_equal(means, g["means"])  # the 2^15-gain input of the IPE must be bit-exact
    assert np.max(np.abs(covs - g["covs"]) / np.abs(g["covs"])) <= 2.5e-7
    assert maxerr(O.ipe(g["means"], g["covs"]), g["ipe"]) <= 2.4e-7
    assert maxerr(O.dir_enc(g["rays"][:, 9:12]), g["dirs"]) <= 1.2e-7
    feat = O.encode(g["rays"], g["t_vals"], cyl)
    assert maxerr(feat[:, :96], g["ipe"].reshape(-1, 96)) <= 2.4e-7
    r6 = O.mlp_forward(feat, synthetic.make_state_dict(True, 11), True)
    r4 = O.mlp_forward(feat, synthetic.make_state_dict(False, 12), False)
    assert maxerr(r6, g["raw6"].reshape(-1, 6)) <= 1e-6
    assert maxerr(r4, g["raw4"].reshape(-1, 4)) <= 1e-6


@pytest.mark.parametrize("tag", ["blender_mus_noise", "blender_plain", "blender_white", "llff_white", "real360_mus",
                                 "blender_empty"])
def test_composite(golden, tag):
    g = golden("composite_" + tag)
    n = g["raw"].shape[0]
    rays = np.zeros((n, 12), np.float32)
    rays[:, 3:6] = g["rd"]
    o = O.composite(g["raw"], g["t_vals"], rays, g.get("noise"), g.get("mus"), bool(g["flags"][0]), bool(g["flags"][1]))
    for k in ("rgb_map", "disp", "acc", "weights", "depth", "rgb"):
        assert maxerr(o[k], g[k]) <= 2e-6, k
    if "cdisp" in g:
        assert maxerr(o["cdisp"], g["cdisp"]) <= 2e-6


@pytest.mark.parametrize("tag", ["c64f129", "c16f17", "c33f70", "c1f9"])
def test_samplers_indices_bit_exact(golden, tag):
    g = golden("sampler_" + tag)
    near, far, nc, ns = g["meta"]
    for pad in (1, 0):
        for det in (1, 0):
            key = "pad%d_det%d" % (pad, det)
            rnd = None if det else g["rand"]
            s, ind = O.sample_pdf_mu_sigma(g["bins"], g["weights"], g["mus"], g["sigmas"], g["part"], g["left"],
                                           g["u_dd_det"] if det else g["arange_dd"], rnd, near, far, bool(pad))
            if "ddind_" + key in g:
                assert np.array_equal(ind, g["ddind_" + key]), key  # the "bit-exact sample indices" bar
            assert maxerr(s, g["dd_" + key]) <= 1e-6, key
            assert np.all(np.diff(s, axis=1) >= 0)
            if "mip_" + key in g:
                sm = O.sample_pdf(g["bins"], g["weights"], g["u_mip_det"] if det else g["arange_mip"], rnd, bool(pad))
                assert np.array_equal(sm, g["mip_" + key]), key


@pytest.mark.parametrize("tag", ["blender_drop", "blender_full", "llff", "blender_allzero"])
def test_dp_loss(golden, tag):
    g = golden("dploss_" + tag)
    v, rows = O.dp_loss(g["t1"], g["t0"], g["w1"], g["w0"], g["mus"], g["sig"], g["left"], g["part"], bool(g["is_blender"]))
    ref = float(g["loss"])
    assert abs(v - ref) <= 1e-6 * max(abs(ref), 1e-6)
    if tag == "blender_allzero":
        assert rows == 0 and v == 0.0


@pytest.mark.parametrize("name", runiter_names())
def test_run_iter(name):
    c = load_runiter(name)
    g = c["g"]
    out = O.run_iter(g["ro"], g["rd"], g["rad"], c["sd_coarse"], c["sd_fine"], model="dd" if c["dd"] else "mip",
                     nc=c["nc"], nf=c["nf"], near=c["near"], far=c["far"], blender=c["blender"],
                     pdf_padding=c["pdf_padding"], smooth=c["smooth"], dist_reg=c["dist_reg"], t_lin=c["t_lin"],
                     t_rand=c["t_rand"], noise0=c["noise0"], u_det=c["u_det"], u_rand=c["u_rand"], noise1=c["noise1"])
    for lvl in (0, 1):
        for k in ("rgb", "depth", "disp", "acc", "weights"):
            assert maxerr(out[lvl][k], g["o%d_%s" % (lvl, k)]) <= 1e-5, (lvl, k)  # north_star bar is 1e-4
    if c["dd"]:
        assert maxerr(out[1]["t_vals"], g["s_out"]) <= 1e-5
        for k, gk in (("mus", "s_mus"), ("ssig", "s_ssig"), ("spart", "s_spart"), ("sleft", "s_sleft"),
                      ("sigmas", "d_sig0"), ("left", "d_left0"), ("part", "d_part0")):
            assert maxerr(out[0][k], g[gk]) <= 1e-6, k
        ref = float(g["o1_dp_loss"][0])
        assert abs(float(out[1]["dp_loss"]) - ref) <= 2e-5 * max(abs(ref), 1e-3)
        assert abs(out[0]["mus_reg"] - g["o0_mus_reg"][0]) <= 1e-7


@pytest.mark.parametrize("name", ["fullsize_trained_dd_blender_4096_64x128", "fullsize_trained_dd_llff_4096_64x128"])
def test_run_iter_on_trained_weights(name):
    """The oracle on the reference's own TRAINED networks (3000 iterations of its training loop, make_golden.py gen_trained: peaked
    densities, saturated colours, confident depth distributions -- not the seeded-uniform weights of every other fixture): the rays
    whose outputs the fixture stores (every 61st of 4096), per-ray quantities within 1e-5 of the reference's."""
    import numpy as np
    import torch
    from _cases import load_fullsize
    from ddnerf_amd import synthetic

    c = load_fullsize(name)
    g, st = c["g"], c["stride"]
    ro, rd, rad, _ = synthetic.make_rays(c["kind"], c["n"], 1)
    ro, rd, rad = ro[::st], rd[::st], rad[::st]
    out = O.run_iter(ro, rd, rad, c["sd_coarse"], c["sd_fine"], model="dd", nc=c["nc"], nf=c["nf"], near=c["near"], far=c["far"],
                     blender=c["kind"] == "blender", white_bkgd=False, pdf_padding=c["pdf_padding"], smooth=c["smooth"],
                     dist_reg=c["dist_reg"], t_lin=torch.linspace(0, 1, c["nc"] + 1).numpy(), t_rand=None, noise0=None,
                     u_det=torch.linspace(0.0, 0.9999, c["nf"] + 1).numpy(), u_rand=None, noise1=None, want_dp_loss=False)
    for lvl in (0, 1):
        for k in ("rgb", "depth", "acc", "weights"):
            assert maxerr(out[lvl][k], g["o%d_%s" % (lvl, k)]) <= 1e-5, (lvl, k)
    assert float(g["psnr"][1]) > 30.0 or c["kind"] != "blender"     # (the blender fixture: the reference's fit of its training scene)


def test_fp32_remainder_recipe_is_bit_exact():
    """The encode kernel replaces fmodf by a four-instruction fp32 recipe (rays_encode.hip: remainder_pos); this is the
    same arithmetic in numpy, checked bit-for-bit against torch.remainder on values that include near-multiples of T."""
    import torch

    f = np.float32
    T = f(314.159271)
    inv_up = f(0.0031830994)
    assert float(inv_up) > 1.0 / float(T) and float(inv_up) < (1.0 / float(T)) * (1 + 4 * 2.0 ** -24)
    rng = np.random.default_rng(1)
    x = np.concatenate([rng.uniform(-2 ** 21, 2 ** 21, 1_000_000),
                        rng.integers(-6000, 6000, 500_000) * np.float64(T) + rng.uniform(-1e-2, 1e-2, 500_000),
                        rng.uniform(-400, 400, 200_000)]).astype(f)
    a = np.abs(x)
    q = np.trunc((a * inv_up).astype(f))
    r64 = a.astype(np.float64) - q.astype(np.float64) * np.float64(T)      # what the fma returns before rounding
    r = r64.astype(f)
    assert np.all(r.astype(np.float64) == r64)                              # ... is exactly representable
    r = np.where(r < 0, (r + T).astype(f), r)
    r = np.copysign(r, x)
    assert np.array_equal(r, np.fmod(x, T))
    out = np.where((r != 0) & (r < 0), (r + T).astype(f), r)
    assert np.array_equal(out, torch.remainder(torch.from_numpy(x), torch.tensor(T)).numpy())
