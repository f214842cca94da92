"""Consistency of the round-5 fixtures the reference produced (tests/golden/make_golden.py gen_trained, gen_drift1500): no GPU, no oracle."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _load(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def test_trained_weights_come_from_the_run_whose_first_1500_iterations_are_the_curve_fixture():
    """trained_weights_dd_blender.npz: the reference's training loop carried on to 3000 iterations; its recorded curve starts with the
    train1500 fixture's records BIT FOR BIT (the same loop, deterministic on 8 ATen threads) and ends at 35 dB"""
    w, c = _load("trained_weights_dd_blender"), _load("train1500_dd_blender")
    k = len(c["it"]) - 1                      # (the 1500-step run's last record is iteration 1499, off the 25-grid)
    assert np.array_equal(w["it"][:k], c["it"][:k]) and np.array_equal(w["loss"][:k], c["loss"][:k]) and np.array_equal(w["mse"][:k], c["mse"][:k])
    assert int(w["meta"][0]) == 3000 and int(w["it"][-1]) == 2999
    assert -10 * np.log10(w["mse"][-1, 1]) > 33.0
    names = sorted(k_ for k_ in w.files if k_.startswith(("c.", "f.")))
    assert len([n for n in names if n.startswith("c.")]) == 26 and len([n for n in names if n.startswith("f.")]) == 24
    assert sum(w[n].size for n in names if n.startswith("c.")) == 612998 and sum(w[n].size for n in names if n.startswith("f.")) == 612740
    for kind in ("blender", "llff"):
        f = _load("fullsize_trained_dd_%s_4096_64x128" % kind)
        assert f["o1_rgb"].shape == (68, 3) and np.isfinite(f["o1_depth"]).all()
    assert _load("fullsize_trained_dd_blender_4096_64x128")["psnr"][1] > 35.0      # the reference's fit of its own training scene


def test_reference_self_drift_fixture():
    """train1500_drift_dd_blender.npz: the reference against itself under two perturbations of round-off size; it really drifts (the
    bars of test_hip_baseline_size.py's 1500-iteration test are 1.25 x these distances) and it still learns the same scene"""
    g, d = _load("train1500_dd_blender"), _load("train1500_drift_dd_blender")
    assert np.array_equal(d["it"], g["it"])
    ps = lambda m: -10 * np.log10(m)
    for tag in ("ulp", "thr4"):
        dp = np.abs(ps(d["mse_" + tag]) - ps(g["mse"]))
        assert 0.3 < dp.max() < 1.0 and 0.03 < dp.mean() < 0.15, (tag, dp.max(), dp.mean())
        assert d["loss_" + tag][0] == g["loss"][0] or abs(d["loss_" + tag][0] - g["loss"][0]) < 1e-6      # iteration 0: before any drift
        assert abs(ps(d["mse_" + tag][-1, 1]) - ps(g["mse"][-1, 1])) < 0.5
