"""The two north_star bars pinned AT BASELINE SIZE against the reference itself (fixtures: tests/golden/make_golden.py
gen_sampler4096 / gen_grad4096 / gen_train300):
  * sampler boundary: the reference's own cfg2 coarse pass (4096 rays x 64 bins) in, bit-exact bin indices on all 4096 x 129
    samples (models/samplers.py:124-215);
  * parameter gradients of the whole 4096-ray x (64 + 128) training pass, with the dp term on and off;
  * training parity ("PSNR vs ref (2)"): 300 iterations of the reference's loop on a procedural scene, the HIP
    path's loss / PSNR curve against the reference's (train_model.py:132-177)."""
import os

import numpy as np
import pytest
import torch

from _cases import GOLDEN
from ddnerf_amd import synthetic

pytestmark = pytest.mark.gpu


def _load(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz")))


@pytest.mark.parametrize("name", ["sampler4096_cfg2", "sampler4096_trained"])
def test_sampler_indices_bit_exact_at_4096_rays(name):
    """the sampler boundary of a 4096-ray coarse pass, exactly as the reference handed it over: seeded-uniform weights (cfg2) and the
    TRAINED coarse network of the reference's 3000-iteration run (peaked weights, mostly empty rays)"""
    from ddnerf_amd import ops

    g = _load(name)
    n, nc, ns, near, far, pad = g["meta"]
    n, nc, ns = int(n), int(nc), int(ns)
    d = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
    bins = d(np.broadcast_to(g["bins_row"], (n, nc + 1)))
    u_det = torch.linspace(0.0, 0.9999, ns).cuda()                       # models/samplers.py:156 (det)
    out, ind = ops.sample_pdf_mu_sigma(bins, d(g["weights"]), d(g["mus"]), d(g["ssig"]), d(g["spart"]), d(g["sleft"]), u_det, None,
                                       float(near), float(far), bool(pad), want_ind=True)
    ind = ind.cpu().numpy()
    ref_ind = g["bins_ind"].astype(np.int32)
    assert ind.shape == ref_ind.shape == (n, ns)
    assert np.array_equal(ind, ref_ind), "%d of %d sample indices differ" % (int((ind != ref_ind).sum()), ind.size)
    # every bin is used somewhere in the batch, rows differ from each other: the comparison is not vacuous
    assert len(np.unique(ref_ind)) == nc and len(np.unique(ref_ind, axis=0)) > n // (2 if name.endswith("cfg2") else 8)
    s, ref = out.cpu().numpy(), g["samples"]
    assert np.abs(s - ref).max() <= 2e-6 * far                            # erf / erfinv / exp implementations differ by ulps
    assert np.all(np.diff(s, axis=1) >= 0)


def _cfg2_train_model(sharpen, dp_coef, mlp_dtype, meta):
    from test_hip_run_iter import build_model

    n, nc, nf, _sh, noise, near, far, dist_reg, smooth, pad = (float(v) for v in meta[:10])
    c = dict(g={}, dd=True, kind="blender", nc=int(nc), nf=int(nf), noise=0.0, near=near, far=far, dist_reg=dist_reg, smooth=smooth,
             pdf_padding=bool(pad), train=False, sd_coarse=synthetic.make_state_dict(True, 11, sharpen),
             sd_fine=synthetic.make_state_dict(False, 12, sharpen), dp_coef=dp_coef)
    model = build_model(c)
    for mode in ("train", "validation"):
        model.cfg.nerf[mode]["perturb"] = False
        model.cfg.nerf[mode]["radiance_field_noise_std"] = 0.0
    model.cfg.nerf["mlp_dtype"] = mlp_dtype
    model._set_mlp_dtype()
    return model


@pytest.mark.parametrize("mlp_dtype", ["fp32", "x3"])
@pytest.mark.parametrize("tag", ["cfg2_dp0", "cfg2_dp1", "trained_dp0", "trained_dp1"])
def test_gradients_at_4096_rays(tag, mlp_dtype):
    """loss.backward() of the whole cfg2-size training pass vs the reference's parameter gradients (every 61st entry + norms); "trained":
    at the state the reference's own 3000-iteration run reached (make_golden.py gen_grad4096_trained), on the targets it was trained on"""
    g = _load("grad4096_" + tag)
    n = int(g["meta"][0])
    dp_coef = float(g["meta"][10])
    model = _cfg2_train_model(float(g["meta"][3]), dp_coef, mlp_dtype, g["meta"])
    trained = tag.startswith("trained")
    if trained:
        from _cases import trained_state_dicts

        sd_c, sd_f = trained_state_dicts()
        model.coarse.load_state_dict({k: torch.from_numpy(v) for k, v in sd_c.items()})
        model.fine.load_state_dict({k: torch.from_numpy(v) for k, v in sd_f.items()})
    model.cfg.train_params.dp_coeficient = dp_coef
    model.train()
    ro, rd, rad, tgt = (torch.from_numpy(x).cuda() for x in synthetic.make_rays("blender", n, 6))
    if trained:
        o_, d_, _, _ = synthetic.make_rays("blender", n, 6)
        tgt = torch.from_numpy(synthetic.procedural_targets(o_, d_)).cuda()
    out = model.run_iter(ro, rd, rad, mode="train", rgb_target=tgt)
    coef = model.cfg.train_params.loss_coeficients
    mses = [torch.nn.functional.mse_loss(out[j]["rgb"], tgt) for j in range(2)]
    loss = sum(coef[j] * mses[j] for j in range(2)) + dp_coef * out[1]["dp_loss"].mean()
    assert np.allclose([float(m) for m in mses], g["mse"], rtol=2e-5, atol=0)
    assert abs(float(out[1]["dp_loss"][0]) - float(g["dp_loss"][0])) <= 2e-4 * max(abs(float(g["dp_loss"][0])), 1e-2)
    assert abs(float(loss) - float(g["loss"])) <= 2e-5 * max(1.0, abs(float(g["loss"])))
    loss.backward()
    for pfx, net in (("c", model.coarse), ("f", model.fine)):
        # (the dp-loss gradient into the coarse net is ill-conditioned in the reference itself: tests/test_hip_backward.py)
        chaotic = pfx == "c" and dp_coef != 0.0
        # (trained state: the MSE gradients are tiny there -- loss 6.6e-4 -- so the coarse net's gradient IS the dp term's ill-conditioned
        # one: measured 0.43 of a parameter's norm, 0.2 - 0.3 of the whole net's; the dp0 fixture holds the same backward path to 1e-2)
        tol = (0.7 if trained else 0.25) if chaotic else 1e-2
        mine_all, ref_all = [], []
        for pname, p in net.named_parameters():
            ref_sub = torch.from_numpy(g["g%s_%s_sub" % (pfx, pname)]).double()
            ref_norm = float(g["g%s_%s_stat" % (pfx, pname)][0])
            mine = p.grad.reshape(-1)[::61].cpu().double()
            mine_all.append(mine)
            ref_all.append(ref_sub)
            if chaotic and trained:
                continue          # (per parameter the sampled entries of this gradient are noise against noise: only the whole net is held, below)
            assert float((mine - ref_sub).norm()) <= tol * float(ref_sub.norm()) + 1e-12, (pfx, pname)
            assert abs(float(p.grad.double().norm()) - ref_norm) <= tol * ref_norm + 1e-9, (pfx, pname)
        a, b = torch.cat(mine_all), torch.cat(ref_all)
        assert float((a - b).norm()) <= ((0.5 if trained else 3e-2) if chaotic else 5e-3) * float(b.norm()), (pfx, float((a - b).norm() / b.norm()))


def _psnr(m):
    return -10.0 * np.log10(np.maximum(m, 1e-12))


def _training_curve(name, mlp_dtype, monkeypatch):
    """runs the fixture's schedule on the HIP path; returns (recorded iterations, losses, MSEs) and the fixture"""
    from ddnerf_amd import train_step
    from test_hip_run_iter import build_model

    if mlp_dtype == "x3-exact":        # the x3 tier with exact hi/lo-word records (DDNERF_X3_WGRAD=exact)
        monkeypatch.setenv("DDNERF_X3_WGRAD", "exact")
        mlp_dtype = "x3"
    else:
        monkeypatch.delenv("DDNERF_X3_WGRAD", raising=False)
    g = _load(name)
    meta = [float(v) for v in g["meta"]]
    n, nc, nf, iters, delay, near, far = meta[:7]
    n, nc, nf, iters = int(n), int(nc), int(nf), int(iters)
    max_steps = int(meta[7]) if len(meta) > 7 else iters     # (train1500: the first 1500 steps of a run of the config's full length)
    every = int(g["it"][1] - g["it"][0])
    dd = "_dd_" in name
    c = dict(g={}, dd=dd, kind="blender", nc=nc, nf=nf, noise=0.0, near=near, far=far, dist_reg=0.0, smooth=1.7, pdf_padding=True,
             train=False, sd_coarse=synthetic.make_state_dict(dd, 11, 1.0), sd_fine=synthetic.make_state_dict(False, 12, 1.0) if dd else None)
    model = build_model(c)
    cfg = model.cfg
    for mode in ("train", "validation"):
        cfg.nerf[mode]["perturb"] = False
        cfg.nerf[mode]["radiance_field_noise_std"] = 0.0
    cfg.nerf["mlp_dtype"] = mlp_dtype
    model._set_mlp_dtype()
    # the shipped config's own schedules (smoothing 1.7 -> final_smooth over finnish_smooth, automatic dist_reg), as in the fixture
    from ddnerf_amd.cfgnode import CfgNode

    ref_cfg = CfgNode.load(os.path.join(os.path.dirname(GOLDEN), "..", "configs", "config_blender.yml" if dd else "config_blender_mipnerf.yml"))
    for k in ("gaussian_smooth_factor", "final_smooth", "finnish_smooth", "set_automatic_dist_reg_coeficient", "dist_reg_coeficient",
              "max_pdf_pad_iters", "pdf_padding", "dp_coeficient", "loss_coeficients"):
        setattr(cfg.train_params, k, getattr(ref_cfg.train_params, k))
    cfg.experiment.train_iters = max_steps
    cfg["scheduler"] = {"lr_init": 0.0005, "lr_final": 5e-6, "lr_delay_steps": int(delay), "lr_delay_mult": 0.01}
    stepper = train_step.TrainStepper(model, cfg)
    got_loss, got_mse = [], []
    for i in range(iters):
        ro, rd, rad, _ = synthetic.make_rays("blender", n, 5000 + i)
        tgt = synthetic.procedural_targets(ro, rd)
        loss, parts, _ = stepper.step(*(torch.from_numpy(x).cuda() for x in (ro, rd, rad, tgt)))
        if i % every == 0 or i == iters - 1:
            got_loss.append(float(loss))
            got_mse.append([float(p) for p in parts[:2]])
    assert list(g["it"]) == [i for i in range(iters) if i % every == 0 or i == iters - 1]
    return np.array(got_loss), np.array(got_mse), g


@pytest.mark.parametrize("mlp_dtype", ["fp32", "x3", "x3-exact"])
@pytest.mark.parametrize("name", ["train300_dd_blender", "train300_mip_blender"])
def test_training_curve_tracks_the_reference(name, mlp_dtype, monkeypatch):
    """300 iterations (fresh 256-ray batch per iteration, analytic targets, perturb / noise off, 50-step lr warm-up): the HIP
    path's loss and PSNR, every 10 iterations, against the reference loop's.  Measured in round 4 (tools/train_curve_stats.py): every tier
    within 0.002 dB of the reference at every recorded iteration, losses within 0.05 % -- held to 0.05 dB / 0.02 dB (mean of the last
    50 iterations) / 0.5 %."""
    got_loss, got_mse, g = _training_curve(name, mlp_dtype, monkeypatch)
    ref_loss, ref_mse = g["loss"], g["mse"]
    assert ref_mse[-1, 1] < 0.35 * ref_mse[0, 1]                       # the reference run really learns (about 5 dB in 300 steps)
    assert abs(got_loss[0] - ref_loss[0]) <= 2e-5 * max(1.0, abs(ref_loss[0]))   # iteration 0: plain forward parity
    d_psnr = np.abs(_psnr(got_mse) - _psnr(ref_mse))
    assert d_psnr.max() <= 0.05, (d_psnr.max(), int(d_psnr.argmax()))
    assert abs(_psnr(got_mse[-6:, 1]).mean() - _psnr(ref_mse[-6:, 1]).mean()) <= 0.02
    assert np.all(np.abs(got_loss - ref_loss) <= 0.005 * np.abs(ref_loss) + 1e-5)


def _reference_self_distance():
    """How far the REFERENCE drifts from ITSELF over the 1500 iterations of train1500_dd_blender under a perturbation of fp32 round-off
    size (tests/golden/train1500_drift_dd_blender.npz, make_golden.py gen_drift1500: every initial weight moved one ulp; the same
    arithmetic on 4 ATen threads instead of 8), in the metrics of the test below: the larger of the two perturbations' distances.
    Round 5: max |d PSNR| 0.60 / 0.63 dB (both at record 43 = iteration 1075, where the HIP tiers' maxima sit too), mean 0.087 / 0.088,
    mean of the last six records 0.004 / 0.046, relative loss 6.0 / 5.3 %."""
    g, d = _load("train1500_dd_blender"), _load("train1500_drift_dd_blender")
    out = dict(max=0.0, mean=0.0, last6=0.0, loss=0.0)
    for tag in ("ulp", "thr4"):
        dp = np.abs(_psnr(d["mse_" + tag]) - _psnr(g["mse"]))
        out["max"] = max(out["max"], float(dp.max()))
        out["mean"] = max(out["mean"], float(dp.mean()))
        out["last6"] = max(out["last6"], float(abs(_psnr(d["mse_" + tag][-6:, 1]).mean() - _psnr(g["mse"][-6:, 1]).mean())))
        out["loss"] = max(out["loss"], float(np.max(np.abs(d["loss_" + tag] - g["loss"]) / np.abs(g["loss"]))))
    return out


@pytest.mark.parametrize("mlp_dtype", ["fp32", "x3", "x3-exact"])
def test_training_curve_1500_iterations_of_the_real_schedule(mlp_dtype, monkeypatch):
    """The FIRST 1500 steps of a real run: the schedule train_model.py hard-wires (:101-107 -- 5e-4 -> 5e-6 over the config's
    train_iters with the 2500-step x0.01 warm-up), DDNerfModel, 256 fresh rays per iteration, loss / PSNR every 25 iterations against
    the reference loop's (tests/golden/train1500_dd_blender.npz, make_golden.py gen_train1500): PSNR 11.9 -> 29.8 dB.  Two fp32-class
    implementations of one chaotic optimisation drift apart slowly -- and so does the reference from itself: the bars are 1.25 x the
    REFERENCE-vs-REFERENCE distance under a perturbation of round-off size (_reference_self_distance: 0.79 dB max / 0.11 dB mean /
    0.058 dB on the last six records / 7.5 % of the loss), not chosen.  Measured (tools/train_curve_stats.py): max |d PSNR| 0.47 dB
    (fp32 tier: LESS than the reference's own 0.60 - 0.63) / 0.66 (x3) / 0.72 (x3, exact records), all at iteration 1075 like the
    reference's; mean 0.07 - 0.09 dB; last six records within 0.009 dB; losses within 4 - 6 %."""
    got_loss, got_mse, g = _training_curve("train1500_dd_blender", mlp_dtype, monkeypatch)
    ref_loss, ref_mse = g["loss"], g["mse"]
    bar = {k: 1.25 * v for k, v in _reference_self_distance().items()}
    assert 0.5 <= bar["max"] <= 1.0 and 0.05 <= bar["mean"] <= 0.15, bar     # (the calibration itself: a fixture that no longer drifts would make the bars vacuous)
    assert ref_mse[-1, 1] < 0.05 * ref_mse[0, 1]                       # the reference run learns: 18 dB in these 1500 steps
    assert abs(got_loss[0] - ref_loss[0]) <= 2e-5 * max(1.0, abs(ref_loss[0]))
    d_psnr = np.abs(_psnr(got_mse) - _psnr(ref_mse))
    assert d_psnr.max() <= bar["max"], (d_psnr.max(), int(d_psnr.argmax()), bar)
    assert d_psnr.mean() <= bar["mean"], (d_psnr.mean(), bar)
    assert abs(_psnr(got_mse[-6:, 1]).mean() - _psnr(ref_mse[-6:, 1]).mean()) <= bar["last6"], bar
    assert np.all(np.abs(got_loss - ref_loss) <= bar["loss"] * np.abs(ref_loss) + 1e-5), (float(np.max(np.abs(got_loss - ref_loss) / np.abs(ref_loss))), bar)
