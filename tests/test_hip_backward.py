"""GPU tests of the hand-written backward kernels: against torch autograd of plain fp32 restatements
(tests/torch_ref.py) and against the parameter gradients the reference itself produced (golden fixtures)."""
import numpy as np
import pytest
import torch

from _cases import load_runiter, runiter_names
from ddnerf_amd import synthetic
import torch_ref as R

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available()
    from ddnerf_amd import ops as _ops
    return _ops


def dev(x):
    return None if x is None else torch.from_numpy(np.ascontiguousarray(x)).cuda()


def close(a, b, rtol, atol):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    err = (a - b).abs()
    tol = atol + rtol * b.abs()
    assert bool((err <= tol).all()), "max err %.3g (|ref| max %.3g)" % (float(err.max()), float(b.abs().max()))


@pytest.mark.parametrize("white,blender,ldr,S", [(False, True, 4, 64), (True, False, 6, 128), (False, True, 6, 33)])
def test_composite_backward(ops, white, blender, ldr, S):
    g = torch.Generator().manual_seed(3)
    n = 37
    raw = (torch.randn(n, S, ldr, generator=g) * 3).cuda()
    t = torch.sort(torch.rand(n, S + 1, generator=g) * 4 + 2, dim=1)[0].cuda()
    rays = torch.zeros(n, 12).cuda()
    rays[:, 3:6] = torch.randn(n, 3, generator=g).cuda()
    noise = torch.randn(n, S, generator=g).cuda()
    G = torch.randn(n, 3, generator=g).cuda()
    GW = torch.randn(n, S, generator=g).cuda()
    raw_r = raw.clone().requires_grad_()
    rgb_map, w = R.composite(raw_r, t, rays[:, 3:6].norm(dim=-1), noise, white, blender)
    ((rgb_map * G).sum() + (w * GW).sum()).backward()
    g_raw = ops.composite_backward(raw, t, rays, noise, white, blender, G, GW)
    close(g_raw[..., :4], raw_r.grad[..., :4], 2e-4, 2e-6)
    assert float(g_raw[..., 4:].abs().max()) == 0.0 if ldr > 4 else True
    g2 = ops.composite_backward(raw, t, rays, noise, white, blender, G, None)   # no upstream on the weights
    raw_r.grad = None
    rgb_map, w = R.composite(raw_r, t, rays[:, 3:6].norm(dim=-1), noise, white, blender)
    (rgb_map * G).sum().backward()
    close(g2[..., :4], raw_r.grad[..., :4], 2e-4, 2e-6)


def test_dd_head_backward(ops):
    g = torch.Generator().manual_seed(4)
    raw6 = (torch.randn(29, 64, 6, generator=g) * 2).cuda()
    gm, gs = torch.randn(29, 64, generator=g).cuda(), torch.randn(29, 64, generator=g).cuda()
    gsc = torch.randn(4, generator=g).cuda()
    rr = raw6.clone().requires_grad_()
    mus, sig, scal = R.dd_head(rr, 0.0156)
    ((mus * gm).sum() + (sig * gs).sum() + (scal * gsc).sum()).backward()
    out = torch.zeros_like(raw6)
    ops.dd_head_backward_(raw6, 0.0156, gm, gs, gsc, out)
    close(out, rr.grad, 1e-5, 1e-7)


@pytest.mark.parametrize("tag", ["blender_drop", "blender_full", "llff"])
def test_dp_loss_backward_matches_reference_grads(ops, golden, tag):
    g = golden("dploss_" + tag)
    args = [dev(g[k]) for k in ("t1", "t0", "w1", "w0", "mus", "sig", "left", "part")]
    gw, gm, gs = ops.dp_loss_backward(*args, bool(g["is_blender"]), torch.ones((), device="cuda"))
    for mine, key in ((gw, "g_w0"), (gm, "g_mus"), (gs, "g_sig")):
        ref = torch.from_numpy(g[key])
        scale = float(ref.abs().max())
        close(mine, ref, 2e-3, 2e-5 * scale)


@pytest.mark.parametrize("depth,M", [(True, 200), (False, 129)])
def test_mlp_backward_vs_autograd(ops, depth, M):
    from ddnerf_amd import functions as F
    from ddnerf_amd import base_architectures as BA

    g = torch.Generator().manual_seed(5)
    net = (BA.DepthMipNeRFModel if depth else BA.MipNeRFModel)(hidden_size=256, include_input_dir=True)
    sd = {k: torch.from_numpy(v) for k, v in synthetic.make_state_dict(depth, 9, 3.0).items()}
    net.load_state_dict(sd)
    net.cuda()
    feat = torch.zeros(M, 128)
    feat[:, :123] = torch.rand(M, 123, generator=g) * 2 - 1
    feat = feat.cuda()
    G = torch.randn(M, 6 if depth else 4, generator=g).cuda()
    raw = F.mlp(feat, net)
    (raw * G).sum().backward()
    sdr = {k: v.clone().cuda().requires_grad_() for k, v in sd.items()}
    raw_r = R.mlp(feat, sdr, depth)
    close(raw, raw_r, 1e-5, 1e-5)
    (raw_r * G).sum().backward()
    for name, p in net.named_parameters():
        ref = sdr[name].grad
        close(p.grad, ref, 1e-3, 2e-5 * float(ref.abs().max()))
    # the parameter gradients are views of ONE flat buffer (the data-parallel bucket)
    flat = net.last_flat_grad
    assert next(net.parameters()).grad.data_ptr() == flat.data_ptr()


TRAIN_CASES = [n for n in runiter_names() if n.endswith("_train")]


@pytest.mark.parametrize("name", TRAIN_CASES)
def test_train_step_gradients_match_reference(name):
    """loss.backward() through the whole HIP path vs the parameter gradients of the reference (golden)."""
    from test_hip_run_iter import build_model

    c = load_runiter(name)
    g = c["g"]
    model = build_model(c)
    model.train()
    d = lambda x: torch.from_numpy(x).cuda()
    tgt = d(g["tgt"])
    out = model.run_iter(d(g["ro"]), d(g["rd"]), d(g["rad"]), mode="train", rgb_target=tgt)
    coef = model.cfg.train_params.loss_coeficients
    loss = sum(coef[j] * torch.nn.functional.mse_loss(out[j]["rgb"], tgt) for j in range(2))
    if c["dd"]:
        loss = loss + model.cfg.train_params.dp_coeficient * out[1]["dp_loss"].mean()
    assert abs(float(loss) - float(g["loss"])) <= 2e-5 * max(1.0, abs(float(g["loss"])))
    loss.backward()
    nets = [("c", model.coarse)] + ([("f", model.fine)] if c["dd"] else [])
    for pfx, net in nets:
        for pname, p in net.named_parameters():
            ref_sub = torch.from_numpy(g["g%s_%s_sub" % (pfx, pname)])
            ref_norm = float(g["g%s_%s_stat" % (pfx, pname)][0])
            mine = p.grad.reshape(-1)[::61].cpu()
            err = float((mine - ref_sub).abs().max())
            assert err <= 2e-3 * max(float(ref_sub.abs().max()), 1e-8) + 1e-7, (pfx, pname, err)
            assert abs(float(p.grad.double().norm()) - ref_norm) <= 2e-3 * ref_norm + 1e-9, (pfx, pname)
