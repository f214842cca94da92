"""The render path's folded launches against the one-kernel-per-reference-function entry points they replace: same arithmetic, same
outputs (bit for bit, except the two L2 regularisers at nc != 64, whose partial sums are grouped differently).
  ops.pack_rays_first_cycle  = pack_rays + sample_first_cycle                      (models/models.py:144-162, models/samplers.py:30-62)
  ops.dd_coarse_forward      = dd_head + composite_forward + dd_records            (models/models.py:242-295)
  ops.dd_coarse_forward(sample=...) = ... + sample_pdf_mu_sigma in the same launch  (models/models.py:227-237, models/samplers.py:124-215)
  ops.composite_forward_keep + dp_loss_forward_kept = composite_forward + dp_loss_forward   (models/dd_utils.py:16)"""
import numpy as np
import pytest
import torch

from ddnerf_amd import synthetic

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available()
    from ddnerf_amd import ops as _ops
    return _ops


def _rays(kind, n, seed):
    ro, rd, rad, _ = synthetic.make_rays(kind, n, seed)
    return tuple(torch.from_numpy(x).cuda() for x in (ro, rd, rad))


@pytest.mark.parametrize("lindisp", [False, True])
@pytest.mark.parametrize("n,nc", [(1, 4), (37, 64), (4096, 64), (300, 33)])
def test_pack_rays_first_cycle(ops, n, nc, lindisp):
    ro, rd, rad = _rays("real360" if lindisp else "blender", n, 3)
    near, far = (0.2, 2.8) if lindisp else (2.0, 6.0)
    t_lin = torch.linspace(0.0, 1.0, nc + 1).cuda()
    for t_rand in (None, torch.rand(n, nc + 1, device="cuda")):
        rays_a = ops.pack_rays(ro, rd, rad, near, far)
        t_a = ops.sample_first_cycle(rays_a, t_lin, t_rand, lindisp)
        rays_b, t_b = ops.pack_rays_first_cycle(ro, rd, rad, near, far, t_lin, t_rand, lindisp)
        assert torch.equal(rays_a, rays_b) and torch.equal(t_a, t_b)


@pytest.mark.parametrize("kind", ["fp32", "bf16", "fp16"])
@pytest.mark.parametrize("lindisp,cyl", [(False, False), (True, False), (False, True)])
@pytest.mark.parametrize("n,nc", [(1, 4), (37, 64), (4096, 64), (300, 33), (5, 128)])
def test_encode_first_cycle(ops, n, nc, lindisp, cyl, kind):
    """pack + first cycle + coarse encode in one launch: rays, fenceposts and features bit for bit those of the separate launches (a block
    of 32 samples straddles rays when nc is not a multiple of 32; the last fencepost of a ray is written by its last sample's thread)"""
    ro, rd, rad = _rays("real360" if lindisp else "blender", n, 5)
    near, far = (0.2, 2.8) if lindisp else (2.0, 6.0)
    t_lin = torch.linspace(0.0, 1.0, nc + 1).cuda()
    rays_a, t_a = ops.pack_rays_first_cycle(ro, rd, rad, near, far, t_lin, None, lindisp)
    feat_a = ops.encode(rays_a, t_a, cylinder=cyl, kind=kind)
    rays_b, t_b, feat_b = ops.encode_first_cycle(ro, rd, rad, near, far, t_lin, lindisp, cylinder=cyl, kind=kind)
    assert torch.equal(rays_a, rays_b) and torch.equal(t_a, t_b)
    assert feat_a.dtype == feat_b.dtype and torch.equal(feat_a.view(torch.int16 if kind != "fp32" else torch.int32),
                                                        feat_b.view(torch.int16 if kind != "fp32" else torch.int32))


def _coarse_inputs(n, nc, seed, kind="blender", zero_rows=False):
    g = torch.Generator(device="cuda").manual_seed(seed)
    ro, rd, rad = _rays(kind, n, seed)
    near, far = synthetic.NEAR_FAR[kind]
    from ddnerf_amd import ops
    rays = ops.pack_rays(ro, rd, rad, near, far)
    t = torch.sort(torch.rand(n, nc + 1, device="cuda", generator=g) * (far - near) + near, dim=1)[0]
    t[:, 0], t[:, -1] = near, far
    raw = torch.randn(n, nc, 6, device="cuda", generator=g) * 3.0
    if zero_rows:
        raw[::5, :, 3] = -60.0          # softplus underflows: all-zero weight rows (0/0 pdf, dropped by the dp loss's filter)
    return rays, t.contiguous(), raw.contiguous()


@pytest.mark.parametrize("white,blender,with_noise", [(False, True, True), (True, True, False), (False, False, False), (True, False, True)])
@pytest.mark.parametrize("n,nc", [(4096, 64), (37, 64), (50, 16), (9, 33), (130, 128)])
def test_dd_coarse_forward_equals_its_parts(ops, n, nc, white, blender, with_noise):
    rays, t, raw = _coarse_inputs(n, nc, 7 + n, zero_rows=True)
    noise = torch.randn(n, nc, device="cuda") if with_noise else None
    smooth, dist_reg = 1.7, 0.0156
    head = ops.dd_head(raw, smooth, dist_reg)
    c = ops.composite_forward(raw, t, rays, noise, head["mus"], white, blender)
    rec = ops.dd_records(c["weights"], head["mus"], head["sigmas"], head["ssig"])
    c2, head2, ticket = ops.dd_coarse_forward(raw, t, rays, noise, smooth, dist_reg, white, blender)
    rec2 = ops.dd_records_finish(ticket)
    for k in ("mus", "sigmas", "left", "part", "ssig", "sleft", "spart"):
        assert torch.equal(head[k], head2[k]), k
    if nc == 64:
        assert torch.equal(head["scal"], head2["scal"])
    else:
        assert torch.allclose(head["scal"], head2["scal"], rtol=2e-6, atol=0)
    for k in ("rgb_map", "disp", "acc", "weights", "depth", "cdisp"):
        a, b = c[k], c2[k]
        assert torch.equal(torch.isnan(a), torch.isnan(b)) and torch.equal(torch.nan_to_num(a), torch.nan_to_num(b)), k
    assert len(rec) == len(rec2) == 3
    for a, b in zip(rec, rec2):
        assert a.shape == b.shape and torch.equal(a, b)


@pytest.mark.parametrize("perturb,pad", [(False, True), (True, True), (True, False)])
@pytest.mark.parametrize("n,nc,nf", [(4096, 64, 128), (37, 64, 128), (50, 16, 16), (9, 33, 40), (130, 128, 64), (5, 1, 8)])
def test_dd_coarse_forward_with_the_sampler_folded_in(ops, n, nc, nf, perturb, pad):
    """the coarse launch that also draws the fine fenceposts: every output of the launch without the sampler, and samples bit for bit
    those of the stand-alone sampler kernel on that launch's weights / mus / smoothed head values (rows with all-zero weights, the
    nc = 1 special case of models/samplers.py:185-190 and row lengths off the sort network's power-of-two sizes included)"""
    rays, t, raw = _coarse_inputs(n, nc, 11 + n, zero_rows=True)
    noise = torch.randn(n, nc, device="cuda")
    smooth, dist_reg, near, far = 1.7, 0.0156, 2.0, 6.0
    ns = nf + 1
    if perturb:
        u_base = (torch.arange(ns) * (1 / (ns - 1))).float().cuda()
        rnd = torch.rand(n, ns, device="cuda")
    else:
        u_base, rnd = torch.linspace(0.0, 0.9999, ns).cuda(), None
    c, head, ticket = ops.dd_coarse_forward(raw, t, rays, noise, smooth, dist_reg, False, True)
    rec = ops.dd_records_finish(ticket)
    want = ops.sample_pdf_mu_sigma(t, c["weights"], head["mus"], head["ssig"], head["spart"], head["sleft"], u_base, rnd, near, far, pad)
    c2, head2, ticket2, got = ops.dd_coarse_forward(raw, t, rays, noise, smooth, dist_reg, False, True, sample=(u_base, rnd, near, far, pad))
    rec2 = ops.dd_records_finish(ticket2)
    eq = lambda a, b: torch.equal(torch.isnan(a), torch.isnan(b)) and torch.equal(torch.nan_to_num(a), torch.nan_to_num(b))
    assert got.shape == want.shape == (n, ns) and eq(got, want)
    for k in head:
        assert eq(head[k], head2[k]), k
    for k in ("rgb_map", "disp", "acc", "weights", "depth", "cdisp"):
        assert eq(c[k], c2[k]), k
    for a, b in zip(rec, rec2):
        assert torch.equal(a, b)


@pytest.mark.parametrize("dp_blender", [True, False])
@pytest.mark.parametrize("n,nc,nf", [(4096, 64, 128), (41, 16, 16), (10, 32, 48)])
def test_fine_compositing_with_row_filter(ops, n, nc, nf, dp_blender):
    rays, t0, raw0 = _coarse_inputs(n, nc, 3 + n)
    _, t1, raw1 = _coarse_inputs(n, nf, 5 + n, zero_rows=True)
    raw1 = raw1[..., :4].contiguous()
    head = ops.dd_head(raw0, 1.7, 0.0156)
    c0 = ops.composite_forward(raw0, t0, rays, None, head["mus"], False, True)
    for blender in (True, False):
        c1 = ops.composite_forward(raw1, t1, rays, None, None, False, blender)
        args = (t1, t0, c1["weights"], c0["weights"], head["mus"], head["sigmas"], head["left"], head["part"])
        loss_a, total_a = ops.dp_loss_forward(*args, dp_blender, reg_scal=head["scal"])
        c1b, ws = ops.composite_forward_keep(raw1, t1, rays, None, None, False, blender, dp_blender)
        loss_b, total_b = ops.dp_loss_forward_kept(*args[:2], c1b["weights"], *args[3:], ws, head["scal"])    # rows kernel + finish kernel
        for k in ("rgb_map", "disp", "acc", "weights", "depth"):
            a, b = c1[k], c1b[k]
            assert torch.equal(torch.isnan(a), torch.isnan(b)) and torch.equal(torch.nan_to_num(a), torch.nan_to_num(b)), k
        assert torch.equal(loss_a, loss_b) and torch.equal(total_a, total_b)
        keep = ws[: 4 * n].view(torch.int32)
        want = (c1["weights"].sum(1) > 1e-10).int() if dp_blender else torch.ones(n, dtype=torch.int32, device="cuda")
        assert int((keep != want).sum()) == 0 and (not dp_blender or int(keep.sum()) < n)      # the filter really drops rows


@pytest.mark.parametrize("name", ["runiter_dd_blender_64x128_validation", "runiter_dd_llff_16x16_validation", "runiter_dd_real360_32x48_validation",
                                  "runiter_mip_blender_64x128_validation"])
def test_run_iter_same_outputs_with_and_without_folding(name):
    """whole render pass, replayed random draws: the folded path (default) and the one-kernel-per-function path give identical dicts"""
    from _cases import load_runiter
    from ddnerf_amd import models as M
    from test_hip_run_iter import build_model

    c = load_runiter(name)
    g = c["g"]
    d = lambda x: torch.from_numpy(x).cuda()
    outs = []
    for fuse in (True, False):
        M.FUSE_RENDER = fuse
        try:
            model = build_model(c)
            model.eval()
            with torch.no_grad():
                outs.append(model.run_iter(d(g["ro"]), d(g["rd"]), d(g["rad"]), mode="validation", rgb_target=d(g["tgt"])))
        finally:
            M.FUSE_RENDER = True
    a, b = outs
    assert a.keys() == b.keys()
    for lvl in a:
        assert a[lvl].keys() == b[lvl].keys()
        for k in a[lvl]:
            x, y = a[lvl][k], b[lvl][k]
            assert (x is None) == (y is None), (lvl, k)
            if x is not None:
                assert x.shape == y.shape, (lvl, k)
                if k in ("mus_loss", "sig_loss", "mus_reg", "sig_reg", "dp_loss") and c["nc"] != 64:
                    # (the two L2 regularisers' partial sums are grouped per 4 rays instead of per 256 elements: same at nc = 64)
                    assert torch.allclose(x, y, rtol=3e-6, atol=0), (lvl, k)
                else:
                    assert torch.equal(torch.nan_to_num(x), torch.nan_to_num(y)), (lvl, k)


def test_kernel_drawn_noise_is_the_materialised_noise(ops):
    """the density noise the compositing kernels draw themselves (Philox4x32-10 + Box-Muller keyed by torch's generator state): compositing
    with it is BIT-IDENTICAL to compositing with the tensor ddnerf_debug_philox_normal writes for the same (seed, offset, base, std) --
    coarse launch (with the sampler folded in) and fine launch"""
    n, nc, nf = 513, 64, 128
    rays, t0, raw0 = _coarse_inputs(n, nc, 21)
    _, t1, raw1 = _coarse_inputs(n, nf, 22)
    raw1 = raw1[..., :4].contiguous()
    kn = ops.KernelNoise(seed=1234567, offset=40, base=0, std=0.7)
    u_base = torch.linspace(0.0, 0.9999, nf + 1).cuda()
    sample = (u_base, None, 2.0, 6.0, True)
    a = ops.dd_coarse_forward(raw0, t0, rays, kn.at(0), 1.7, 0.0156, False, True, sample=sample)
    b = ops.dd_coarse_forward(raw0, t0, rays, kn.at(0).materialise(n * nc, raw0.device).view(n, nc), 1.7, 0.0156, False, True, sample=sample)
    for k in ("rgb_map", "disp", "acc", "weights", "depth", "cdisp"):
        assert torch.equal(a[0][k], b[0][k]), k
    assert torch.equal(a[3], b[3])
    ops.dd_records_finish(a[2])
    ops.dd_records_finish(b[2])
    c, _ = ops.composite_forward_keep(raw1, t1, rays, kn.at(n * nc), None, False, True, True)
    d, _ = ops.composite_forward_keep(raw1, t1, rays, kn.at(n * nc).materialise(n * nf, raw1.device).view(n, nf), None, False, True, True)
    for k in ("rgb_map", "disp", "acc", "weights", "depth"):
        assert torch.equal(c[k], d[k]), k
    e, _ = ops.composite_forward_keep(raw1, t1, rays, None, None, False, True, True)
    assert not torch.equal(c["weights"], e["weights"])                      # (the noise really enters)


def test_kernel_drawn_noise_is_standard_normal(ops):
    """2^22 values: moments of N(0, std^2), a Kolmogorov-Smirnov distance of a sample that size, no repeats between streams"""
    N, std = 1 << 22, 1.0
    kn = ops.KernelNoise(seed=42, offset=0, base=0, std=std)
    x = kn.materialise(N, torch.device("cuda")).double()
    assert torch.isfinite(x).all()
    assert abs(float(x.mean())) < 5 * std / N ** 0.5 and abs(float(x.std()) - std) < 3e-3
    z = x / std
    assert abs(float((z ** 3).mean())) < 0.01 and abs(float((z ** 4).mean()) - 3.0) < 0.03
    xs = torch.sort(z)[0]
    cdf = 0.5 * (1 + torch.erf(xs / 2 ** 0.5))
    emp = (torch.arange(1, N + 1, device="cuda", dtype=torch.float64)) / N
    assert float((cdf - emp).abs().max()) < 2.5 / N ** 0.5                    # (1.63 / sqrt(N) is the 1 % point)
    assert float(x.abs().max()) > 4.5                                           # the tails are there (Box-Muller on 32-bit uniforms reaches 6.6)
    # another offset, another seed, another base: other values; the same arguments: the same values
    assert torch.equal(x.float(), kn.materialise(N, torch.device("cuda")).double().float())
    for other in (ops.KernelNoise(42, 4, 0, std), ops.KernelNoise(43, 0, 0, std), kn.at(N)):
        y = other.materialise(4096, torch.device("cuda")).double()
        assert float((y == x[:4096]).double().mean()) < 0.01
        assert abs(float(torch.corrcoef(torch.stack([y, x[:4096]]))[0, 1])) < 0.08
    assert torch.equal(kn.at(100).materialise(50, torch.device("cuda")), x[100:150].float())   # base = an index into ONE stream


def test_render_with_kernel_noise_follows_torch_seeding():
    """run_iter(mode="validation") with the shipped noise std 1.0: torch.manual_seed reproduces the render, two renders in a row differ
    (the generator's offset advances), and DDNERF_KERNEL_NOISE=0 (a torch generator launch again) gives a statistically equal image"""
    from _cases import load_runiter
    from ddnerf_amd import models as M
    from test_hip_run_iter import build_model

    c = load_runiter("runiter_dd_blender_64x128_validation")
    g = c["g"]
    d = lambda x: torch.from_numpy(x).cuda()
    model = build_model(c)
    model.rng = M.TorchRng()
    for mode in ("train", "validation"):
        model.cfg.nerf[mode]["radiance_field_noise_std"] = 1.0
    model.eval()
    outs = []
    for seed in (7, 7, 8):
        torch.manual_seed(seed)
        with torch.no_grad():
            outs.append(model.run_iter(d(g["ro"]), d(g["rd"]), d(g["rad"]), mode="validation", rgb_target=d(g["tgt"])))
            again = model.run_iter(d(g["ro"]), d(g["rd"]), d(g["rad"]), mode="validation", rgb_target=d(g["tgt"]))
        assert not torch.equal(outs[-1][1]["rgb"], again[1]["rgb"])
    assert torch.equal(outs[0][1]["rgb"], outs[1][1]["rgb"]) and torch.equal(outs[0][0]["weights"], outs[1][0]["weights"])
    assert not torch.equal(outs[0][1]["rgb"], outs[2][1]["rgb"])
    # the same render with a torch generator launch for the noise (DDNERF_KERNEL_NOISE=0): another draw of the same distribution --
    # the image means over 16 seeds of either source agree within four standard errors
    def image_means(kernel_noise):
        M.KERNEL_NOISE = kernel_noise
        try:
            vals = []
            for seed in range(100, 116):
                torch.manual_seed(seed)
                with torch.no_grad():
                    o = model.run_iter(d(g["ro"]), d(g["rd"]), d(g["rad"]), mode="validation", rgb_target=d(g["tgt"]))
                vals.append(float(o[1]["rgb"].double().mean()))
            return np.array(vals)
        finally:
            M.KERNEL_NOISE = True

    a, b = image_means(True), image_means(False)
    assert a.std() > 0 and b.std() > 0
    assert abs(a.mean() - b.mean()) <= 4.0 * np.sqrt(a.var(ddof=1) / len(a) + b.var(ddof=1) / len(b)) + 1e-6, (a.mean(), b.mean(), a.std(), b.std())


def test_deferred_ray_rows_are_filled_whatever_happens_in_predict(monkeypatch):
    """run_iter hands predict ray rows that the coarse encoder's launch fills (GeneralMipNerfModel._rays_batches, defer): if predict raises
    before that launch -- or reads the rows first -- run_iter's `finally` fills them; the public get_rays_batches always returns filled
    rows (round-4 review, weak 13)."""
    from _cases import load_runiter
    from test_hip_run_iter import build_model

    c = load_runiter("runiter_dd_blender_64x128_validation")
    g = c["g"]
    d = lambda x: torch.from_numpy(x).cuda()
    model = build_model(c)
    model.eval()
    ro, rd, rad = d(g["ro"]), d(g["rd"]), d(g["rad"])
    want = model.get_rays_batches(ro, rd, rad, "validation")          # the public call: filled
    assert len(want) == 1 and bool(torch.isfinite(want[0]).all()) and float(want[0][:, 9:12].norm(dim=1).sub(1).abs().max()) < 1e-5
    seen = {}

    def boom(self, ray_batch, mode, depth_analysis_validation, rgb_target=None):
        seen["rows"] = ray_batch                                      # (handed out empty: nothing has been launched on them yet)
        raise RuntimeError("predict failed before its first launch")

    monkeypatch.setattr(type(model), "predict", boom)
    with pytest.raises(RuntimeError, match="predict failed"):
        with torch.no_grad():
            model.run_iter(ro, rd, rad, mode="validation", rgb_target=d(g["tgt"]))
    torch.cuda.synchronize()
    assert torch.equal(seen["rows"], want[0])                          # filled by the `finally`, bit for bit the public call's rows
    assert getattr(model, "_first_pending", None) is None and getattr(model, "_ray_table", None) is None


def test_a_records_ticket_is_good_for_one_finish(ops):
    from ddnerf_amd import _lib

    n, nc = 64, 32
    rays, t0, raw0 = _coarse_inputs(n, nc, 9)
    head = ops.dd_head(raw0, 1.7, 0.0156)
    c0 = ops.composite_forward(raw0, t0, rays, None, head["mus"], False, True)
    ticket = ops.dd_records_launch(c0["weights"], head["mus"], head["sigmas"], head["ssig"])
    a = ops.dd_records_finish(ticket)
    assert len(a) == 3 and a[0].shape == a[1].shape == a[2].shape
    with pytest.raises(_lib.DDNerfHipError, match="finished already"):
        ops.dd_records_finish(ticket)
