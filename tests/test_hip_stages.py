"""GPU parity tests, stage by stage: each HIP kernel (through the C ABI) against the CPU oracle on the same
seeded inputs and against the golden vectors produced by the reference.  Bars: integer/index results and
every quantity that only uses IEEE +,-,*,/ are bit-exact; results through sin/exp/erf/log are held to a few
fp32 ulps; the fused fp32 MLP to 2e-6 absolute (north_star: RGB/depth within 1e-4)."""
import numpy as np
import pytest
import torch

import oracle as O
from _cases import maxerr, relerr
from ddnerf_amd import synthetic

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    from ddnerf_amd import ops as _ops
    return _ops


def dev(x):
    return None if x is None else torch.from_numpy(np.ascontiguousarray(x)).cuda()


def host(t):
    return t.detach().cpu().numpy()


def flat_params(sd, depth_head):
    names = [n for n, _, _ in synthetic.layer_table(depth_head)]
    return np.concatenate([np.concatenate([sd[n + ".weight"].ravel(), sd[n + ".bias"].ravel()]) for n in names])


@pytest.mark.parametrize("name", ["blender_cone", "llff_cone", "real360_cone", "blender_cylinder"])
def test_pack_rays_encode_mlp(ops, golden, name):
    g = golden("encode_" + name)
    cyl = name.endswith("cylinder")
    rays = ops.pack_rays(dev(g["ro"]), dev(g["rd"]), dev(g["rad"]), float(g["near"]), float(g["far"]))
    assert np.array_equal(host(rays), O.pack_rays(g["ro"], g["rd"], g["rad"], float(g["near"]), float(g["far"])))
    assert maxerr(host(rays), g["rays"]) <= 1.2e-7
    feat = host(ops.encode(dev(g["rays"]), dev(g["t_vals"]), cylinder=cyl))
    assert feat.shape[1] == 128 and np.all(feat[:, 123:] == 0)
    ofeat = O.encode(g["rays"], g["t_vals"], cyl)
    assert maxerr(feat[:, :123], ofeat) <= 4e-7          # vs oracle (libm vs ocml sin/exp)
    assert maxerr(feat[:, :96], g["ipe"].reshape(-1, 96)) <= 4e-7   # vs the reference itself
    assert maxerr(feat[:, 96:123].reshape(g["rays"].shape[0], -1, 27)[:, 0], g["dirs"]) <= 2.4e-7
    # bf16 feature variant (columns in MFMA k-order): the fp32 feature rounded to nearest-even, except that its sin / exp come
    # from the hardware transcendentals (absolute error ~2e-6) and safe_sin's remainder from one fma (<= 2e-5, rays_encode.hip): a
    # few values per hundred sit on the other side of a bf16 rounding boundary (one bf16 ulp), and values near zero differ by
    # that absolute error
    fb = ops.encode(dev(g["rays"]), dev(g["t_vals"]), cylinder=cyl, bf16=True)
    want = torch.from_numpy(feat[:, ops.K_ORDER]).to(torch.bfloat16)
    diff = (fb.cpu().float() - want.float()).abs()
    assert bool((diff <= want.float().abs() * 2.0 ** -7 + 3e-5).all()), float(diff.max())
    assert float((fb.cpu() == want).float().mean()) >= 0.95
    # bf16-MFMA MLP: bf16 operands, fp32 accumulate -- its own tolerance tier
    for depth, seed, key in ((True, 11, "raw6"), (False, 12, "raw4")):
        sd = synthetic.make_state_dict(depth, seed)
        packed = ops.mlp_bf16_pack(dev(flat_params(sd, depth)), depth)
        raw = host(ops.mlp_bf16_forward(fb, packed, depth))
        ref = g[key].reshape(raw.shape)
        assert maxerr(raw, ref) <= 4e-3, key
    # fused fp32 MLP on the golden features
    for depth, seed, key in ((True, 11, "raw6"), (False, 12, "raw4")):
        sd = synthetic.make_state_dict(depth, seed)
        packed = ops.mlp_f32_pack(dev(flat_params(sd, depth)), depth)
        raw = host(ops.mlp_f32_forward(dev(feat), packed, depth))
        ref = g[key].reshape(raw.shape)
        assert maxerr(raw, ref) <= 2e-6, key
        assert maxerr(raw, O.mlp_forward(feat, sd, depth)) <= 2e-6, key


def test_mlp_f32_ragged_and_large(ops):
    """M not a multiple of the 128-sample tile; sharpened weights (large activations); vs the oracle."""
    rng = np.random.default_rng(5)
    for M, depth, sharpen in ((1, True, 1.0), (127, False, 20.0), (129, True, 20.0), (1000, False, 1.0)):
        feat = np.zeros((M, 128), np.float32)
        feat[:, :123] = rng.uniform(-1, 1, (M, 123)).astype(np.float32)
        sd = synthetic.make_state_dict(depth, 3, sharpen)
        packed = ops.mlp_f32_pack(dev(flat_params(sd, depth)), depth)
        raw = host(ops.mlp_f32_forward(dev(feat), packed, depth))
        ref = O.mlp_forward(feat, sd, depth)
        assert maxerr(raw, ref) <= 2e-6 * max(1.0, np.abs(ref).max()), (M, depth)


def test_mlp_f32_persistent_tiles(ops):
    """The fp32 forward is persistent (one workgroup per CU walks the 128-sample tiles, handing the next tile's first weight slice,
    fragments, bias tile and features over in registers / LDS): more tiles than workgroups with a ragged last tile, against the oracle
    on rows of the first tile, of a second-trip tile and of the ragged tail, and row-for-row against launches of those tiles alone;
    the training forward (same body with a recorder) writes the records of the same activations at that size."""
    rng = np.random.default_rng(15)
    M, depth = 128 * (2 * 256 + 3) + 77, True     # 515 full tiles + 77 samples: workgroups take 2 or 3 tiles
    feat = np.zeros((M, 128), np.float32)
    feat[:, :123] = rng.uniform(-1, 1, (M, 123)).astype(np.float32)
    sd = synthetic.make_state_dict(depth, 3, 4.0)
    flat = dev(flat_params(sd, depth))
    packed = ops.mlp_f32_pack(flat, depth)
    fd = dev(feat)
    raw = ops.mlp_f32_forward(fd, packed, depth)
    for lo, hi in ((0, 128), (128 * 256, 128 * 257 + 5), (128 * 513, 128 * 514), (M - 77 - 128, M)):
        ref = O.mlp_forward(feat[lo:hi], sd, depth)
        assert maxerr(host(raw[lo:hi]), ref) <= 2e-6 * max(1.0, np.abs(ref).max()), (lo, hi)
        alone = ops.mlp_f32_forward(fd[lo:hi].contiguous(), packed, depth)     # the same rows as the first tiles of their workgroups
        assert torch.equal(alone, raw[lo:hi]), (lo, hi)
    raw_t, acts = ops.mlp_f32_forward_train(fd, packed, depth)
    raw_r, rec = ops.mlp_f32_forward_train(fd, packed, depth, rec=True)
    assert torch.equal(raw_t, raw) and torch.equal(raw_r, raw)
    ld = acts.shape[1]
    assert torch.equal(rec.view(torch.int32)[:, :ld].reshape(-1, 2560, 16)[: M // 16, :2555], ops.x3_split(acts).view(torch.int32).reshape(-1, 2560, 16)[: M // 16, :2555])
    cols = torch.tensor([0, 127, 128 * 256 + 3, 128 * 514 + 100, M - 1], device="cuda")
    a = host(acts[:, cols])                         # [2560, 5]: layer outputs (post-ReLU), fc_feat, dir hidden, input columns
    for k, c in enumerate(cols.tolist()):
        x = feat[c]
        assert np.array_equal(a[2432:2555, k], x[:123]), c
        h0 = np.maximum(sd["layers_xyz.0.weight"].astype(np.float64) @ x[:96].astype(np.float64) + sd["layers_xyz.0.bias"], 0.0)
        assert np.abs(a[0:256, k] - h0).max() <= 1e-5 * max(1.0, np.abs(h0).max()), c


def test_mlp_f32_many_launches_stay_identical(ops):
    """The fp32 kernels hand data from slice to slice and tile to tile through LDS on ONE barrier per slice (a tile's first fragments
    and bias tile are read behind the barrier of the slice before; activations and deltas ride through a per-wave scratch area without
    any): 40 back-to-back launches of the forward, the record-writing training forward and its backward at the benchmark's fine-pass
    size must reproduce the first launch bit for bit -- a missing wait or a misplaced barrier shows up as a rare differing row."""
    M, depth = 4096 * 128, True
    g = torch.Generator(device="cuda").manual_seed(3)
    feat = torch.zeros(M, 128, device="cuda")
    feat[:, :123] = torch.rand(M, 123, device="cuda", generator=g) * 2 - 1
    flat = dev(flat_params(synthetic.make_state_dict(depth, 5, 4.0), depth))
    packed, packed_t = ops.mlp_f32_pack(flat, depth), ops.mlp_f32_pack_t(flat, depth)
    G = torch.randn(M, 6, device="cuda", generator=g)
    raw0 = ops.mlp_f32_forward(feat, packed, depth).clone()
    raw_t0, rec0 = ops.mlp_f32_forward_train(feat, packed, depth, rec=True)
    raw_t0, rec0 = raw_t0.clone(), rec0.clone()
    blocks = lambda t, rows: t.view(torch.int32).reshape(-1, 2560, 16)[:, :rows]    # (the rows a record defines: the rest is never written)
    d0 = blocks(ops.mlp_f32_backward_data(G, packed_t, rec0, depth, rec=True), 2438).clone()
    rec0 = blocks(rec0, 2555).clone()
    assert torch.equal(raw0, raw_t0)
    for it in range(40):
        assert torch.equal(ops.mlp_f32_forward(feat, packed, depth), raw0), it
        if it % 4 == 0:
            raw_t, rec = ops.mlp_f32_forward_train(feat, packed, depth, rec=True)
            assert torch.equal(raw_t, raw0) and torch.equal(blocks(rec, 2555), rec0), it
            assert torch.equal(blocks(ops.mlp_f32_backward_data(G, packed_t, rec, depth, rec=True), 2438), d0), it
            del raw_t, rec


def test_dd_records_match_boolean_indexing(ops):
    """ops.dd_records (three small kernels) against the reference's expression pdf = w / w.sum(-1); x[pdf > 0.1] evaluated by
    torch on the CPU (ATen's summation order), incl. an all-zero row (NaN pdf: nothing selected) and ragged widths"""
    rng = np.random.default_rng(8)
    for n, nc in ((37, 64), (5, 16), (130, 33), (3, 1)):
        w = (rng.random((n, nc)) ** 6).astype(np.float32)
        w[n // 2] = 0.0
        mus, sig, ssig = (rng.random((n, nc)).astype(np.float32) for _ in range(3))
        wt = torch.from_numpy(w)
        mask = (wt / torch.sum(wt, dim=-1, keepdim=True)) > 0.1
        got = ops.dd_records(dev(w), dev(mus), dev(sig), dev(ssig))
        for g, src in zip(got, (mus, sig, ssig)):
            assert np.array_equal(host(g), torch.from_numpy(src)[mask].numpy()), (n, nc)


def test_mlp_x3_fp32_class_accuracy(ops):
    """The bf16x3 kernel (exact hi/lo operand splits, three MFMAs per product) against the oracle's fp32 evaluation:
    ragged sizes, both heads, default and sharpened weights.  Bar: 2e-5 of the output scale -- 30x looser than the exact
    fp32 kernel's bar, 50x tighter than what plain bf16 reaches, and far inside the 1e-4 RGB parity budget."""
    rng = np.random.default_rng(5)
    for M, depth, sharpen in ((1, True, 1.0), (127, False, 20.0), (129, True, 20.0), (1000, False, 1.0), (4097, True, 4.0)):
        feat = np.zeros((M, 128), np.float32)
        feat[:, :123] = rng.uniform(-1, 1, (M, 123)).astype(np.float32)
        sd = synthetic.make_state_dict(depth, 3, sharpen)
        packed = ops.mlp_x3_pack(dev(flat_params(sd, depth)), depth)
        raw = host(ops.mlp_x3_forward(dev(feat), packed, depth))
        ref = O.mlp_forward(feat, sd, depth)
        assert maxerr(raw, ref) <= 2e-5 * max(1.0, np.abs(ref).max()), (M, depth, maxerr(raw, ref))


def test_mlp_bf16_ragged_vs_bf16_emulation(ops):
    """bf16 kernel against an fp64 evaluation of the SAME bf16-rounded weights/features/activations:
    isolates kernel bugs (wrong k-permutation, tile maps) from the expected bf16 quantisation error."""
    rng = np.random.default_rng(6)

    def bf(x):
        return torch.from_numpy(np.asarray(x, np.float32)).to(torch.bfloat16).to(torch.float64).numpy()

    # 70001 samples = 274 tiles of 256: more tiles than CUs, so the persistent workgroups walk on to a second tile
    # (features prefetched during the first) and the last tile is ragged
    for M, depth in ((1, True), (255, False), (257, True), (1500, False), (70001, True), (66000, False)):
        feat = np.zeros((M, 128), np.float32)
        feat[:, :123] = rng.uniform(-1, 1, (M, 123)).astype(np.float32)
        sd = synthetic.make_state_dict(depth, 4, 1.0)
        W = {k: (bf(v) if k.endswith("weight") else v.astype(np.float64)) for k, v in sd.items()}
        x = bf(feat)
        xyz, dirs = x[:, :96], x[:, 96:123]
        h = xyz
        for i in range(8):
            inp = np.concatenate([xyz, h], 1) if i == 5 else h
            h = bf(np.maximum(inp @ W["layers_xyz.%d.weight" % i].T + W["layers_xyz.%d.bias" % i], 0))
        ft = bf(h @ W["fc_feat.weight"].T + W["fc_feat.bias"])
        alpha = ft @ W["fc_alpha.weight"].T + W["fc_alpha.bias"]
        hd = bf(np.maximum(np.concatenate([ft, dirs], 1) @ W["layers_dir.0.weight"].T + W["layers_dir.0.bias"], 0))
        outs = [hd @ W["fc_rgb.weight"].T + W["fc_rgb.bias"], alpha]
        if depth:
            outs.append(hd @ W["fc_mu_sigma.weight"].T + W["fc_mu_sigma.bias"])
        ref = np.concatenate(outs, 1)
        fb = torch.from_numpy(np.ascontiguousarray(feat[:, ops.K_ORDER])).to(torch.bfloat16).cuda().contiguous()
        packed = ops.mlp_bf16_pack(dev(flat_params(sd, depth)), depth)
        raw = host(ops.mlp_bf16_forward(fb, packed, depth))
        # The kernel sums each dot product in a different order than numpy, so a pre-activation that sits on a bf16 rounding
        # boundary may round the other way: one such flip moves an output by ~1e-4.  Bulk agreement is ~1e-8; a wrong
        # k-permutation or tile map puts EVERY entry off by ~0.1.
        col_scale = np.maximum(np.abs(ref).max(0), 0.1)
        err = np.abs(raw - ref) / col_scale
        assert (err <= 2e-5).mean() >= 0.97, (M, depth, (err <= 2e-5).mean())
        assert err.max() <= 6e-3, (M, depth, err.max(0))


@pytest.mark.parametrize("tag", ["lin", "disp", "ndc"])
def test_first_cycle_bit_exact(ops, golden, tag):
    g = golden("first_cycle")
    near, far, nc, lind = g[tag + "_meta"]
    rays = np.zeros((19, 12), np.float32)
    rays[:, 7], rays[:, 8] = near, far
    for mode, pert in (("train", True), ("validation", False)):
        t = ops.sample_first_cycle(dev(rays), dev(g[tag + "_lin"]), dev(g[tag + "_train_rand"]) if pert else None, bool(lind))
        assert np.array_equal(host(t), g["%s_%s_t" % (tag, mode)])


@pytest.mark.parametrize("tag", ["blender_mus_noise", "blender_plain", "blender_white", "llff_white", "real360_mus",
                                 "blender_empty"])
def test_composite(ops, golden, tag):
    g = golden("composite_" + tag)
    n = g["raw"].shape[0]
    rays = np.zeros((n, 12), np.float32)
    rays[:, 3:6] = g["rd"]
    white, blender = bool(g["flags"][0]), bool(g["flags"][1])
    o = ops.composite_forward(dev(g["raw"]), dev(g["t_vals"]), dev(rays), dev(g.get("noise")), dev(g.get("mus")), white,
                              blender, want_rgb=True)
    oo = O.composite(g["raw"], g["t_vals"], rays, g.get("noise"), g.get("mus"), white, blender)
    for k in ("rgb_map", "disp", "acc", "weights", "depth", "rgb"):
        assert relerr(host(o[k]), oo[k]) <= 1e-6, k      # oracle
        assert relerr(host(o[k]), g[k]) <= 2e-6, k       # reference
    if "cdisp" in g:
        assert relerr(host(o["cdisp"]), g["cdisp"]) <= 2e-6


@pytest.mark.parametrize("tag", ["c64f129", "c16f17", "c33f70", "c1f9"])
def test_samplers_indices_bit_exact(ops, golden, tag):
    g = golden("sampler_" + tag)
    near, far, nc, ns = g["meta"]
    d = {k: dev(g[k]) for k in ("bins", "weights", "mus", "sigmas", "part", "left")}
    for pad in (1, 0):
        for det in (1, 0):
            key = "pad%d_det%d" % (pad, det)
            rnd = None if det else g["rand"]
            ub = g["u_dd_det"] if det else g["arange_dd"]
            s, ind = ops.sample_pdf_mu_sigma(d["bins"], d["weights"], d["mus"], d["sigmas"], d["part"], d["left"], dev(ub),
                                             dev(rnd), near, far, bool(pad), want_ind=True)
            so, indo = O.sample_pdf_mu_sigma(g["bins"], g["weights"], g["mus"], g["sigmas"], g["part"], g["left"], ub, rnd,
                                             near, far, bool(pad))
            assert np.array_equal(host(ind), indo), key            # bit-exact sample indices vs oracle
            if "ddind_" + key in g:
                assert np.array_equal(host(ind), g["ddind_" + key]), key   # ... and vs the reference
            assert maxerr(host(s), g["dd_" + key]) <= 1e-6, key
            assert np.all(np.diff(host(s), axis=1) >= 0)
            if "mip_" + key in g:
                ubm = g["u_mip_det"] if det else g["arange_mip"]
                sm = ops.sample_pdf(d["bins"], d["weights"], dev(ubm), dev(rnd), bool(pad))
                assert np.array_equal(host(sm), g["mip_" + key]), key      # pure IEEE ops: bit-exact


@pytest.mark.parametrize("tag", ["blender_drop", "blender_full", "llff", "blender_allzero"])
def test_dp_loss_forward(ops, golden, tag):
    g = golden("dploss_" + tag)
    blender = bool(g["is_blender"])
    v = ops.dp_loss_forward(*[dev(g[k]) for k in ("t1", "t0", "w1", "w0", "mus", "sig", "left", "part")], blender)
    ref = float(g["loss"])
    vo, rows = O.dp_loss(g["t1"], g["t0"], g["w1"], g["w0"], g["mus"], g["sig"], g["left"], g["part"], blender)
    assert abs(float(v) - ref) <= 2e-6 * max(abs(ref), 1e-6)      # incl. the row-filter misalignment quirk
    assert abs(float(v) - vo) <= 2e-6 * max(abs(vo), 1e-6)
    if tag == "blender_allzero":
        assert float(v) == 0.0 and rows == 0


def test_dd_head(ops):
    rng = np.random.default_rng(8)
    raw6 = (rng.standard_normal((37, 64, 6)) * 2).astype(np.float32)
    d = ops.dd_head(dev(raw6), 1.7, 0.0156)
    o = O.dd_head(raw6, 1.7, 0.0156)
    for k in ("mus", "sigmas", "left", "part", "ssig", "sleft", "spart"):
        assert maxerr(host(d[k]), o[k]) <= 2.4e-7, k
    sc = host(d["scal"])
    for i, k in enumerate(("mus_loss", "sig_loss", "mus_reg", "sig_reg")):
        assert abs(sc[i] - o[k]) <= 2e-6 * abs(o[k]), k


def test_full_size_properties(ops):
    """BASELINE config 2 shapes (4096 rays x 64/128): size-independent properties of the whole chain."""
    n, nc, nf = 4096, 64, 128
    ro, rd, rad, _ = synthetic.make_rays("blender", n, 1)
    rays = ops.pack_rays(dev(ro), dev(rd), dev(rad), 2.0, 6.0)
    t_lin = torch.linspace(0.0, 1.0, nc + 1).cuda()
    t0 = ops.sample_first_cycle(rays, t_lin, torch.rand(n, nc + 1, device="cuda"))
    assert bool((t0[:, 1:] >= t0[:, :-1]).all()) and bool((t0[:, 0] == 2.0).all()) and bool((t0[:, -1] == 6.0).all())
    feat = ops.encode(rays, t0)
    assert feat.shape == (n * nc, 128) and bool(torch.isfinite(feat).all()) and float(feat[:, :96].abs().max()) <= 1.0
    sd = synthetic.make_state_dict(True, 11, 20.0)
    packed = ops.mlp_f32_pack(dev(flat_params(sd, True)), True)
    raw = ops.mlp_f32_forward(feat, packed, True).reshape(n, nc, 6)
    # spot-check 256 random samples of the big launch against the oracle
    idx = np.random.default_rng(0).integers(0, n * nc, 256)
    ref = O.mlp_forward(host(feat)[idx], sd, True)
    assert maxerr(host(raw.reshape(-1, 6))[idx], ref) <= 2e-6 * max(1.0, np.abs(ref).max())
    head = ops.dd_head(raw, 1.7, 1 / 64)
    c0 = ops.composite_forward(raw, t0, rays, None, head["mus"], False, True)
    w = c0["weights"]
    assert bool((w >= 0).all()) and bool((c0["acc"] <= 1.0 + 1e-5).all())
    u = torch.linspace(0.0, 0.9999, nf + 1).cuda()
    t1, ind = ops.sample_pdf_mu_sigma(t0, w, head["mus"], head["ssig"], head["spart"], head["sleft"], u, None, 2.0, 6.0,
                                      True, want_ind=True)
    assert bool((t1[:, 1:] >= t1[:, :-1]).all())                        # sortedness
    assert bool((t1[:, 0] == 2.0).all()) and bool((t1[:, -1] == 6.0).all())
    assert int(ind.min()) >= 0 and int(ind.max()) <= nc - 1
    assert bool((ind[:, 1:] >= ind[:, :-1]).all())                      # det u is increasing -> bins are too
    # idempotence: same inputs -> bit-identical outputs (no atomics / races anywhere on the path)
    t1b = ops.sample_pdf_mu_sigma(t0, w, head["mus"], head["ssig"], head["spart"], head["sleft"], u, None, 2.0, 6.0, True)
    assert torch.equal(t1, t1b)
    raw_b = ops.mlp_f32_forward(feat, packed, True).reshape(n, nc, 6)
    assert torch.equal(raw, raw_b)


@pytest.mark.parametrize("tag", ["a", "b"])
def test_ray_generation(ops, golden, tag):
    """get_ray_bundle / ndc_mipnerf_rays on device vs the reference (next-tier row: the caller upstream of the path)"""
    g = golden("raygen")
    H, W, focal = g[tag + "_hwf"]
    H, W = int(H), int(W)
    o, d, r = ops.ray_bundle(H, W, float(focal), g[tag + "_pose"])
    assert np.array_equal(host(o), g[tag + "_o"])
    assert maxerr(host(d), g[tag + "_d"]) <= 2.4e-7
    assert maxerr(host(r), g[tag + "_r"]) <= 1e-7 * float(np.abs(g[tag + "_r"]).max()) + 1e-9
    on, dn, rn = ops.ndc_rays(H, W, float(focal), dev(g[tag + "_o"]), dev(g[tag + "_d"]), 1.0)
    assert relerr(host(on), g[tag + "_on"]) <= 2e-6 and relerr(host(dn), g[tag + "_dn"]) <= 2e-6
    assert relerr(host(rn), g[tag + "_rn"]) <= 2e-5
