"""The view-direction columns once per RAY (round-4 review, item 3; the reference encodes a ray's direction once and broadcasts it over the
samples, models/models.py:128-133): ddnerf_encode_rays + ddnerf_mlp_f32_forward_rays / ddnerf_mlp_x3_forward_rays against ddnerf_encode +
the plain forwards on rows that carry the columns -- BIT FOR BIT --, and the whole render pass with and without them."""
import numpy as np
import pytest
import torch

from ddnerf_amd import synthetic

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    from ddnerf_amd import ops as _ops
    return _ops


def _flat(depth, seed, sharpen):
    sd = synthetic.make_state_dict(depth, seed, sharpen)
    names = [n for n, _, _ in synthetic.layer_table(depth)]
    return torch.from_numpy(np.concatenate([np.concatenate([sd[k + ".weight"].ravel(), sd[k + ".bias"].ravel()]) for k in names])).cuda()


def _inputs(ops, kind, n, S, seed):
    o, d, rad, _ = synthetic.make_rays(kind, n, seed)
    near, far = synthetic.NEAR_FAR[kind]
    rays = ops.pack_rays(*(torch.from_numpy(x).cuda() for x in (o, d, rad)), near, far)
    g = torch.Generator(device="cuda").manual_seed(seed + 1)
    t = near + (far - near) * torch.sort(torch.rand(n, S + 1, device="cuda", generator=g), dim=1).values
    return (o, d, rad), rays, t.float().contiguous()


@pytest.mark.parametrize("kind,n,S,cyl", [("blender", 4096, 128, False), ("blender", 300, 64, False), ("llff", 41, 33, False), ("real360", 7, 5, True),
                                          ("blender", 1000, 2, False)])
def test_encode_rays_is_encode_without_the_view_direction_columns(ops, kind, n, S, cyl):
    _, rays, t = _inputs(ops, kind, n, S, n + S)
    full = ops.encode(rays, t, cylinder=cyl, kind="fp32").view(n, S, 128)
    feat, dirs = ops.encode_rays(rays, t, cylinder=cyl)
    torch.cuda.synchronize()
    assert torch.equal(feat.view(n, S, 128)[..., :96], full[..., :96])
    for j in (0, S - 1):                                   # the table row of a ray IS columns 96..127 of each of its rows
        assert torch.equal(dirs, full[:, j, 96:])


@pytest.mark.parametrize("mlp", ["fp32", "x3"])
@pytest.mark.parametrize("depth", [False, True])
def test_forward_rays_bit_identical_to_the_forward_on_full_rows(ops, mlp, depth):
    flat = _flat(depth, 12, 20.0)
    packed = {"fp32": ops.mlp_f32_pack, "x3": ops.mlp_x3_pack}[mlp](flat, depth)
    plain = {"fp32": ops.mlp_f32_forward, "x3": ops.mlp_x3_forward}[mlp]
    by_ray = {"fp32": ops.mlp_f32_forward_rays, "x3": ops.mlp_x3_forward_rays}[mlp]
    n_cu = torch.cuda.get_device_properties(0).multi_processor_count
    for kind, n, S in (("blender", 4096, 128), ("blender", 4096, 64), ("llff", 41, 33), ("blender", 3, 7), ("real360", 2 * n_cu + 3, 128)):
        _, rays, t = _inputs(ops, kind, n, S, n + 3 * S)
        want = plain(ops.encode(rays, t, kind="fp32"), packed, depth)
        feat, dirs = ops.encode_rays(rays, t)
        feat.view(n, S, 128)[..., 96:] = float("nan")       # (the forward must not read them)
        got = by_ray(feat, dirs, S, packed, depth)
        torch.cuda.synchronize()
        assert torch.equal(got, want), (mlp, kind, n, S, int((got != want).any(dim=1).sum()))


@pytest.mark.parametrize("mlp", ["fp32", "x3"])
@pytest.mark.parametrize("name", ["fullsize_cfg2_dd_blender_4096_64x128", "fullsize_cfg5_mip_blender_4096_64x128", "fullsize_trained_dd_llff_4096_64x128"])
def test_run_iter_is_the_same_with_per_ray_and_per_sample_view_directions(name, mlp, monkeypatch):
    from _cases import load_fullsize
    from ddnerf_amd import models as M
    from test_hip_run_iter import build_model

    c = load_fullsize(name)
    ro, rd, rad, tgt = (torch.from_numpy(x).cuda() for x in synthetic.make_rays(c["kind"], c["n"], 1))

    def run(on, chunk=None):
        monkeypatch.setattr(M, "RAY_DIRS", on)
        model = build_model(c)
        model.cfg.nerf["mlp_dtype"] = mlp
        model._set_mlp_dtype()
        model.eval()
        if chunk:
            model.cfg.nerf.validation["chunksize"] = chunk
        with torch.no_grad():
            return model.run_iter(ro, rd, rad, mode="validation", rgb_target=tgt)

    ref = run(False)
    for chunk in (None, 1500):
        out = run(True, chunk)
        for lvl in ref:
            for k in ("rgb", "depth", "acc", "disp", "weights"):
                assert torch.equal(torch.nan_to_num(out[lvl][k]), torch.nan_to_num(ref[lvl][k])), (chunk, lvl, k)


@pytest.mark.parametrize("depth", [False, True])
def test_training_forward_takes_per_ray_view_directions(ops, depth):
    """(round 5) the fp32 tier's values-record training forward with the view-direction columns from the per-ray table: raw, the activation
    record (the transposed input columns included: the weight gradients of layers_dir contract them) and the sign record BIT FOR BIT those of
    the forward on full rows."""
    flat = _flat(depth, 13, 20.0)
    packed = ops.mlp_f32_pack(flat, depth)
    for kind, n, S in (("blender", 512, 128), ("llff", 41, 33), ("blender", 3, 7)):
        _, rays, t = _inputs(ops, kind, n, S, n + 5 * S)
        raw0, acts0, signs0 = ops.mlp_f32_forward_train(ops.encode(rays, t, kind="fp32"), packed, depth, rec="values")
        feat, dirs = ops.encode_rays(rays, t)
        feat.view(n, S, 128)[..., 96:] = float("nan")
        raw1, acts1, signs1 = ops.mlp_f32_forward_train(feat, packed, depth, rec="values", dirs=dirs, S=S)
        torch.cuda.synchronize()
        M = n * S
        assert torch.equal(raw1, raw0)
        a0, a1 = ops.x3_unblock(acts0)[:2555, :M], ops.x3_unblock(acts1)[:2555, :M]
        assert torch.equal(a1, a0) and not torch.isnan(a1).any()
        assert torch.equal(signs1, signs0)


@pytest.mark.parametrize("depth", [False, True])
def test_x3_training_forward_takes_per_ray_view_directions(ops, depth):
    """the x3 tier's training forward likewise: raw, the row-pair record and the sign words bit for bit"""
    flat = _flat(depth, 14, 20.0)
    packed = ops.mlp_x3_pack(flat, depth)
    for kind, n, S in (("blender", 512, 128), ("llff", 41, 33), ("blender", 3, 7)):
        _, rays, t = _inputs(ops, kind, n, S, n + 7 * S)
        raw0, acts0, bits0 = ops.mlp_x3_forward_train(ops.encode(rays, t, kind="fp32"), packed, depth)
        feat, dirs = ops.encode_rays(rays, t)
        feat.view(n, S, 128)[..., 96:] = float("nan")
        raw1, acts1, bits1 = ops.mlp_x3_forward_train(feat, packed, depth, dirs=dirs, S=S)
        torch.cuda.synchronize()
        M = n * S
        assert torch.equal(raw1, raw0)
        a0, a1 = ops.x3_unpair(acts0)[:2555, :M], ops.x3_unpair(acts1)[:2555, :M]
        assert torch.equal(a1, a0) and not torch.isnan(a1).any()
        # (sign words of the rows the backward masks: layers_xyz.0-7 = record rows 0..2047 = words 0..127, layers_dir.0 = rows 2304..2431 = words
        # 144..151; the other words of the buffer are never written)
        for lo, hi in ((0, 128), (144, 152)):
            assert torch.equal(bits1[lo:hi, :M], bits0[lo:hi, :M])


@pytest.mark.parametrize("mlp_dtype", ["fp32", "x3"])
def test_training_step_is_the_same_with_per_ray_and_per_sample_view_directions(mlp_dtype, monkeypatch):
    """loss and every parameter gradient of one run_iter training pass, RAY_DIRS on against off: bit-identical"""
    from _cases import load_fullsize
    from ddnerf_amd import models as M
    from test_hip_run_iter import build_model

    c = load_fullsize("fullsize_cfg2_dd_blender_4096_64x128")
    ro, rd, rad, tgt = (torch.from_numpy(x).cuda()[:1024] for x in synthetic.make_rays(c["kind"], c["n"], 1))

    def run(on):
        monkeypatch.setattr(M, "RAY_DIRS", on)
        model = build_model(c)
        model.cfg.nerf["mlp_dtype"] = mlp_dtype
        model._set_mlp_dtype()
        model.rng = M.TorchRng()            # (the fixture is a validation case: its replay list is empty; training draws)
        torch.manual_seed(0)
        model.train()
        out = model.run_iter(ro, rd, rad, mode="train", rgb_target=tgt)
        loss = sum(((out[j]["rgb"] - tgt) ** 2).mean() for j in range(2)) + 0.01 * out[1]["dp_loss"].mean()
        loss.backward()
        return loss.detach(), [p.grad.clone() for net in (model.coarse, model.fine) for p in net.parameters()], model

    l0, g0, _ = run(False)
    l1, g1, model = run(True)
    assert getattr(model.fine, "_fwd_calls", 0) >= 1
    assert torch.equal(l0, l1)
    for a, b in zip(g0, g1):
        assert torch.equal(a, b)
