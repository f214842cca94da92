"""The fused encoder + MLP kernel of the bf16 tier and its fp16 twin (ddnerf_encode_mlp_bf16_forward / _f16_, mlp_bf16_g2e.hip: run_network of
models/models.py:117-142 as ONE launch -- cast_rays, integrated_pos_enc, the view directions' encoding and the network) against the two
launches it replaces, ddnerf_encode(feat_dtype = 1) + ddnerf_mlp_bf16_forward: BIT FOR BIT, on BASELINE's fine and coarse pass, ragged
last tiles, several tiles per workgroup (the steady state: the encoder of the NEXT tile rides in the MFMA gaps of layers 6 - 8),
S / 64 not a power of two (the ray of a group comes from a multiply-high), NDC rays, both heads, and launch after launch."""
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


def _inputs(ops, kind, n, S, seed, sorted_random=True):
    o, d, rad, _ = synthetic.make_rays(kind, n, seed)
    near, far = synthetic.NEAR_FAR[kind]
    rays = ops.pack_rays(*(torch.from_numpy(x).cuda() for x in (o, d, rad)), near, far)
    g = torch.Generator(device="cuda").manual_seed(seed + 1)
    if sorted_random:      # fenceposts as a fine pass sees them: sorted, clustered, some intervals of zero width
        t = torch.rand(n, S + 1, device="cuda", generator=g) ** 2
        m = min(t[:, ::7].shape[1], t[:, 1::7].shape[1])
        t[:, 0:7 * m:7] = t[:, 1::7][:, :m]
        t = near + (far - near) * torch.sort(t, dim=1).values
    else:
        t = (near + (far - near) * torch.linspace(0, 1, S + 1, device="cuda")).expand(n, S + 1).contiguous()
    return rays, t.float().contiguous()


def _both(ops, rays, t, packed, depth, kind="bf16"):
    feat = ops.encode(rays, t, kind=kind)
    want = {"bf16": ops.mlp_bf16_forward, "fp16": ops.mlp_f16_forward}[kind](feat, packed, depth)
    got = ops.encode_mlp_bf16_forward(ops.ray_table(rays, kind), t, packed, depth, kind=kind)
    torch.cuda.synchronize()
    return got, want


def _assert_same(got, want, what):
    assert got.shape == want.shape
    same = (got == want) | (torch.isnan(got) & torch.isnan(want))
    assert bool(same.all()), (what, int((~same).any(dim=1).sum()), int((~same).any(dim=1).nonzero()[0]), float((got - want).abs().nan_to_num().max()))


@pytest.mark.parametrize("kind", ["bf16", "fp16"])
def test_ray_table_repeats_the_encoder(ops, kind):
    """the table's view-direction row IS columns 96..127 of every encoded row of that ray; its fp32 words are the packed ray's"""
    rays, t = _inputs(ops, "blender", 300, 64, 3)
    tab = ops.ray_table(rays, kind)
    feat = ops.encode(rays, t, kind=kind).view(300, 64, 128)
    dirs = tab.view(torch.bfloat16 if kind == "bf16" else torch.float16).view(300, 64)[:, 32:]
    assert torch.equal(dirs.view(torch.int16), feat[:, 0, 96:].contiguous().view(torch.int16))
    assert torch.equal(dirs.view(torch.int16), feat[:, 63, 96:].contiguous().view(torch.int16))
    assert torch.equal(tab[:, 0:6], rays[:, 0:6]) and torch.equal(tab[:, 6], rays[:, 6] * rays[:, 6])
    assert torch.equal(tab[:, 7:10], rays[:, 3:6] * rays[:, 3:6])


@pytest.mark.parametrize("kind", ["bf16", "fp16"])
@pytest.mark.parametrize("depth", [False, True])
def test_fused_kernel_bit_identical_to_encode_then_mlp(ops, depth, kind):
    flat = _flat(depth, 12, 20.0)
    packed = {"bf16": ops.mlp_bf16_pack, "fp16": ops.mlp_f16_pack}[kind](flat, depth)
    n_cu = torch.cuda.get_device_properties(0).multi_processor_count
    cases = [("blender", 4096, 128, True),          # BASELINE's fine pass: four tiles per workgroup
             ("blender", 4096, 64, False),          # its coarse pass: two
             ("blender", 1, 64, True),              # one group of one ragged tile
             ("blender", 100, 64, True),            # 12.5 tiles
             ("blender", 7, 192, True),             # S / 64 = 3: groups per ray not a power of two
             ("llff", 1000, 128, True),             # NDC rays (near 0: means near zero, wide Gaussians)
             ("real360", 2 * n_cu * 4 + 5, 128, True)]   # every workgroup two tiles and a few a third, ragged
    for rkind, n, S, rnd in cases:
        rays, t = _inputs(ops, rkind, n, S, n + S, rnd)
        got, want = _both(ops, rays, t, packed, depth, kind)
        _assert_same(got, want, (kind, rkind, n, S, depth))
        assert bool(torch.isfinite(want).all())


def test_fused_kernel_rejects_what_it_does_not_cover(ops):
    from ddnerf_amd import _lib
    flat = _flat(False, 3, 1.0)
    packed = ops.mlp_bf16_pack(flat, False)
    rays, t = _inputs(ops, "blender", 8, 48, 1)
    assert not ops.encode_mlp_bf16_supported(48, 8 * 48) and ops.encode_mlp_bf16_supported(128, 4096 * 128)
    with pytest.raises(_lib.DDNerfHipError):
        ops.encode_mlp_bf16_forward(ops.ray_table(rays), t, packed, False)


def test_fused_kernel_launch_after_launch(ops):
    """the scratch rows are rewritten by every launch and every tile: 60 launches on fresh fenceposts, each against the two-launch path"""
    flat = _flat(True, 5, 4.0)
    packed = ops.mlp_bf16_pack(flat, True)
    rays, _ = _inputs(ops, "blender", 2048, 128, 11)
    tab = ops.ray_table(rays)
    for it in range(60):
        _, t = _inputs(ops, "blender", 2048, 128, 100 + it)
        got = ops.encode_mlp_bf16_forward(tab, t, packed, True)
        want = ops.mlp_bf16_forward(ops.encode(rays, t, kind="bf16"), packed, True)
        torch.cuda.synchronize()
        _assert_same(got, want, it)


@pytest.mark.parametrize("tier", ["bf16", "fp16"])
@pytest.mark.parametrize("name", ["fullsize_cfg2_dd_blender_4096_64x128", "fullsize_cfg3_dd_llff_4096_64x128", "fullsize_trained_dd_blender_4096_64x128"])
def test_run_iter_is_the_same_with_the_encoder_inside_or_outside_the_mlp_kernel(name, tier, monkeypatch):
    """The whole render pass of the bf16 tier at BASELINE size: DDNERF_FUSE_ENCODER = all (default: both passes one launch each), fine (the
    coarse pass keeps its encode launch) and 0 (two launches per pass) give the SAME output dict, bit for bit -- the kernels are bit-identical, so
    every downstream value (sampler, compositing, dp loss, records) is too.  Also as an image of several chunks (the per-chunk ray table)."""
    from _cases import load_fullsize
    from ddnerf_amd import models as M
    from test_hip_run_iter import build_model

    c = load_fullsize(name)
    ro, rd, rad, tgt = (torch.from_numpy(x).cuda() for x in synthetic.make_rays(c["kind"], c["n"], 1))

    def run(mode, chunk=None):
        monkeypatch.setattr(M, "FUSE_ENCODER", mode)
        model = build_model(c)
        model.cfg.nerf["mlp_dtype"] = tier
        model._set_mlp_dtype()
        model.eval()
        if chunk:
            model.cfg.nerf.validation["chunksize"] = chunk
        with torch.no_grad():
            return model.run_iter(ro, rd, rad, mode="validation", rgb_target=tgt)

    ref = run("0")
    for mode, chunk in (("all", None), ("fine", None), ("all", 1024)):
        out = run(mode, chunk)
        assert out.keys() == ref.keys()
        for lvl in ref:
            for k, v in ref[lvl].items():
                if v is None or v is False or chunk and k in ("dp_loss", "mus_reg", "sig_reg", "mus_loss", "sig_loss", "mus", "sigmas", "smoothed_sigmas"):
                    continue          # (per-chunk scalars and records have another shape when the batch is split)
                w = out[lvl][k]
                assert w.shape == v.shape, (mode, chunk, lvl, k)
                assert torch.equal(torch.nan_to_num(w), torch.nan_to_num(v)), (mode, chunk, lvl, k)
