"""The fp16 MLP tier (mlp_f16.hip / mlp_f16_g2.hip: the bf16 kernels built on v_mfma_f32_16x16x32_f16 and v_cvt_pk_f16_f32): its two
kernels agree bit for bit, its operand rounding is that of an 11-bit significand (8x below the bf16 kernel's), and at BASELINE sizes
it is held against the reference's own outputs.  Reference stage: models/base_architectures.py:40-61, 103-126 (fp32)."""
import numpy as np
import pytest
import torch

from _cases import fullsize_names, load_fullsize
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


def _rows(M, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    feat = torch.zeros(M, 128, device="cuda")
    feat[:, :123] = torch.rand(M, 123, device="cuda", generator=g) * 2 - 1
    return feat


@pytest.mark.parametrize("depth", [False, True])
def test_fp16_kernels_agree_bit_for_bit(ops, depth):
    flat = _flat(depth, 12, 20.0)
    p1, p2, p = ops.mlp_f16g1_pack(flat, depth), ops.mlp_f16g2_pack(flat, depth), ops.mlp_f16_pack(flat, depth)
    n_cu = torch.cuda.get_device_properties(0).multi_processor_count
    ko = torch.as_tensor(ops.K_ORDER, device="cuda")
    for M in (1, 513, 2 * 512 * n_cu + 77, 524288):
        fh = _rows(M, M)[:, ko].to(torch.float16).contiguous()
        a, b, c = ops.mlp_f16g1_forward(fh, p1, depth), ops.mlp_f16g2_forward(fh, p2, depth), ops.mlp_f16_forward(fh, p, depth)
        torch.cuda.synchronize()
        assert torch.equal(a, b) and torch.equal(a, c), (M, depth, float((a - b).abs().max()))


def test_fp16_operand_rounding_is_an_eighth_of_bf16s(ops):
    """same rows, same weights, three kernels: the exact fp32 kernel is the reference; the fp16 kernel's error must be that of
    11-bit operands (about 1/8 of the bf16 kernel's 8-bit ones), hidden activations and the x20 fc_alpha weights inside fp16's range"""
    flat = _flat(True, 12, 20.0)
    M = 65536
    feat = _rows(M, 5)
    ko = torch.as_tensor(ops.K_ORDER, device="cuda")
    ref = ops.mlp_f32_forward(feat.contiguous(), ops.mlp_f32_pack(flat, True), True)
    h = ops.mlp_f16_forward(feat[:, ko].to(torch.float16).contiguous(), ops.mlp_f16_pack(flat, True), True)
    b = ops.mlp_bf16_forward(feat[:, ko].to(torch.bfloat16).contiguous(), ops.mlp_bf16_pack(flat, True), True)
    torch.cuda.synchronize()
    assert torch.isfinite(h).all()
    eh, eb = float((h - ref).abs().max()), float((b - ref).abs().max())
    rh, rb = float((h - ref).norm() / ref.norm()), float((b - ref).norm() / ref.norm())
    assert rh < rb / 5.0 and eh < eb / 4.0, (eh, eb, rh, rb)
    assert rh < 6e-4, rh


def test_encoder_fp16_rows(ops):
    """the fp16 rows are the bf16 rows' values (same hardware transcendentals) rounded to 11 bits instead of 8, in the same k-order"""
    rays = torch.from_numpy(np.concatenate([x.reshape(256, -1) for x in synthetic.make_rays("blender", 256, 3)[:3]], axis=1)).cuda()
    ro, rd, rad = rays[:, :3].contiguous(), rays[:, 3:6].contiguous(), rays[:, 6:7].contiguous()
    packed = ops.pack_rays(ro, rd, rad, 2.0, 6.0)
    t = ops.sample_first_cycle(packed, torch.linspace(0, 1, 65).cuda(), None, False)
    f32 = ops.encode(packed, t, kind="fp32")
    f16 = ops.encode(packed, t, kind="fp16")
    b16 = ops.encode(packed, t, kind="bf16")
    ko = torch.as_tensor(ops.K_ORDER, device="cuda")
    assert f16.dtype == torch.float16 and f16.shape == f32.shape
    e16 = float((f16.float() - f32[:, ko]).abs().max())
    eb = float((b16.float() - f32[:, ko]).abs().max())
    # half an fp16 ulp at 1 is 2.4e-4; + 2e-6 of the hardware sin; + (round 5) up to 2e-5 of the one-fma remainder the 16-bit rows share: 2.7e-4 measured
    assert e16 <= 2.9e-4 and e16 < eb / 4, (e16, eb)


@pytest.mark.parametrize("name", fullsize_names())
def test_full_size_fp16_tier(name):
    """fp16-MFMA MLP at BASELINE sizes against the reference's fp32 outputs (tools/tier_errors.py).
    Seeded-uniform weights (configs 1-5): RGB <= 1.9e-5, depth <= 7.1e-5, acc <= 2.6e-5, weights <= 1.3e-5 -- inside the absolute
    1e-4 bar of the exact-fp32 and x3 kernels, and held to it here.
    TRAINED weights (round 5: the reference's own 3000-iteration run, blender and NDC rays): RGB <= 2.0e-4, depth <= 2.1e-4,
    weights <= 1.1e-4 -- the fp16 tier does NOT meet north_star's 1e-4 on a trained network (its 1.2x margin on seeded weights was the
    weight family, as the round-4 review suspected); it is a fast tier 10x closer to the reference than bf16, not a parity tier.  The
    parity tiers are fp32 (<= 3.6e-6 on the same fixtures) and x3 (<= 6.7e-6).  Held to 4e-4 there."""
    from test_hip_fullsize import _run
    from _cases import maxerr

    c = load_fullsize(name)
    g, st = c["g"], c["stride"]
    out = _run(c, "fp16")
    bar = 4e-4 if c["tag"] == "trained" else 1e-4
    for lvl in (0, 1):
        for k in ("rgb", "depth", "acc", "weights"):
            e = maxerr(out[lvl][k][::st].cpu().numpy(), g["o%d_%s" % (lvl, k)])
            assert e <= bar, (name, lvl, k, e)
        rgb, ref = out[lvl]["rgb"][::st].cpu().numpy(), g["o%d_rgb" % lvl]
        psnr = -10.0 * np.log10(max(float(np.mean((rgb - ref) ** 2)), 1e-20))
        assert psnr >= (80.0 if c["tag"] == "trained" else 100.0), (lvl, psnr)      # (trained, measured: 84.8 dB on the NDC rays)
