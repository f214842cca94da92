"""The two bf16 MLP kernels behind ddnerf_mlp_bf16_forward -- one 64-sample group per wave (mlp_bf16.hip) and two groups per weight pass
(mlp_bf16_g2.hip, tile body generated as assembly) -- run the same arithmetic in the same order: their outputs are compared BIT FOR
BIT, on ragged sizes, one tile, several tiles per workgroup (the steady-state path of the persistent loop: its memory-counter
bookkeeping, the features parked for the next tile) and both heads.  The entry point that picks between them by launch size must
therefore give results that do not depend on the size of the launch a sample travels in."""
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


def _rows(ops, M, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    feat = torch.zeros(M, 128, device="cuda")
    feat[:, :123] = torch.rand(M, 123, device="cuda", generator=g) * 2 - 1
    return feat[:, torch.as_tensor(ops.K_ORDER, device="cuda")].to(torch.bfloat16).contiguous()


@pytest.mark.parametrize("depth", [False, True])
def test_two_group_kernel_bit_identical_to_one_group(ops, depth):
    flat = _flat(depth, 12, 20.0)      # (sharpened weights: activations of every magnitude, ReLU zeros in every layer)
    p1, p2 = ops.mlp_bf16g1_pack(flat, depth), ops.mlp_bf16g2_pack(flat, depth)
    n_cu = torch.cuda.get_device_properties(0).multi_processor_count
    # 1 / 37: one ragged tile; 512, 513: the tile boundary; 3 tiles on 3 workgroups; every workgroup 2 tiles + a ragged one;
    # BASELINE's fine pass (4096 x 128: four tiles per workgroup on 256 CUs)
    for M in (1, 37, 512, 513, 3 * 512 - 5, 2 * 512 * n_cu + 77, 524288):
        fb = _rows(ops, M, M)
        a, b = ops.mlp_bf16g1_forward(fb, p1, depth), ops.mlp_bf16g2_forward(fb, p2, depth)
        torch.cuda.synchronize()
        assert a.shape == b.shape == (M, 6 if depth else 4)
        assert torch.equal(a, b), (M, depth, int((a != b).any(dim=1).sum()), float((a - b).abs().max()))


def test_two_group_kernel_is_deterministic_and_leaves_its_neighbours_alone(ops):
    """the outputs go through a bounds-checked buffer: nothing is written past row M - 1, nothing depends on what follows the rows"""
    flat = _flat(True, 5, 4.0)
    p2 = ops.mlp_bf16g2_pack(flat, True)
    M = 70001
    fb_big = _rows(ops, M + 600, 3)
    fb = fb_big[:M].contiguous()
    a = ops.mlp_bf16g2_forward(fb, p2, True)
    b = ops.mlp_bf16g2_forward(fb, p2, True)
    c = ops.mlp_bf16g2_forward(fb_big, p2, True)[:M]     # the same rows inside a longer launch (other tile count, other neighbours)
    torch.cuda.synchronize()
    assert torch.equal(a, b) and torch.equal(a, c)


def test_entry_point_picks_by_size_and_the_choice_is_invisible(ops, monkeypatch):
    """ddnerf_mlp_bf16_forward: one weight image for both kernels; a sample's result does not depend on the launch it is part of"""
    flat = _flat(False, 9, 1.0)
    packed = ops.mlp_bf16_pack(flat, False)
    p1 = ops.mlp_bf16g1_pack(flat, False)
    assert packed.numel() > p1.numel() + 1_000_000        # (both images)
    fb = _rows(ops, 100000, 7)
    big = ops.mlp_bf16_forward(fb, packed, False)                      # 196 tiles of 512 on 256 CUs: the two-group kernel
    small = torch.cat([ops.mlp_bf16_forward(fb[i:i + 20000].contiguous(), packed, False) for i in range(0, 100000, 20000)])
    ref = ops.mlp_bf16g1_forward(fb, p1, False)
    torch.cuda.synchronize()
    assert torch.equal(big, small) and torch.equal(big, ref)


def test_many_launches_back_to_back_stay_identical(ops):
    """A race in the assembly kernel's bookkeeping (a wait one too lax, a barrier too few) would be rare and timing-dependent: 120
    launches back to back on fresh rows, each compared bit for bit with the one-group kernel (scratch/g2_stress.py runs the same
    loop for minutes: 35,000 launches without a mismatch on the final build)."""
    flat = _flat(False, 12, 20.0)
    p1, p2 = ops.mlp_bf16g1_pack(flat, False), ops.mlp_bf16g2_pack(flat, False)
    for it in range(40):
        M = (524288, 262144, 400000 + 977 * it)[it % 3]
        fb = _rows(ops, M, 1000 + it)
        a = ops.mlp_bf16g1_forward(fb, p1, False)
        outs = [ops.mlp_bf16g2_forward(fb, p2, False) for _ in range(3)]
        torch.cuda.synchronize()
        assert all(torch.equal(a, b) for b in outs), (it, M)


def test_launches_beyond_the_32_bit_row_offsets_are_split(ops):
    """the assembly body addresses rows with 32-bit byte offsets: the entry point cuts a launch into pieces of 4 M samples (each with
    its own first tile and ragged last tile); 4 M + 513 samples, depth head (6 outputs per sample: the other output stride)"""
    flat = _flat(True, 3, 4.0)
    p1, p2 = ops.mlp_bf16g1_pack(flat, True), ops.mlp_bf16g2_pack(flat, True)
    M = (1 << 22) + 513
    fb = _rows(ops, M, 11)
    a, b = ops.mlp_bf16g1_forward(fb, p1, True), ops.mlp_bf16g2_forward(fb, p2, True)
    torch.cuda.synchronize()
    assert torch.equal(a, b) and bool(torch.isfinite(b).all())
