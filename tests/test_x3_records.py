"""The record format of the x3 training tier ("blocked hi/lo words", include/ddnerf_hip.h: ddnerf_mlp_x3_wgrad_packed): a numpy
restatement of the documented layout pins the host helper (ops.x3_unsplit, CPU) and the C entry point ddnerf_mlp_x3_split (GPU),
and the records the training kernels write are read back through the same helper (tests/test_hip_backward.py)."""
import numpy as np
import pytest
import torch

ROWS = 2560


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available()
    from ddnerf_amd import ops as _ops
    return _ops


def words_of(x):
    """fp32 array -> uint32 words (bf16(x) << 16) | bf16(x - bf16(x)), round-to-nearest-even conversions"""
    t = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32))
    hi = t.to(torch.bfloat16)
    lo = (t - hi.to(torch.float32)).to(torch.bfloat16)
    h = hi.view(torch.int16).numpy().astype(np.uint16).astype(np.uint32)
    l = lo.view(torch.int16).numpy().astype(np.uint16).astype(np.uint32)
    return (h << 16) | l


def record_of(x):
    """[2560, ld] fp32 ([feature][sample]) -> the record as a flat uint32 array: word ((m >> 4) * 2560 + row) * 16 + (m & 15)"""
    ld = x.shape[1]
    w = words_of(x)
    rec = np.zeros(ROWS * ld, dtype=np.uint32)
    rows, m = np.meshgrid(np.arange(ROWS), np.arange(ld), indexing="ij")
    rec[((m >> 4) * ROWS + rows) * 16 + (m & 15)] = w
    return rec


def test_unsplit_inverts_the_documented_layout():
    from ddnerf_amd import ops

    rng = np.random.default_rng(0)
    x = (rng.standard_normal((ROWS, 64)) * np.exp(rng.uniform(-20, 20, (ROWS, 64)))).astype(np.float32)
    x[5, 7] = 0.0
    rec = torch.from_numpy(record_of(x).view(np.float32).reshape(ROWS, 64))
    back = ops.x3_unsplit(rec).numpy()
    assert np.all(np.abs(back - x) <= 2.0 ** -16 * np.abs(x))      # hi + lo carries 16+ significant bits
    assert back[5, 7] == 0.0
    # the split is exact where 16 bits suffice
    y = np.zeros((ROWS, 16), dtype=np.float32)
    y[:, 3] = np.arange(ROWS, dtype=np.float32)
    assert np.array_equal(ops.x3_unsplit(torch.from_numpy(record_of(y).view(np.float32).reshape(ROWS, 16))).numpy(), y)


@pytest.mark.gpu
def test_split_entry_point_writes_the_documented_layout(ops):
    g = torch.Generator(device="cuda").manual_seed(3)
    x = torch.randn(ROWS, 160, device="cuda", generator=g) * torch.exp(torch.rand(ROWS, 160, device="cuda", generator=g) * 30 - 15)
    rec = ops.x3_split(x)
    want = record_of(x.cpu().numpy())
    assert np.array_equal(rec.cpu().numpy().view(np.uint32).reshape(-1), want)
