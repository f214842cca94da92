"""The two record formats of the training tiers (include/ddnerf_hip.h): "blocked hi/lo words" (fp32 tier, ddnerf_mlp_x3_wgrad_packed)
and "bf16 row pairs" (x3 tier, ddnerf_mlp_x3_wgrad_pairs).  A numpy restatement of the documented layouts pins the host helpers
(ops.x3_unsplit / ops.x3_unpair, CPU) and the C entry points ddnerf_mlp_x3_split / _split_pairs (GPU); the records the training
kernels write are read back through the same helpers (tests/test_hip_backward.py)."""
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


@pytest.mark.gpu
@pytest.mark.parametrize("M", [1, 17, 33, 1000 - 13, 4096 + 31])
def test_packed_weight_gradients_ignore_the_pad_columns(ops, M):
    """ddnerf_mlp_x3_wgrad_packed / _skip contract samples 0 .. M-1 only: records whose pad columns M .. ld-1 hold NaN words
    (any third-party record, or ddnerf_mlp_x3_split of a torch.empty-padded matrix) give the gradients of the M samples."""
    from ddnerf_amd import _lib

    ld = (M + 127) // 128 * 128
    g = torch.Generator(device="cuda").manual_seed(M)
    acts = torch.randn(ROWS, ld, device="cuda", generator=g)
    deltas = torch.randn(ROWS, ld, device="cuda", generator=g)
    acts_nan, deltas_nan = acts.clone(), deltas.clone()
    acts_nan[:, M:] = float("nan")
    deltas_nan[:, M:] = float("nan")
    acts[:, M:] = 0.0
    deltas[:, M:] = 0.0
    nws = _lib.lib().ddnerf_mlp_f32_wgrad_workspace_floats(M)
    ws = torch.empty(nws, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    P = lambda t: t.data_ptr()

    def run(a, d):
        ra, rd = ops.x3_split(a), ops.x3_split(d)
        out = []
        for (drow0, n_out, arow0, n_in, used) in ((256, 256, 0, 256, 256), (2304, 128, 2048, 256, 256), (512, 3, 2304, 128, 128), (0, 256, 2432, 96, 96)):
            dst = torch.full((n_out, used), 7.0, device="cuda")
            db = torch.full((n_out,), 7.0, device="cuda")
            _lib.check(_lib.lib().ddnerf_mlp_x3_wgrad_packed(P(rd), drow0, n_out, P(ra), arow0, n_in, used, M, ld, P(dst), used, 0, P(db), P(ws), 0, st), "wgrad_packed")
            out += [dst, db]
        dst = torch.full((256, 352), 7.0, device="cuda")
        db = torch.full((256,), 7.0, device="cuda")
        _lib.check(_lib.lib().ddnerf_mlp_x3_wgrad_packed_skip(P(rd), 1280, P(ra), 2432, 1024, M, ld, P(dst), P(db), P(ws), 0, st), "wgrad_packed_skip")
        torch.cuda.synchronize()
        return out + [dst, db]

    clean, dirty = run(acts, deltas), run(acts_nan, deltas_nan)
    for c, d in zip(clean, dirty):
        assert torch.isfinite(d).all()
        assert torch.equal(c, d)
    # and they are the gradients: fp64 contraction of the first M samples
    want = deltas[256:512, :M].double() @ acts[0:256, :M].double().T
    assert float((clean[0].double() - want).abs().max()) <= 1e-4 * float(want.abs().max() + 1)
    assert float((clean[1].double() - deltas[256:512, :M].double().sum(1)).abs().max()) <= 1e-4 * (1 + M ** 0.5)


def pair_record_of(x):
    """[2560, ld] fp32 -> the record of bf16 row pairs as a flat uint32 array: word ((m >> 4) * 1280 + (row >> 1)) * 16 + (m & 15) =
    bf16(x[row even][m]) | bf16(x[row odd][m]) << 16"""
    ld = x.shape[1]
    t = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).to(torch.bfloat16).view(torch.int16).numpy().astype(np.uint16).astype(np.uint32)
    w = t[0::2] | (t[1::2] << 16)
    rec = np.zeros(ROWS // 2 * ld, dtype=np.uint32)
    p, m = np.meshgrid(np.arange(ROWS // 2), np.arange(ld), indexing="ij")
    rec[((m >> 4) * (ROWS // 2) + p) * 16 + (m & 15)] = w
    return rec


def test_unpair_inverts_the_documented_layout():
    from ddnerf_amd import ops

    rng = np.random.default_rng(1)
    x = (rng.standard_normal((ROWS, 48)) * np.exp(rng.uniform(-20, 20, (ROWS, 48)))).astype(np.float32)
    rec = torch.from_numpy(pair_record_of(x).view(np.float32).reshape(ROWS // 2, 48))
    back = ops.x3_unpair(rec)
    assert torch.equal(back, torch.from_numpy(x).bfloat16().float())


@pytest.mark.gpu
def test_split_pairs_entry_point_writes_the_documented_layout(ops):
    g = torch.Generator(device="cuda").manual_seed(4)
    x = torch.randn(ROWS, 160, device="cuda", generator=g) * torch.exp(torch.rand(ROWS, 160, device="cuda", generator=g) * 30 - 15)
    rec = ops.x3_split_pairs(x)
    assert rec.shape == (ROWS // 2, 160)
    assert np.array_equal(rec.cpu().numpy().view(np.uint32).reshape(-1), pair_record_of(x.cpu().numpy()))


@pytest.mark.gpu
@pytest.mark.parametrize("M", [1, 17, 33, 1000 - 13, 4096 + 31])
def test_pair_weight_gradients_ignore_the_pad_columns(ops, M):
    """the same guarantee for the x3 tier's kernel (records of bf16 row pairs): only samples 0 .. M-1 are contracted"""
    from ddnerf_amd import _lib

    ld = (M + 127) // 128 * 128
    g = torch.Generator(device="cuda").manual_seed(100 + M)
    acts = torch.randn(ROWS, ld, device="cuda", generator=g)
    deltas = torch.randn(ROWS, ld, device="cuda", generator=g)
    acts_nan, deltas_nan = acts.clone(), deltas.clone()
    acts_nan[:, M:] = float("nan")
    deltas_nan[:, M:] = float("nan")
    acts[:, M:] = 0.0
    deltas[:, M:] = 0.0
    ws = torch.empty(_lib.lib().ddnerf_mlp_f32_wgrad_workspace_floats(M), device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    P = lambda t: t.data_ptr()

    def run(a, d):
        ra, rd = ops.x3_split_pairs(a), ops.x3_split_pairs(d)
        out = []
        for (drow0, n_out, arow0, n_in, used) in ((256, 256, 0, 256, 256), (2304, 160, 2048, 256, 256), (2432, 6, 2304, 128, 128), (0, 256, 2432, 96, 96),
                                                  (2304, 128, 2528, 32, 27)):
            dst = torch.full((n_out, used), 7.0, device="cuda")
            db = torch.full((n_out,), 7.0, device="cuda")
            _lib.check(_lib.lib().ddnerf_mlp_x3_wgrad_pairs(P(rd), drow0, n_out, P(ra), arow0, n_in, used, M, ld, P(dst), used, 0, P(db), P(ws), 0, st), "wgrad_pairs")
            out += [dst, db]
        dst = torch.full((256, 352), 7.0, device="cuda")
        db = torch.full((256,), 7.0, device="cuda")
        _lib.check(_lib.lib().ddnerf_mlp_x3_wgrad_pairs_skip(P(rd), 1280, P(ra), 2432, 1024, M, ld, P(dst), P(db), P(ws), 0, st), "wgrad_pairs_skip")
        torch.cuda.synchronize()
        return out + [dst, db]

    clean, dirty = run(acts, deltas), run(acts_nan, deltas_nan)
    for c, d in zip(clean, dirty):
        assert torch.isfinite(d).all() and torch.equal(c, d)
    dq, aq = deltas.bfloat16().double(), acts.bfloat16().double()
    want = dq[256:512, :M] @ aq[0:256, :M].T                          # exact on the bf16-rounded operands (fp32 accumulation)
    assert float((clean[0].double() - want).abs().max()) <= 1e-5 * float(want.abs().max() + 1)
    assert float((clean[1].double() - dq[256:512, :M].sum(1)).abs().max()) <= 1e-5 * (1 + M ** 0.5)
    want = dq[2304:2464, :M] @ aq[2048:2304, :M].T                     # the 160-row job that feeds layers_dir.0 and fc_alpha
    assert float((clean[2].double() - want).abs().max()) <= 1e-5 * float(want.abs().max() + 1)
    want = dq[2304:2432, :M] @ aq[2528:2555, :M].T                     # the ragged 27-of-32-column job
    assert float((clean[8].double() - want).abs().max()) <= 1e-5 * float(want.abs().max() + 1)
    # the odd-row job offsets a record row cannot express are refused
    assert _lib.lib().ddnerf_mlp_x3_wgrad_pairs(P(acts), 1, 2, P(acts), 0, 32, 32, M, ld, P(ws), 32, 0, None, P(ws), 0, st) == -2
