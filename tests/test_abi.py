"""CPU-side checks of the boundary: the C-ABI library loads without a GPU and exports exactly the symbols
include/ddnerf_hip.h declares; argument validation happens on the host before any launch."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "ddnerf_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(ddnerf_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from ddnerf_amd import _lib

    lib = _lib.lib()
    names = declared_symbols()
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), "libddnerf_hip.so does not export %s" % n
    assert sorted(_lib._SIGS) == names, "ctypes signature table and header disagree"
    assert lib.ddnerf_abi_version() >= 1


def test_argument_validation_without_gpu():
    from ddnerf_amd import _lib

    lib = _lib.lib()
    assert lib.ddnerf_pack_rays(None, None, None, 2.0, 6.0, None, 16, None) == -1      # DDNERF_E_ARG
    assert lib.ddnerf_encode(ctypes.c_void_p(16), ctypes.c_void_p(16), ctypes.c_void_p(16), 4, 4, 7, 0, None) == -2
    assert lib.ddnerf_encode(ctypes.c_void_p(16), ctypes.c_void_p(16), ctypes.c_void_p(20), 4, 4, 0, 0, None) == -3
    assert lib.ddnerf_sample_pdf(ctypes.c_void_p(16), ctypes.c_void_p(16), ctypes.c_void_p(16), None,
                                 ctypes.c_void_p(16), 4, 1, 9, 1, None) == -1                 # nc == 1 rejected
    assert b"aligned" in lib.ddnerf_error_string(-3)
    assert lib.ddnerf_mlp_f32_packed_floats(0) > 600000 and lib.ddnerf_mlp_bf16_packed_bytes(1) > 1200000


def test_ops_refuse_cpu_tensors():
    import torch

    from ddnerf_amd import _lib, ops

    with pytest.raises(_lib.DDNerfHipError):
        ops.pack_rays(torch.zeros(4, 3), torch.ones(4, 3), torch.ones(4, 1), 2.0, 6.0)


def test_model_surface_and_checkpoint_keys():
    """constructor / attribute surface of the drop-in models and state-dict compatibility"""
    import torch

    from ddnerf_amd import synthetic
    from ddnerf_amd.cfgnode import CfgNode
    from models import models

    cfg = CfgNode.load(os.path.join(ROOT, "configs", "config_blender.yml"))
    m = getattr(models, cfg.nerf.type)(cfg)
    assert m.coarse is not m.fine and m.cfg is cfg
    assert [k for k in m.coarse.state_dict()] == [n + s for n, _, _ in synthetic.layer_table(True) for s in (".weight", ".bias")]
    assert sum(p.numel() for p in m.coarse.parameters()) == 612998 and sum(p.numel() for p in m.fine.parameters()) == 612740
    sd = {k: torch.from_numpy(v) for k, v in synthetic.make_state_dict(True, 3).items()}
    m.load_weights_from_checkpoint({"model_1_state_dict": sd, "model_2_state_dict": m.fine.state_dict()})
    flat = m.coarse.flat_params()
    assert flat.shape == (612998,) and torch.equal(flat[:24576].view(256, 96), sd["layers_xyz.0.weight"])
    # optimiser updates write through to the flat buffer (parameters are views of it)
    opt = torch.optim.Adam(m.coarse.parameters(), lr=1e-3)
    for p in m.coarse.parameters():
        p.grad = torch.ones_like(p)
    before = flat.clone()
    opt.step()
    assert not torch.equal(m.coarse.flat_params(), before) and m.coarse.flat_params().data_ptr() == flat.data_ptr()
    cfg2 = CfgNode.load(os.path.join(ROOT, "configs", "config_blender_mipnerf.yml"))
    g = getattr(models, cfg2.nerf.type)(cfg2)
    assert g.fine is g.coarse and g.coarse.depth_head is False
    for name in ("run_iter", "to", "train", "eval", "load_weights_from_checkpoint", "predict", "run_network",
                 "get_rays_batches"):
        assert callable(getattr(m, name))


def test_packed_weight_cache_follows_parameter_updates():
    """Regression: the parameters alias the flat buffer through `.data`, so an optimizer step bumps THEIR version
    counters and not the buffer's -- the kernel-format weight cache must be keyed on the former, or training would
    keep evaluating the initial weights."""
    import torch
    from ddnerf_amd import base_architectures as ba, functions as F

    net = ba.MipNeRFModel(include_input_dir=True)
    calls = []

    def builder(flat, depth_head):
        calls.append(float(flat.sum()))
        return len(calls)

    assert F._cached_pack(net, "x", builder) == 1 and F._cached_pack(net, "x", builder) == 1
    opt = torch.optim.Adam(net.parameters(), lr=0.1)
    for p in net.parameters():
        p.grad = torch.ones_like(p)
    opt.step()
    assert F._cached_pack(net, "x", builder) == 2 and calls[1] != calls[0]      # repacked from the updated buffer
    sd = {k: torch.zeros_like(v) for k, v in net.state_dict().items()}
    net.load_state_dict(sd)
    assert F._cached_pack(net, "x", builder) == 3 and calls[2] == 0.0
    with torch.no_grad():
        net.flat_params().add_(1.0)          # direct write (what a broadcast does) needs the explicit invalidation
    net.invalidate_packed()
    assert F._cached_pack(net, "x", builder) == 4 and calls[3] == float(net.flat_params().numel())


def test_the_frozen_fork_of_the_16_bit_mfma_body_has_not_drifted():
    """mlp_mfma16_hilo.inc is a frozen fork of mlp_mfma16.inc (the x3 tier's strict mode, DDNERF_X3_WGRAD=exact: records of hi/lo words
    instead of bf16 row pairs).  A fix to the shared body has to be made in BOTH files or the strict mode silently diverges: this test
    pins the set of lines in which the two files differ (tests/golden/mfma16_fork.diff).  When it fails, either port the change to the
    other file, or -- if the difference is a deliberate, record-format-specific one -- regenerate the pinned diff (the command is in the
    assertion message) and say so in the commit."""
    got = _fork_diff()
    want = [ln.rstrip("\n") for ln in open(os.path.join(ROOT, "tests", "golden", "mfma16_fork.diff"))]
    want = [ln[:2] + ln[2:].strip() for ln in want]
    assert sorted(got) == sorted(want), ("the two forks' differences changed: %d lines now, %d pinned; port the change, or regenerate with "
                                         "`python tests/test_abi.py --pin-fork-diff`" % (len(got), len(want)))


def _fork_diff():
    import difflib

    csrc = os.path.join(ROOT, "ddnerf_amd", "csrc")
    a = [ln.rstrip() for ln in open(os.path.join(csrc, "mlp_mfma16.inc"))]
    b = [ln.rstrip() for ln in open(os.path.join(csrc, "mlp_mfma16_hilo.inc"))]
    return [("< " if ln[0] == "-" else "> ") + ln[1:].strip() for ln in difflib.unified_diff(a, b, lineterm="", n=0)
            if ln[:1] in "+-" and not ln.startswith(("+++", "---"))]


if __name__ == "__main__" and "--pin-fork-diff" in __import__("sys").argv:
    open(os.path.join(ROOT, "tests", "golden", "mfma16_fork.diff"), "w").write("\n".join(_fork_diff()) + "\n")
