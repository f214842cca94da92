#!/usr/bin/env python3
"""In-kernel clock, cycles per 512-sample tile and (when the build has them) per-period / per-block / SGPR stamps of a diagnostic build
of the two-group bf16 kernel.  GPU box: `python3 tools/g2_clock.py tools/lib/g2_<name>.so [seconds] [--pack-g1]`.

Prints one summary line per library (launch time of a 524,288-sample launch, fraction of the 2.5 PFLOP/s bf16 peak, in-kernel clock =
d(s_memtime) / d(s_memrealtime) x 100 MHz, cycles per tile against the 154,240 the tile's 9,640 MFMAs per wave need)."""
import ctypes as C
import importlib.util
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ddnerf_amd import synthetic  # noqa: E402


def load_gen():
    spec = importlib.util.spec_from_file_location("gen", os.path.join(ROOT, "ddnerf_amd", "csrc", "gen_bf16_g2.py"))
    argv, sys.argv = sys.argv, ["gen"]
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    sys.argv = argv
    return gen


def setup(so, M=524288, zeros=False):
    sd = synthetic.make_state_dict(False, 12, 20.0)
    names = [n for n, _, _ in synthetic.layer_table(False)]
    flat = torch.from_numpy(np.concatenate([np.concatenate([sd[k + ".weight"].ravel(), sd[k + ".bias"].ravel()]) for k in names])).cuda()
    fb = (torch.rand(M, 128, device="cuda") * 2 - 1).to(torch.bfloat16).contiguous()
    if zeros:
        fb.zero_()
    raw = torch.empty(M, 4, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    V = C.c_void_p
    L = C.CDLL(so)
    L.ddnerf_mlp_bf16g2_packed_bytes.restype = C.c_size_t
    packed = torch.empty(L.ddnerf_mlp_bf16g2_packed_bytes(0), dtype=torch.uint8, device="cuda")
    L.ddnerf_mlp_bf16g2_pack.argtypes = [V, C.c_int, V, V]
    assert L.ddnerf_mlp_bf16g2_pack(flat.data_ptr(), 0, packed.data_ptr(), st) == 0
    f = L.ddnerf_mlp_bf16g2_forward
    f.argtypes = [V, V, C.c_int, V, C.c_long, V]
    keep = (flat, fb, raw, packed)
    return L, (lambda: f(fb.data_ptr(), packed.data_ptr(), 0, raw.data_ptr(), M, st)), keep


def main():
    so = sys.argv[1]
    seconds = float(sys.argv[2]) if len(sys.argv) > 2 and not sys.argv[2].startswith("-") else 2.5
    M = 524288
    L, launch, keep = setup(so, M)
    n_cu = torch.cuda.get_device_properties(0).multi_processor_count
    stamps = torch.zeros(n_cu * (6 + 192), dtype=torch.int64, device="cuda")   # (six values per workgroup, then 192 per workgroup: the grid is one workgroup per CU)
    has_stamps = hasattr(L, "ddnerf_debug_set_stamps_g2")
    if has_stamps:
        L.ddnerf_debug_set_stamps_g2.argtypes = [C.c_void_p]
        assert L.ddnerf_debug_set_stamps_g2(stamps.data_ptr()) == 0
    t0 = time.time()
    n = 0
    while time.time() - t0 < seconds:
        for _ in range(50):
            launch()
        torch.cuda.synchronize()
        n += 50
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        launch()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 50
    name = os.path.basename(so)
    if not has_stamps:
        print("%s: launch %.4f ms (%.4f of peak)" % (name, ms, 1220608 * M / ms / 1e9 / 2500))
        return
    allst = stamps.cpu().numpy()
    s = allst[:n_cu * 6].reshape(n_cu, 6).astype(np.float64)
    clk = (s[:, 2] - s[:, 0]) / (s[:, 3] - s[:, 1]) * 100.0
    cyc = (s[:, 2] - s[:, 0]) / s[:, 4]
    print("%s: launch %.4f ms (%.4f of peak); clock median %.0f MHz; %.0f cycles per 512-sample tile (ideal 154240 -> pipe %.1f %% busy); loop %.1f us"
          % (name, ms, 1220608 * M / ms / 1e9 / 2500, np.median(clk), np.median(cyc), 100 * 154240 / np.median(cyc), np.median(s[:, 3] - s[:, 1]) / 100), flush=True)
    gen = load_gen()
    allp = allst[n_cu * 6:].reshape(n_cu, 192).astype(np.float64)
    if allp[:, 1:gen.NPER + 1].any():
        ps = allp[:, :gen.NPER + 1]
        med = np.median(np.diff(ps, axis=1), axis=0)
        tot_ideal = tot = 0
        passes = {}
        for p_, (pi, ci) in enumerate(gen.PERIODS):
            ideal = sum(gen.K[l] // 32 for l, b in gen.CHUNKS[ci]) * 64
            passes.setdefault(gen.PASSES[pi], []).append((med[p_], ideal))
        print("pass     periods: cycles (lost against 16 cycles per MFMA)")
        for (l, g), lst in passes.items():
            print("L%d g%d  %6.0f lost %5.0f | %s" % (l, g, sum(x for x, _ in lst), sum(x - y for x, y in lst), "  ".join("%5.0f(%+5.0f)" % (x, x - y) for x, y in lst)))
            tot += sum(x for x, _ in lst)
            tot_ideal += sum(y for _, y in lst)
        print("sum of periods %.0f (ideal %d); tile boundary (per-tile cycles - periods) %.0f" % (tot, tot_ideal, np.median(cyc) - tot))
        lo, hi = gen.STAMP_BLOCKS
        j = gen.NPER + 1
        for p_ in range(lo, hi):
            nb = len(gen.CHUNKS[gen.PERIODS[p_][1]])
            t = np.concatenate([ps[:, p_:p_ + 1], allp[:, j:j + nb], ps[:, p_ + 1:p_ + 2]], axis=1)
            dd = np.median(np.diff(t, axis=1), axis=0)
            print("period %2d %s: blocks %s | barrier+tail %4.0f" % (p_, gen.PASSES[gen.PERIODS[p_][0]], " ".join("%5.0f" % x for x in dd[:-1]), dd[-1]))
            j += nb
    sg = allp[:, 150:150 + gen.NSTAMP_SGPR]
    if sg.any():
        k = int((np.median(sg, axis=0) > 0).sum())
        d = np.diff(sg[:, :k], axis=1)
        print("sgpr stamps (%d block boundaries): cycles per block, median over workgroups: %s" % (k, " ".join("%5.0f" % x for x in np.median(d, axis=0))))
        print("   10th / 90th percentile: %s / %s" % (" ".join("%5.0f" % x for x in np.percentile(d, 10, axis=0)), " ".join("%5.0f" % x for x in np.percentile(d, 90, axis=0))))


if __name__ == "__main__":
    main()
