#!/usr/bin/env python3
"""Interleaved A/B, ONE process, of the two ways the bf16 tier runs run_network (models/models.py:117-142) on a pass of n rays x S samples:
  A  ddnerf_encode(bf16 rows) + ddnerf_mlp_bf16_forward      two launches, the rows through HBM
  B  ddnerf_encode_mlp_bf16_forward                          one launch (mlp_bf16_g2e.hip: the encoder inside the MLP kernel)
Outputs are compared bit for bit first (exit 1 when they differ), then 2 s of ramp and 16 rounds of 30 passes of each in turn, HIP
events around the passes and, in separate rounds, around every launch.  python3 tools/fused_ab.py [n] [S]"""
import os
import statistics
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ddnerf_amd import ops, synthetic  # noqa: E402

FLOP = 1220608


def timed(fn, reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    S = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    depth = S == 64
    M = n * S
    sd = synthetic.make_state_dict(depth, 12, 20.0)
    names = [k for k, _, _ in synthetic.layer_table(depth)]
    flat = torch.from_numpy(np.concatenate([np.concatenate([sd[k + ".weight"].ravel(), sd[k + ".bias"].ravel()]) for k in names])).cuda()
    packed = ops.mlp_bf16_pack(flat, depth)
    o, d, rad, _ = synthetic.make_rays("blender", n, 1)
    rays = ops.pack_rays(*(torch.from_numpy(x).cuda() for x in (o, d, rad)), 2.0, 6.0)
    torch.manual_seed(0)
    t = (2.0 + 4.0 * torch.sort(torch.rand(n, S + 1, device="cuda"), dim=1).values).contiguous()
    tab = ops.ray_table(rays)
    feat = ops.encode(rays, t, kind="bf16")

    def a_pass():
        return ops.mlp_bf16_forward(ops.encode(rays, t, kind="bf16"), packed, depth)

    def b_pass():
        return ops.encode_mlp_bf16_forward(tab, t, packed, depth)

    parts = {"A encode": lambda: ops.encode(rays, t, kind="bf16"), "A mlp": lambda: ops.mlp_bf16_forward(feat, packed, depth),
             "B fused": b_pass, "B ray table": lambda: ops.ray_table(rays)}
    ra, rb = a_pass(), b_pass()
    torch.cuda.synchronize()
    same = bool(((ra == rb) | (torch.isnan(ra) & torch.isnan(rb))).all())
    print("n %d  S %d  M %d  depth head %s: outputs %s" % (n, S, M, depth, "BIT-IDENTICAL" if same else "DIFFER (max |diff| %.3g)" % float((ra - rb).abs().max())))
    t0 = time.time()
    while time.time() - t0 < 2.0:
        for _ in range(10):
            a_pass()
            b_pass()
        torch.cuda.synchronize()
    tp = {"A encode + mlp (pass)": [], "B fused (pass)": []}
    tk = {k: [] for k in parts}
    for rnd in range(16):
        tp["A encode + mlp (pass)"].append(timed(a_pass, 30))
        tp["B fused (pass)"].append(timed(b_pass, 30))
        for k, fn in parts.items():
            tk[k].append(timed(fn, 30))
    for k, ts in list(tp.items()) + list(tk.items()):
        med = statistics.median(ts)
        line = "%-26s median %.4f ms  min %.4f" % (k, med, min(ts))
        if "mlp" in k or "fused" in k:
            line += "   %.4f of 2.5 PFLOP/s (algorithmic MLP FLOP)" % (FLOP * M / med / 1e9 / 2500)
        print(line)
    a, b = statistics.median(tp["A encode + mlp (pass)"]), statistics.median(tp["B fused (pass)"])
    print("pass: fused / unfused = %.4f  (%+.1f %%)" % (b / a, 100 * (b / a - 1)))
    if not same:
        sys.exit(1)


if __name__ == "__main__":
    main()
