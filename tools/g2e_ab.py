#!/usr/bin/env python3
"""Interleaved A/B, ONE process, of builds of the fused bf16 kernel (tools/g2e_variant.py) against the product's two launches
(ddnerf_encode + ddnerf_mlp_bf16_forward) on BASELINE's fine pass (4096 rays x 128 samples):

    python3 tools/g2e_ab.py [--allow-differ] [n S] tools/lib/g2e_a.so tools/lib/g2e_b.so ...

per library: outputs against the unfused path's (bit for bit; exit 1 on a difference unless --allow-differ: timing-only bodies), median
launch time over 16 interleaved rounds of 30, and from the tile-loop stamps the in-kernel clock and the cycles per 512-sample tile."""
import ctypes as C
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


def timed(fn, reps=30):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    nums = [a for a in args if a.isdigit()]
    libs = [a for a in args if not a.isdigit()]
    n, S = (int(nums[0]), int(nums[1])) if len(nums) >= 2 else (4096, 128)
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
    n_cu = torch.cuda.get_device_properties(0).multi_processor_count
    st = torch.cuda.current_stream().cuda_stream
    V = C.c_void_p
    runs = [("unfused: encode + mlp", lambda: ops.mlp_bf16_forward(ops.encode(rays, t, kind="bf16"), packed, depth), None, None),
            ("unfused: mlp alone", lambda: ops.mlp_bf16_forward(feat, packed, depth), None, None),
            ("product fused", lambda: ops.encode_mlp_bf16_forward(tab, t, packed, depth), None, None)]
    want = ops.mlp_bf16_forward(feat, packed, depth)
    keep = []
    for so in libs:
        L = C.CDLL(so)
        L.ddnerf_encode_mlp_bf16_scratch_bytes.restype = C.c_size_t
        scratch = torch.empty(L.ddnerf_encode_mlp_bf16_scratch_bytes(), dtype=torch.uint8, device="cuda")
        raw = torch.empty_like(want)
        stamps = torch.zeros(n_cu * 6, dtype=torch.int64, device="cuda")
        L.ddnerf_debug_set_stamps_g2e.argtypes = [V]
        f = L.ddnerf_encode_mlp_bf16_forward
        f.argtypes = [V, V, V, C.c_int, V, C.c_int, C.c_int, V, V]
        keep.append((L, scratch, raw, stamps))

        def launch(L=L, f=f, scratch=scratch, raw=raw, stamps=stamps):
            assert L.ddnerf_debug_set_stamps_g2e(stamps.data_ptr()) == 0
            assert f(tab.data_ptr(), t.data_ptr(), packed.data_ptr(), int(depth), raw.data_ptr(), n, S, scratch.data_ptr(), st) == 0

        runs.append((os.path.basename(so), launch, raw, stamps))
    differ = False
    for name, launch, raw, _ in runs:
        if raw is None:
            continue
        launch()
        torch.cuda.synchronize()
        same = bool(((raw == want) | (torch.isnan(raw) & torch.isnan(want))).all())
        differ |= not same
        print("%-28s outputs %s" % (name, "BIT-IDENTICAL to the unfused path's" if same else "DIFFER (max |diff| %.3g)" % float((raw - want).abs().nan_to_num().max())))
    t0 = time.time()
    while time.time() - t0 < 2.0:
        for _, launch, _, _ in runs:
            for _ in range(5):
                launch()
        torch.cuda.synchronize()
    times = {name: [] for name, _, _, _ in runs}
    clocks = {name: [] for name, _, _, _ in runs}
    for rnd in range(16):
        for name, launch, raw, stamps in runs:
            times[name].append(timed(launch))
            if stamps is not None:
                s = stamps.cpu().numpy().reshape(n_cu, 6).astype(np.float64)
                s = s[s[:, 4] > 0]
                clocks[name].append((np.median((s[:, 2] - s[:, 0]) / (s[:, 3] - s[:, 1]) * 100.0), np.median((s[:, 2] - s[:, 0]) / s[:, 4])))
    base = statistics.median(times["unfused: encode + mlp"])
    for name, ts in times.items():
        med = statistics.median(ts)
        line = "%-28s median %.4f ms  min %.4f  %.4f of 2.5 PFLOP/s   vs unfused pass %+.1f %%" % (name, med, min(ts), FLOP * M / med / 1e9 / 2500, 100 * (med / base - 1))
        if clocks[name]:
            line += "   clock %.0f MHz, %.0f cycles per tile (+ prologue share)" % (statistics.median(c for c, _ in clocks[name]), statistics.median(c for _, c in clocks[name]))
        print(line)
    if differ and "--allow-differ" not in sys.argv:
        sys.exit(1)


if __name__ == "__main__":
    main()
