#!/usr/bin/env python3
"""The bare-MFMA-loop ceilings of csrc/mfma_ceiling.hip (diagnostic library), all five modes interleaved on one device:
0 registers, 1 + A fragments from LDS (one ds_read_b128 per 4 MFMAs), 2 + weight staging, 3 / 4 = 1 / 2 with half the fragment reads."""
import ctypes as C
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ddnerf_amd import build as hip_build  # noqa: E402

L = C.CDLL(hip_build.DIAG_SO)
V = C.c_void_p
L.ddnerf_debug_mfma_ceiling_src_bytes.restype = C.c_size_t
f = L.ddnerf_debug_mfma_ceiling
f.argtypes = [C.c_int, V, V, C.c_int, V, V]
dev = torch.device("cuda")
n_cu = torch.cuda.get_device_properties(dev).multi_processor_count
src = (torch.rand(L.ddnerf_debug_mfma_ceiling_src_bytes() // 2, device=dev) * 2 - 1).to(torch.bfloat16).contiguous()
out = torch.empty(n_cu * 256, device=dev)
stamps = torch.zeros(n_cu * 2, dtype=torch.int64, device=dev)
iters = 20000
flop = n_cu * 4 * iters * 96 * (16 * 16 * 32 * 2)
names = ["registers", "lds_fed", "lds_fed_staged", "lds_fed_half", "lds_fed_half_staged"]
res = {m: [] for m in range(5)}
for rep in range(5):
    for m in range(5):
        t0 = time.time()
        while time.time() - t0 < 0.4:
            for _ in range(4):
                assert f(m, src.data_ptr(), out.data_ptr(), iters, stamps.data_ptr(), None) == 0
            torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(4):
            f(m, src.data_ptr(), out.data_ptr(), iters, stamps.data_ptr(), None)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 4
        s = stamps.cpu().numpy().reshape(n_cu, 2).astype(float)
        res[m].append((flop / (ms * 1e-3) / 2.5e15, (s[:, 0] / s[:, 1]).mean() * 100))
for m in range(5):
    fr = sorted(x[0] for x in res[m])[2]
    print("mode %d %-20s frac of 2.5 PFLOP/s %.4f (median of 5: %s), in-kernel clock %.0f MHz" % (m, names[m], fr, " ".join("%.4f" % x[0] for x in res[m]), res[m][2][1]))
