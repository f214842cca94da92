#!/usr/bin/env python3
"""What do the serial chains of the wave-per-ray compositing kernel cost?  (The round-4 review proposed moving them to a lane-per-ray phase.)
`build`: tools/lib/comp_<name>.so = composite.hip + samplers.hip with one ingredient compiled out (TIMING ONLY: wrong results);
`run` (GPU box): the fine pass's launch (4096 rays x 128 samples, ddnerf_composite_forward) of every build, interleaved, HIP events."""
import ctypes as C
import os
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "ddnerf_amd", "csrc")
LIB = os.path.join(ROOT, "tools", "lib")
VARIANTS = {"base": [], "nochain": ["-DCOMP_EXP_NOCHAIN"], "norgbsum": ["-DCOMP_EXP_NORGBSUM"], "neither": ["-DCOMP_EXP_NOCHAIN", "-DCOMP_EXP_NORGBSUM"]}


def build():
    os.makedirs(LIB, exist_ok=True)
    for name, defs in VARIANTS.items():
        so = os.path.join(LIB, "comp_%s.so" % name)
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-Wno-unused-function",
                               "-ffp-contract=off"] + defs + ["-shared", os.path.join(CSRC, "composite.hip"), "-o", so])
        print(so)


def run():
    import torch

    n, S = 4096, 128
    torch.manual_seed(0)
    raw = torch.randn(n, S, 4, device="cuda")
    t = (2.0 + 4.0 * torch.sort(torch.rand(n, S + 1, device="cuda"), dim=1).values).contiguous()
    rays = torch.randn(n, 12, device="cuda")
    outs = [torch.empty(n, 3, device="cuda")] + [torch.empty(n, device="cuda") for _ in range(2)] + [torch.empty(n, S, device="cuda"), torch.empty(n, device="cuda")]
    V = C.c_void_p
    st = torch.cuda.current_stream().cuda_stream
    launches = {}
    for name in VARIANTS:
        L = C.CDLL(os.path.join(LIB, "comp_%s.so" % name))
        f = L.ddnerf_composite_forward
        f.argtypes = [V, C.c_int, V, V, V, V, C.c_int, C.c_int, C.c_int] + [V] * 7 + [V]
        launches[name] = (L, lambda f=f: f(raw.data_ptr(), 4, t.data_ptr(), rays.data_ptr(), None, None, n, S, 2, outs[0].data_ptr(), outs[1].data_ptr(),
                                             outs[2].data_ptr(), outs[3].data_ptr(), outs[4].data_ptr(), None, None, st))
    times = {k: [] for k in launches}
    for rnd in range(12):
        for name, (_, go) in launches.items():
            for _ in range(5):
                go()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50):
                go()
            e1.record()
            torch.cuda.synchronize()
            times[name].append(e0.elapsed_time(e1) / 50 * 1e3)
    for name, ts in times.items():
        print("%-10s %.2f us per launch (median of 12 rounds of 50 back-to-back launches; min %.2f)" % (name, statistics.median(ts), min(ts)))


if __name__ == "__main__":
    build() if sys.argv[1] == "build" else run()
