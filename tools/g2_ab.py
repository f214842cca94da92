#!/usr/bin/env python3
"""Interleaved A/B of builds of the two-group bf16 kernel in ONE process (devices and clock states differ between runs):
python3 tools/g2_ab.py lib1.so lib2.so ...   -- 2 s of ramp, then 16 rounds of 30 launches per library in turn; outputs compared with
the first library's."""
import os
import statistics
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import g2_clock  # noqa: E402


def main():
    M = 524288
    runs = []
    for so in [a for a in sys.argv[1:] if not a.startswith("--")]:
        torch.manual_seed(0)                      # (every library gets the same feature rows)
        L, launch, keep = g2_clock.setup(so, M)
        runs.append((os.path.basename(so), launch, keep))
    outs = []
    for name, launch, keep in runs:
        launch()
        torch.cuda.synchronize()
        outs.append(keep[2].clone())
    differ = False
    for (name, _, _), o in zip(runs[1:], outs[1:]):
        print("%s: outputs %s the first library's (max |diff| %.3g)" % (name, "EQUAL" if torch.equal(o, outs[0]) else "DIFFER from", float((o - outs[0]).abs().max())))
        differ |= not torch.equal(o, outs[0])
    t0 = time.time()
    while time.time() - t0 < 2.0:
        for _, launch, _ in runs:
            for _ in range(10):
                launch()
        torch.cuda.synchronize()
    times = {name: [] for name, _, _ in runs}
    for rnd in range(16):
        for name, launch, _ in runs:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(30):
                launch()
            e1.record()
            torch.cuda.synchronize()
            times[name].append(e0.elapsed_time(e1) / 30)
    for name, ts in times.items():
        med = statistics.median(ts)
        print("%-28s median %.4f ms  min %.4f  frac(median) %.4f" % (name, med, min(ts), 1220608 * M / med / 1e9 / 2500))
    if differ and "--allow-differ" not in sys.argv:      # (timing-only experiment bodies are SUPPOSED to differ: say so on the command line)
        sys.exit("g2_ab: a library's outputs differ from the first library's")


if __name__ == "__main__":
    main()
