#!/usr/bin/env python3
"""Which host-side code enqueues the small fill / copy / elementwise launches of a training step (the step's kernel statistics show ~40
fills and ~26 copies per step, ~5 us each on the step's one stream)?  torch's profiler with Python stacks over a few steps of
bench.py's training leg; prints every aten op that launched a kernel of < 20 us, grouped by the innermost frames of this repository.
GPU box: python3 tools/train_small_ops.py [fp32|x3]"""
import collections
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    mlp = sys.argv[1] if len(sys.argv) > 1 else "fp32"
    args = bench.parse(["--mode", "train", "--mlp", mlp, "--no-cpu-baseline"])
    device = torch.device("cuda:0")
    model, cfg, _, _ = bench.build_model(args, device)
    from ddnerf_amd import synthetic, train_step

    ro, rd, rad, tgt = (torch.from_numpy(x).to(device) for x in synthetic.make_rays(args.ray_kind, args.rays, 1))
    stepper = train_step.TrainStepper(model, cfg)
    for _ in range(3):
        stepper.step(ro, rd, rad, tgt)
    torch.cuda.synchronize()
    steps = 3
    with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU, torch.profiler.ProfilerActivity.CUDA], with_stack=True) as prof:
        for _ in range(steps):
            stepper.step(ro, rd, rad, tgt)
        torch.cuda.synchronize()
    groups = collections.Counter()
    times = collections.Counter()
    for ev in prof.events():
        if not ev.name.startswith("aten::") or ev.device_time_total <= 0 or ev.device_time_total > 20 * max(1, len(ev.kernels)):
            continue
        if not ev.kernels:
            continue
        frames = [f for f in (ev.stack or []) if "/root/repo" in f or ROOT in f or "ddnerf_amd" in f or "train_step" in f]
        where = " <- ".join(os.path.basename(f.split(",")[0]) + ":" + f.split("(")[-1].split(")")[0] if "(" in f else f for f in frames[:2]) or "(autograd engine / no repository frame)"
        key = (ev.name, where)
        groups[key] += len(ev.kernels)
        times[key] += ev.device_time_total
    print("%-28s %8s %10s  %s" % ("op", "launches", "us / step", "where (innermost repository frames)"))
    for key, n in sorted(groups.items(), key=lambda kv: -times[kv[0]]):
        print("%-28s %8.1f %10.1f  %s" % (key[0], n / steps, times[key] / steps, key[1]))
    print("total: %.1f launches, %.1f us per step" % (sum(groups.values()) / steps, sum(times.values()) / steps))


if __name__ == "__main__":
    main()
