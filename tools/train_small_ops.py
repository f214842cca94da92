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
    rf = torch.profiler.record_function
    with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU, torch.profiler.ProfilerActivity.CUDA]) as prof:
        for _ in range(steps):
            # TrainStepper.step, phase by phase (train_step.py)
            with rf("PHASE forward"):
                stepper.smooth.apply(model.cfg, stepper.iter)
                model.train()
                out = model.run_iter(ro, rd, rad, mode="train", rgb_target=tgt)
            with rf("PHASE loss"):
                losses = [torch.nn.functional.mse_loss(out[j]["rgb"], tgt) for j in range(len(out))]
                loss = cfg.train_params.loss_coeficients[0] * losses[0]
                for j in range(1, len(out)):
                    loss = loss + cfg.train_params.loss_coeficients[j] * losses[j]
                if stepper.dd:
                    loss = loss + cfg.train_params.dp_coeficient * out[1]["dp_loss"].mean()
            with rf("PHASE backward"):
                loss.backward()
            with rf("PHASE optimizer"):
                for o in stepper.optims:
                    o.step()
                    o.zero_grad()
            stepper.iter += 1
        torch.cuda.synchronize()
    groups = collections.Counter()
    times = collections.Counter()

    def where(ev):
        chain, p = [], ev.cpu_parent
        while p is not None:
            if p.name.startswith("PHASE ") or "evaluate_function" in p.name or p.name.endswith("Backward") or "Function" in p.name:
                chain.append(p.name.replace("autograd::engine::evaluate_function: ", ""))
            p = p.cpu_parent
        return " <- ".join(chain[:2]) or "(no parent)"

    for ev in prof.events():
        if not ev.name.startswith("aten::") or not ev.kernels:
            continue
        dt = sum(k.duration for k in ev.kernels)
        if dt > 20 * len(ev.kernels):
            continue
        if any(c.kernels for c in ev.cpu_children if c.name.startswith("aten::")):
            continue            # (count the innermost op that launched)
        key = (ev.name, where(ev))
        groups[key] += len(ev.kernels)
        times[key] += dt
    print("%-26s %8s %10s  %s" % ("op", "launches", "us / step", "where"))
    for key, n in sorted(groups.items(), key=lambda kv: -times[kv[0]]):
        print("%-26s %8.1f %10.1f  %s" % (key[0], n / steps, times[key] / steps, key[1]))
    print("total: %.1f launches, %.1f us of kernel time per step (each also costs its launch gap on the step's one stream)" % (sum(groups.values()) / steps, sum(times.values()) / steps))


if __name__ == "__main__":
    main()
