#!/usr/bin/env python3
"""Benchmark of the DDNeRF ray-march hot path on MI355X (contract: see the task's bench.py section).

A "step" is one pass of the hot path over one batch of synthetic rays: `model.run_iter(...)` of the HIP-backed model of
`--config` (default BASELINE.json configs[1]: config_blender.yml, DDNerfModel, 4096 rays x (64 coarse + 128 fine) samples,
8x256 MLPs, fp32) per GPU.  Weak scaling: every rank renders its own ray batch, no data-path collective in render mode;
the train pass adds ONE RCCL all-reduce of the flat gradient buffer per network per step.  Inputs are resident in HBM
before the timed region.

`python bench.py --gpus N ...` with N > 1 and no WORLD_SIZE in the environment starts the N ranks itself (one child process
per GPU with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT set) before anything touches the GPU, and
relays rank 0's line; under a launcher (torch.distributed.run) it checks that WORLD_SIZE == --gpus.

Prints ONE JSON line on rank 0: the metric (render rays/s, whole job), a `roofline` object for the dominant kernel (the
fused fine-MLP forward: algorithmic FLOP per launch / mean launch duration measured with HIP events on the launch stream)
and
  * at N = 1 (default run): `cpu_baseline` (the CPU oracle timed on a bounded sample of the same workload), `x3_tier` /
    `bf16_tier` (the same workload on the split-precision bf16x3 and on the plain bf16 MLP kernels, each with its own roofline
    fraction against the bf16 MFMA peak) and `train_tier` (training steps of the same workload);
  * at N > 1: `train` (forward + backward + gradient all-reduce + Adam of the same per-GPU batch, timed the same way),
    `rccl_ranks` and `backend`.
"""
import argparse
import hashlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FLOP_FINE = 1220608       # per sample, MipNeRFModel forward
FLOP_COARSE_DD = 1221120  # per sample, DepthMipNeRFModel forward
PEAK = {"fp32": 157.3, "bf16": 2500.0, "x3": 2500.0, "fp16": 2500.0}  # dense MFMA TFLOP/s, /opt/skills/guides/MI355X_MICROARCH.md
DTYPE = {"fp32": "f32", "bf16": "bf16 (f32 accumulate)", "fp16": "fp16 (f32 accumulate)", "x3": "f32 as exact hi+lo bf16 splits, 3 bf16 MFMAs per product (f32 accumulate)"}
TRAIN_DTYPE = {"fp32": "f32 forward / backward-data, weight gradients as bf16 hi+lo splits (3 MFMAs per product, f32 accumulate)",
               "x3": "f32 forward / backward-data as exact hi+lo bf16 splits (3 bf16 MFMAs per product), weight gradients from bf16-rounded "
                     "activation / delta records (1 MFMA per product); f32 accumulation throughout"}
# BASELINE.json configs -> (synthetic ray kind, default MLP kernel, default rays per GPU)
CONFIGS = {
    "config_blender.yml": ("blender", "fp32", 4096),          # configs[1] (and [0] at 256 rays)
    "config_ff.yml": ("llff", "bf16", 4096),                  # configs[2]: NDC rays, bf16-MFMA MLP
    "config_360.yml": ("real360", "fp32", 8192),              # configs[3]: 8192 rays, data-parallel training
    "config_blender_mipnerf.yml": ("blender", "fp32", 4096),  # configs[4]: GeneralMipNerfModel, one shared MLP
    "config_ff_mipnerf.yml": ("llff", "fp32", 4096),
    "config_360_mipnerf.yml": ("real360", "fp32", 8192),
}
KERNEL_SOURCES = {"fp32": ["mlp_f32.hip", "mlp_f32_fwd.inc", "mlp_f32_common.h"], "bf16": ["mlp_bf16.hip", "mlp_bf16_g2.hip", "mlp_bf16_g2e.hip", "mlp_bf16_g2_body_d0.gen.inc", "mlp_bf16_g2e_body_d0.gen.inc", "mlp_bf16_g2_tables.gen.inc", "mlp_mfma16.inc", "mlp_bf16_common.h"],
                  "x3": ["mlp_x3_fwd.hip", "mlp_x3_fwd_rays.hip", "mlp_mfma16.inc", "mlp_bf16_common.h"],
                  "fp16": ["mlp_f16.hip", "mlp_f16_g2.hip", "mlp_f16_g2e.hip", "mlp_bf16.hip", "mlp_bf16_g2.hip", "mlp_bf16_g2e.hip", "mlp_f16_g2_body_d0.gen.inc", "mlp_f16_g2e_body_d0.gen.inc", "mlp_bf16_g2_tables.gen.inc", "mlp_mfma16.inc", "mlp_bf16_common.h"]}


def parse(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=20)
    p.add_argument("--warmup", type=int, default=3)
    p.add_argument("--config", default="config_blender.yml", choices=sorted(CONFIGS), help="configs/<name> (BASELINE.json configs)")
    p.add_argument("--rays", type=int, default=None, help="rays per GPU per step (default: the config's BASELINE size)")
    p.add_argument("--global-rays", type=int, default=None, help="strong scaling: this many rays per step split over the ranks")
    p.add_argument("--coarse", type=int, default=64)
    p.add_argument("--fine", type=int, default=128)
    p.add_argument("--mode", choices=["render", "train", "both"], default=None, help="default: render at N = 1, both at N > 1")
    p.add_argument("--mlp", choices=["fp32", "x3", "bf16", "fp16"], default=None, help="default: the config's BASELINE numerical mode")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-bf16-tier", action="store_true", help="skip the extra x3 / bf16 kernel measurements of the default run")
    p.add_argument("--cpu-rays", type=int, default=512, help="rays of the same workload timed on the CPU oracle")
    p.add_argument("--ramp", type=int, default=None, help="untimed steps BEFORE the W warm-up steps: the power-limited bf16 / x3 kernels need "
                   "about 30 steps of sustained load before the chip's clock settles (3 warm-up steps: fine MLP 0.422 ms, 30 or more: 0.404); "
                   "default 40 for the bf16 / x3 render pass, 0 otherwise")
    p.add_argument("--single-rank-rccl", action="store_true", help="run the N > 1 code path (RCCL process group, barrier fences, MAX all-reduce of the "
                   "time, the train leg with its gradient all-reduce) on a ONE-rank group: how a one-GPU box executes those lines (tests/test_bench_contract.py)")
    p.add_argument("--backend", choices=["nccl", "gloo"], default="nccl", help="process-group backend of the N > 1 path; gloo only for rehearsals")
    p.add_argument("--share-gpu", action="store_true", help="rehearsal on a box with fewer GPUs than ranks: rank r uses GPU r %% (GPUs present); needs --backend gloo "
                   "(RCCL refuses two ranks on one device)")
    p.add_argument("--image", default=None, metavar="HxW", help="also render one full validation image of this size through run_iter (all its ray chunks) and "
                   "report seconds per image beside the per-chunk figure (eval_nerf.py's product-level timing)")
    p.add_argument("--no-clock", action="store_true", help="skip the bf16 line's in-kernel clock measurement (thousands of launches of the "
                   "diagnostic build: they would drown the product kernel in a profiler's per-kernel statistics)")
    args = p.parse_args(argv)
    kind, mlp, rays = CONFIGS[args.config]
    args.ray_kind = kind
    args.mlp = args.mlp or mlp
    args.mode = args.mode or ("render" if (args.gpus == 1 and not args.single_rank_rccl) else "both")
    if args.share_gpu and args.backend != "gloo":
        p.error("--share-gpu needs --backend gloo")
    if args.image:
        h, w = args.image.lower().split("x")
        args.image = (int(h), int(w))
    if args.ramp is None:
        args.ramp = TIER_RAMP if (args.mlp in ("bf16", "x3", "fp16") and args.mode == "render") else 0
    args.scaling = "strong" if args.global_rays else "weak"
    if args.global_rays:
        if args.global_rays % args.gpus:
            p.error("--global-rays must be a multiple of --gpus")
        args.rays = args.global_rays // args.gpus
    args.rays = args.rays or rays
    return args


def rank_environments(n, port):
    """The environment of each of the N ranks the launcher starts (what torch.distributed.run would set, one node)."""
    # (the collective timeouts end a rank that waits for a dead peer well inside the supervisor's limit)
    return [dict(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                 MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0", TORCH_NCCL_ASYNC_ERROR_HANDLING="1",
                 TORCH_NCCL_HEARTBEAT_TIMEOUT_SEC="300", DDNERF_PG_TIMEOUT_S="300") for r in range(n)]


def free_port():
    """a TCP port nobody listens on right now (bound to port 0 on 127.0.0.1, then released)"""
    import socket

    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


RANK_TIMEOUT_S = 540  # the whole N-rank run is bounded below the driver's own 600 s limit


def supervise(procs, timeout_s, poll_s=0.2, clock=time.monotonic, sleep=time.sleep):
    """Wait for every child.  On the first non-zero exit, or when `timeout_s` has passed, terminate (then kill) the others:
    a rank that died before rendezvous must not leave rank 0 in a collective until the c10d timeout.  Rank 0's stdout is
    drained by a thread so a chatty child cannot block on a full pipe.  Returns (exit codes, rank 0's stdout, reason)."""
    import threading

    out = []
    reader = None
    if procs and procs[0].stdout is not None:
        reader = threading.Thread(target=lambda: out.append(procs[0].stdout.read()), daemon=True)
        reader.start()
    t0, reason = clock(), None
    while True:
        codes = [p.poll() for p in procs]
        if all(c is not None for c in codes):
            break
        bad = [r for r, c in enumerate(codes) if c not in (None, 0)]
        if bad:
            reason = "rank %d exited with code %s" % (bad[0], codes[bad[0]])
        elif clock() - t0 > timeout_s:
            reason = "no result after %d s" % timeout_s
        if reason:
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            t1 = clock()
            while any(p.poll() is None for p in procs) and clock() - t1 < 10:
                sleep(poll_s)
            for p in procs:
                if p.poll() is None:
                    p.kill()
            for p in procs:
                p.wait()
            break
        sleep(poll_s)
    if reader is not None:
        reader.join(timeout=10)
    return [p.returncode for p in procs], (out[0] if out else ""), reason


def launch_ranks(args, argv):
    """--gpus N without a launcher: start the N ranks as child processes, one per GPU (nothing in this process has touched
    the GPU), supervise them (first failure or the time limit ends all of them), relay rank 0's single JSON line and exit
    non-zero when anything went wrong."""
    import torch

    have = torch.cuda.device_count()  # (does not initialise the GPU)
    if have < (1 if args.share_gpu else args.gpus):
        raise SystemExit("bench.py --gpus %d: only %d GPU(s) visible on this node" % (args.gpus, have))
    port = free_port()
    cmd = [sys.executable, os.path.abspath(__file__)] + list(argv)
    procs = []
    for r, env in enumerate(rank_environments(args.gpus, port)):
        procs.append(subprocess.Popen(cmd, env=dict(os.environ, **env), stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    codes, out, reason = supervise(procs, RANK_TIMEOUT_S)
    lines = [l for l in out.splitlines() if l.startswith("{") and '"metric"' in l]
    if reason or any(codes) or len(lines) != 1:
        sys.stderr.write(out[-4000:])
        raise SystemExit("bench.py --gpus %d: %s; rank exit codes %s, %d result lines" % (args.gpus, reason or "failed", codes, len(lines)))
    print(lines[0], flush=True)
    raise SystemExit(0)


def build_model(args, device, mlp=None):
    import torch

    from ddnerf_amd import synthetic
    from ddnerf_amd.cfgnode import CfgNode
    from models import models

    cfg = CfgNode.load(os.path.join(ROOT, "configs", args.config))
    for mode in ("train", "validation"):
        cfg.nerf[mode]["num_coarse"] = args.coarse
        cfg.nerf[mode]["num_fine"] = args.fine
    cfg.nerf["mlp_dtype"] = mlp or args.mlp
    cfg.train_params.dist_reg_coeficient = min(max(1 / args.coarse, 0.01), 0.12)  # train_model.py:124-125
    if cfg.dataset.get("normalize_poses", False):  # data_utils/data_utils.py:65-74 rescales near/far with the poses
        cfg.dataset.near = cfg.dataset.near / cfg.dataset.normalize_factor
        cfg.dataset.far = cfg.dataset.far / cfg.dataset.normalize_factor
    model = getattr(models, cfg.nerf.type)(cfg)
    dd = cfg.nerf.type == "DDNerfModel"
    sd_c = synthetic.make_state_dict(dd, 11, 20.0)   # weight set B ("sharpened"), SURVEY.md 8d
    sd_f = synthetic.make_state_dict(False, 12, 20.0)
    model.coarse.load_state_dict({k: torch.from_numpy(v) for k, v in sd_c.items()})
    if model.fine is not model.coarse:
        model.fine.load_state_dict({k: torch.from_numpy(v) for k, v in sd_f.items()})
    model.to(device)
    return model, cfg, sd_c, sd_f


class KernelTimer:
    """HIP-event timing of the dominant kernel (the MLP launch of `only` samples: the fine pass) on torch's current stream (the stream
    the C ABI launches on).  An event record is a barrier packet of its own: ~6 us of idle GPU in front of and behind the launch it
    brackets (rocprofv3 kernel trace), so only the launches that are reported carry one -- timing the coarse launch too cost the
    0.75-ms bf16 step another 12 us.  `every`: time every n-th such launch of the timed region only (the 16-bit tiers, whose whole step
    is 0.69 ms: a pair of events around EVERY fine launch is 12 us = 1.7 % of the step it is there to measure; around every fourth, 3 us);
    `seen` counts the launches the timed region contained, `launches_timed` in the line those that carry events."""

    def __init__(self, only=None, every=1):
        self.pairs = []
        self.active = False
        self.only = only
        self.every = max(1, int(every))
        self.seen = 0

    def __call__(self, M, launch):
        import torch

        if not self.active or (self.only is not None and M != self.only):
            return launch()
        self.seen += 1
        if (self.seen - 1) % self.every:
            return launch()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = launch()
        e1.record()
        self.pairs.append((M, e0, e1))
        return out

    def mean_ms(self, M):
        ts = [a.elapsed_time(b) for m, a, b in self.pairs if m == M]
        return (sum(ts) / len(ts), len(ts)) if ts else (None, 0)


def child_line(args, extra, env=None):
    cmd = [sys.executable, os.path.abspath(__file__), "--config", args.config, "--rays", str(args.rays), "--coarse", str(args.coarse),
           "--fine", str(args.fine), "--no-cpu-baseline", "--no-bf16-tier"] + extra
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=dict(os.environ, **(env or {})))
    return json.loads(out.stdout.strip().splitlines()[-1])


TIER_RAMP = 40  # untimed steps in front of a tier's warm-up (see --ramp)


TRAIN_TIERS = {   # name -> (--mlp, environment of the child run, what the weight gradients contract)
    "fp32": ("fp32", {}, None),
    "fp32_pairs": ("fp32", {"DDNERF_WGRAD": "pairs"}, "opt-in speed mode DDNERF_WGRAD=pairs: exact-f32 forward / backward-data, weight gradients from bf16-rounded "
                                                   "row-pair records (1 MFMA per product): NOT fp32-class"),
    "x3": ("x3", {}, None),
    "x3_exact": ("x3", {"DDNERF_X3_WGRAD": "exact"}, "strict mode DDNERF_X3_WGRAD=exact: weight gradients from exact hi/lo-word records (3 MFMAs per product): fp32-class"),
}


def train_tier(args, name):
    """Training throughput (forward + backward + Adam per step) of the same workload, from a child run of
    `bench.py --mode train --mlp <mlp>`; reported beside the render headline, never as `value`."""
    mlp, env, note = TRAIN_TIERS[name]
    try:
        d = child_line(args, ["--mode", "train", "--mlp", mlp, "--steps", "5", "--warmup", "2", "--ramp", "5"], env)
    except Exception as e:
        return {"error": "%s: %s" % (type(e).__name__, e)}
    out = {"value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"], "dtype": d["dtype"], "roofline": d.get("roofline")}
    if note:
        out["dtype"] = note
    return out


def extra_tier(args, mlp):
    """The same render workload on another MLP kernel, reported beside the exact-fp32 headline, never as `value`:
    "bf16" = plain bf16 MFMA (BASELINE configs[2]'s numerical mode; the north-star roofline target is stated against the
    bf16 MFMA peak); "fp16" = the same kernels on the fp16 forms of the instructions (11 significant bits, same rate); "x3" = bf16 MFMA with exact hi/lo operand splits (three MFMAs per product, fp32-class accuracy: it
    meets the same 1e-4 parity bar as the exact kernel) -- its roofline counts the 3x bf16 MFMA work it really issues.
    Measured by a child `bench.py --mlp <tier>` run, started before this process initialises the GPU (inside this process,
    behind the fp32 run, the launch-heavy bf16 step measures up to 4x slower; beside an idle parent context 1.6x)."""
    try:
        d = child_line(args, ["--mlp", mlp, "--steps", str(args.steps), "--warmup", str(args.warmup), "--ramp", str(TIER_RAMP)])
    except Exception as e:  # the headline must not die with a tier
        return {"error": "%s: %s" % (type(e).__name__, e)}
    roof = d["roofline"]
    return {"value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"], "dtype": d["dtype"], "ramp_steps": TIER_RAMP, "roofline": roof}


def image_tier(args, mlp, size="800x800"):
    """seconds per full validation image (all its 16384-ray chunks through run_iter) on one MLP kernel, beside the rate of ONE such chunk
    measured in the same child run"""
    try:
        cmd = [sys.executable, os.path.abspath(__file__), "--config", args.config, "--rays", "16384", "--coarse", str(args.coarse), "--fine", str(args.fine),
               "--no-cpu-baseline", "--no-bf16-tier", "--no-clock", "--mlp", mlp, "--image", size, "--steps", "5", "--warmup", "2"]
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
        d = json.loads(out.stdout.strip().splitlines()[-1])
        im = d["image"]            # (a child line without it -- an error object, an older line format -- is an error HERE, not in the caller)
        im["chunk_rays_per_s"] = d["value"]
        im["image_over_chunk_rate"] = round(im["rays_per_s"] / d["value"], 4)
        return im
    except Exception as e:
        return {"error": "%s: %s" % (type(e).__name__, e)}


def cpu_baseline(args, cfg, sd_c, sd_f, check_model=None):
    """The CPU oracle (a C port of the reference path, oracle/) on a bounded sample of the same workload."""
    import numpy as np
    import torch

    import oracle as O
    from ddnerf_amd import synthetic

    threads = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    threads = min(threads, int(os.environ.get("DDNERF_CPU_THREADS", "64")))
    O.set_threads(threads)
    t_lin = torch.linspace(0.0, 1.0, args.coarse + 1).numpy()
    u_det = torch.linspace(0.0, 0.9999, args.fine + 1).numpy()

    dd = cfg.nerf.type == "DDNerfModel"
    mc = cfg.nerf.validation
    is_blender = str(cfg.dataset.type).lower() == "blender" or str(cfg.dataset.basedir).endswith("segmented")
    if not dd:  # models/samplers.py:93: the mip sampler's deterministic u includes 1.0
        u_det = torch.linspace(0.0, 1.0, args.fine + 1).numpy()

    def run(n):
        ro, rd, rad, _ = synthetic.make_rays(args.ray_kind, n, 1)   # this config's ray kind, near / far, dataset branch, sampler
        rng = np.random.default_rng(0)
        kw = dict(model="dd" if dd else "mip", nc=args.coarse, nf=args.fine, near=float(cfg.dataset.near), far=float(cfg.dataset.far),
                  blender=is_blender, white_bkgd=bool(mc.white_background), lindisp=bool(mc.lindisp),
                  pdf_padding=bool(cfg.train_params.pdf_padding), smooth=float(cfg.train_params.gaussian_smooth_factor),
                  dist_reg=float(cfg.train_params.dist_reg_coeficient), t_lin=t_lin, u_det=u_det,
                  noise0=rng.standard_normal((n, args.coarse)).astype(np.float32),
                  noise1=rng.standard_normal((n, args.fine)).astype(np.float32))
        t0 = time.perf_counter()
        O.run_iter(ro, rd, rad, sd_c, sd_f, **kw)
        return time.perf_counter() - t0

    parity = None
    if check_model is not None:
        # the path that was just timed against the CPU oracle, noise off, 64 rays of the same kind: the line is self-validating (a
        # render that runs fast and computes something else is not a number)
        import torch as _t

        n0 = 64
        ro, rd, rad, _ = synthetic.make_rays(args.ray_kind, n0, 99)
        kw = dict(model="dd" if dd else "mip", nc=args.coarse, nf=args.fine, near=float(cfg.dataset.near), far=float(cfg.dataset.far),
                  blender=is_blender, white_bkgd=bool(mc.white_background), lindisp=bool(mc.lindisp),
                  pdf_padding=bool(cfg.train_params.pdf_padding), smooth=float(cfg.train_params.gaussian_smooth_factor),
                  dist_reg=float(cfg.train_params.dist_reg_coeficient), t_lin=t_lin, u_det=u_det)
        want = O.run_iter(ro, rd, rad, sd_c, sd_f, **kw)
        std = {m: cfg.nerf[m]["radiance_field_noise_std"] for m in ("train", "validation")}
        try:
            for m in std:
                cfg.nerf[m]["radiance_field_noise_std"] = 0.0
            dev = next(check_model.coarse.parameters()).device
            with _t.no_grad():
                got = check_model.run_iter(*(_t.from_numpy(x).to(dev) for x in (ro, rd, rad)), mode="validation")
        finally:
            for m in std:
                cfg.nerf[m]["radiance_field_noise_std"] = std[m]
        errs = {"L%d_%s" % (lvl, k): float(np.abs(got[lvl][k].reshape(want[lvl][k].shape).cpu().numpy() - want[lvl][k]).max())
                for lvl in (0, 1) for k in ("rgb", "depth")}
        parity = {"rays": n0, "noise": "off", "max_abs_err_vs_cpu_oracle": {k: float("%.3g" % v) for k, v in errs.items()},
                  "within_1e-4": bool(max(errs.values()) <= 1e-4)}

    run(64)                                  # warm-up (thread pool, page faults)
    probe = run(args.cpu_rays)               # calibrate, then size the sample for ~15 s of CPU work
    n = int(min(max(args.cpu_rays, args.cpu_rays * 15.0 / max(probe, 1e-3)), 65536)) // 256 * 256
    dt = run(n)
    out = {"value": n / dt, "unit": "rays/s", "cores": O.get_threads(), "kind": "port",
           "sample": "%s, %d %s rays x (%d+%d) samples, render pass, C / OpenMP port of the reference path (oracle/), %.1f s"
                     % (args.config, n, args.ray_kind, args.coarse, args.fine, dt)}
    if parity is not None:
        out["parity_check"] = parity
    return out


def bf16_in_kernel_clock(flat_params, device, seconds=2.5):
    """In-kernel clock and matrix-pipe busy share of the bf16 fine-MLP kernel (MI355X_MICROARCH.md, DVFS give-back item 6):
    the DIAGNOSTIC build of the same kernel sources (ddnerf_amd/csrc/libddnerf_diag.so, -DBF16_STAMP) runs back to back on random
    bf16 feature rows with this model's fine-network weights for >= `seconds`; every workgroup stamps s_memtime / s_memrealtime
    around its tile loop.  clock = d(s_memtime) / d(s_memrealtime) x 100 MHz; cycles per tile against the MFMAs x 16 cycles the tile
    needs (a wave issues 4820 per 256-sample tile of the one-group kernel, 9640 per 512-sample tile of the two-group kernel: at this
    size ddnerf_mlp_bf16_forward runs the latter).  Runs after the timed region; nothing in the product library executes a stamp."""
    import ctypes as C

    import numpy as np
    import torch

    from ddnerf_amd import build as hip_build

    if not os.path.exists(hip_build.DIAG_SO):
        return None
    L = C.CDLL(hip_build.DIAG_SO)
    V = C.c_void_p
    M = 524288
    st = torch.cuda.current_stream().cuda_stream
    fb = (torch.rand(M, 128, device=device) * 2 - 1).to(torch.bfloat16).contiguous()
    raw = torch.empty(M, 4, device=device)
    L.ddnerf_mlp_bf16_packed_bytes.restype = C.c_size_t
    packed = torch.empty(L.ddnerf_mlp_bf16_packed_bytes(0), dtype=torch.uint8, device=device)
    L.ddnerf_mlp_bf16_pack.argtypes = [V, C.c_int, V, V]
    f = L.ddnerf_mlp_bf16_forward
    f.argtypes = [V, V, C.c_int, V, C.c_long, V]
    L.ddnerf_debug_set_stamps.argtypes = [V]
    L.ddnerf_debug_set_stamps_g2.argtypes = [V]
    n_cu = torch.cuda.get_device_properties(device).multi_processor_count
    stamps = {"g1": torch.zeros(n_cu * 6, dtype=torch.int64, device=device),
              "g2": torch.zeros(n_cu * (6 + 192), dtype=torch.int64, device=device)}   # (+ the two-group build's per-period stamps)
    if (L.ddnerf_mlp_bf16_pack(flat_params.data_ptr(), 0, packed.data_ptr(), st) or L.ddnerf_debug_set_stamps(stamps["g1"].data_ptr())
            or L.ddnerf_debug_set_stamps_g2(stamps["g2"].data_ptr())):
        return None
    t0, n = time.time(), 0
    while time.time() - t0 < seconds:
        for _ in range(50):
            f(fb.data_ptr(), packed.data_ptr(), 0, raw.data_ptr(), M, st)
        torch.cuda.synchronize()
        n += 50
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        f(fb.data_ptr(), packed.data_ptr(), 0, raw.data_ptr(), M, st)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 50
    which = "g2" if int(stamps["g2"][:6 * n_cu].abs().sum()) else "g1"
    s = stamps[which][:6 * n_cu].cpu().numpy().reshape(n_cu, 6).astype(np.float64)
    s = s[s[:, 4] > 0]
    clk = float(np.median((s[:, 2] - s[:, 0]) / (s[:, 3] - s[:, 1]) * 100.0))
    cyc = float(np.median((s[:, 2] - s[:, 0]) / s[:, 4]))
    tile = 512 if which == "g2" else 256
    ideal = 4820 * 16 * (tile // 256)
    return {"kernel": "two groups per weight pass (mlp_bf16_g2.hip)" if which == "g2" else "one group (mlp_bf16.hip)",
            "in_kernel_clock_mhz": round(clk), "nominal_clock_mhz": 2400, "tile_samples": tile, "cycles_per_tile": round(cyc),
            "ideal_cycles_per_tile": ideal, "mfma_busy": round(ideal / cyc, 4), "frac_bound_at_this_clock": round(clk / 2400.0, 4),
            "back_to_back_launch_ms": round(ms, 4), "back_to_back_frac": round(FLOP_FINE * M / (ms * 1e-3) / 1e12 / PEAK["bf16"], 4),
            "warm_launches": n, "how": "diagnostic stamp build of the same kernel sources, random bf16 feature rows, after the timed region; "
            "frac ~= mfma_busy x in_kernel_clock / nominal_clock (less launch prologue / tail; the stamps themselves cost the diagnostic "
            "build a few per cent of busy share)"}


def f32_in_kernel_clock(model, device, seconds=0.6):
    """In-kernel clock of the HEADLINE kernel (the fp32 fine-MLP forward) and what its workgroups spend their cycles on, from the
    diagnostic build of the same sources (-DF32_STAMP_TILE: every persistent workgroup stamps s_memtime / s_memrealtime at its first
    and behind its last instruction), run back to back for `seconds` on the encoded fine pass of 4096 synthetic rays with this model's
    weights, after the timed region.  Closes the roofline fraction of THAT run: frac = (algorithmic / issued FLOP) x (MFMA cycles /
    workgroup cycles) x (workgroup time / launch time) x in-kernel clock / 2400."""
    import ctypes as C

    import numpy as np
    import torch

    from ddnerf_amd import build as hip_build
    from ddnerf_amd import functions as F
    from ddnerf_amd import ops, synthetic

    if not os.path.exists(hip_build.DIAG_SO):
        return None
    L = C.CDLL(hip_build.DIAG_SO)
    if not hasattr(L, "ddnerf_debug_f32_tile_stamps"):
        return None
    V = C.c_void_p
    n, S = 4096, 128
    M = n * S
    st = torch.cuda.current_stream().cuda_stream
    o, d, rad, _ = synthetic.make_rays("blender", n, 1)
    rays = ops.pack_rays(*(torch.from_numpy(x).to(device) for x in (o, d, rad)), 2.0, 6.0)
    t = (2.0 + 4.0 * torch.sort(torch.rand(n, S + 1, device=device), dim=1).values).contiguous()
    feat = ops.encode(rays, t, kind="fp32")
    packed = F._packed_weights(model.fine)         # (the product library's image: the diagnostic build reads the same format)
    raw = torch.empty(M, 4, device=device)
    f = L.ddnerf_mlp_f32_forward
    f.argtypes = [V, V, C.c_int, V, C.c_long, V]
    launch = lambda: f(feat.data_ptr(), packed.data_ptr(), 0, raw.data_ptr(), M, st)
    t0, nl = time.time(), 0
    while time.time() - t0 < seconds:
        for _ in range(10):
            launch()
        torch.cuda.synchronize()
        nl += 10
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        launch()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    buf = np.zeros(4096 * 6, dtype=np.uint64)
    if L.ddnerf_debug_f32_tile_stamps(C.c_void_p(buf.ctypes.data)):
        return None
    s = buf.reshape(4096, 6).astype(np.int64)
    s = s[s[:, 2] > 0]                              # one row per persistent workgroup (the stamps of the LAST launch)
    tiles_per_wg = (M // 128) / len(s)
    dur = (s[:, 2] - s[:, 0]).astype(np.float64)
    real = (s[:, 4] - s[:, 3]) / 100e6
    clk = float(np.median(dur / real / 1e6))
    kk = [96] * 8 + [256] * 32 + [352] * 8 + [256] * 24 + [288] * 5 + [128]          # K of every 32-row weight slice of the network
    mfma_cycles = sum(32 * k for k in kk) * tiles_per_wg                            # k / 2 MFMAs of 64 cycles per slice and tile
    mfma_share = float(mfma_cycles / np.median(dur))
    wg_share = float(np.median(real) * 1e3 / ms)
    issued_over_alg = sum(2 * 32 * k for k in kk) / FLOP_FINE                           # per sample: the padded rows / columns the tiles carry
    frac = FLOP_FINE * M / (ms * 1e-3) / 1e12 / PEAK["fp32"]
    pred = mfma_share * wg_share * clk / 2400.0 / issued_over_alg
    return {"in_kernel_clock_mhz": round(clk), "nominal_clock_mhz": 2400, "mfma_share_of_workgroup_cycles": round(mfma_share, 4),
            "workgroup_share_of_launch": round(wg_share, 4), "issued_over_algorithmic_flop": round(issued_over_alg, 4),
            "back_to_back_launch_ms": round(ms, 4), "back_to_back_frac": round(frac, 4), "frac_from_the_stamps": round(pred, 4),
            "closes_within": round(abs(pred / frac - 1.0), 4), "warm_launches": nl,
            "how": "diagnostic stamp build (-DF32_STAMP_TILE) of the headline kernel's sources on the encoded fine pass of 4096 synthetic rays, "
                   "after the timed region; frac_from_the_stamps = mfma_share x workgroup_share x clock / 2400 / issued_over_algorithmic"}


def bf16_unfused_mlp_launch(model, device, rays=4096, S=128, launches=200):
    """The SAME fine MLP as a launch of its own (ddnerf_mlp_bf16_forward on rows ddnerf_encode wrote): what `roofline.frac` was before the
    encoder moved into the kernel, measured in this run on this device -- the timed step's launch also encodes, so its `frac` (MLP FLOP /
    launch time) is not comparable with earlier rounds' without this figure beside it.  After the timed region, back to back."""
    import torch

    from ddnerf_amd import functions as F
    from ddnerf_amd import ops, synthetic

    o, d, rad, _ = synthetic.make_rays("blender", rays, 1)
    packed_rays = ops.pack_rays(*(torch.from_numpy(x).to(device) for x in (o, d, rad)), 2.0, 6.0)
    t = (2.0 + 4.0 * torch.sort(torch.rand(rays, S + 1, device=device), dim=1).values).contiguous()
    feat = ops.encode(packed_rays, t, kind="bf16")
    packed = F._packed_weights(model.fine)
    hook, ops.MLP_LAUNCH_HOOK = ops.MLP_LAUNCH_HOOK, None
    try:
        for _ in range(launches):
            ops.mlp_bf16_forward(feat, packed, False)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(launches):
            ops.mlp_bf16_forward(feat, packed, False)
        e1.record()
        torch.cuda.synchronize()
        enc0, enc1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        enc0.record()
        for _ in range(launches):
            ops.encode(packed_rays, t, kind="bf16")
        enc1.record()
        torch.cuda.synchronize()
    finally:
        ops.MLP_LAUNCH_HOOK = hook
    ms = e0.elapsed_time(e1) / launches
    M = rays * S
    return {"kernel": "mlp_bf16g2_fwd_kernel<fine> on the rows of ddnerf_encode (%d samples/launch)" % M, "launch_ms": round(ms, 4),
            "frac": round(FLOP_FINE * M / (ms * 1e-3) / 1e12 / PEAK["bf16"], 4), "encode_launch_ms": round(enc0.elapsed_time(enc1) / launches, 4),
            "launches_timed": launches, "how": "back to back after the timed region, HIP events around %d launches" % launches}


def bf16_fused_in_kernel_clock(model, device, seconds=2.5):
    """As bf16_in_kernel_clock for the FUSED kernel the bf16 render step runs (mlp_bf16_g2e.hip: the encoder inside the MLP kernel):
    the diagnostic build of the same sources back to back on a fine pass of 4096 synthetic rays x 128 sorted fenceposts; its tile
    loop's cycles include the encoder's work and, once per workgroup, the straight-line encoder of the first tile."""
    import ctypes as C

    import numpy as np
    import torch

    from ddnerf_amd import build as hip_build
    from ddnerf_amd import functions as F
    from ddnerf_amd import ops, synthetic

    if not os.path.exists(hip_build.DIAG_SO):
        return None
    L = C.CDLL(hip_build.DIAG_SO)
    if not hasattr(L, "ddnerf_debug_set_stamps_g2e"):
        return None
    V = C.c_void_p
    n, S = 4096, 128
    M = n * S
    st = torch.cuda.current_stream().cuda_stream
    o, d, rad, _ = synthetic.make_rays("blender", n, 1)
    rays = ops.pack_rays(*(torch.from_numpy(x).to(device) for x in (o, d, rad)), 2.0, 6.0)
    t = (2.0 + 4.0 * torch.sort(torch.rand(n, S + 1, device=device), dim=1).values).contiguous()
    tab = ops.ray_table(rays)
    packed = F._packed_weights(model.fine)
    raw = torch.empty(M, 4, device=device)
    L.ddnerf_encode_mlp_bf16_scratch_bytes.restype = C.c_size_t
    scratch = torch.empty(L.ddnerf_encode_mlp_bf16_scratch_bytes(), dtype=torch.uint8, device=device)
    f = L.ddnerf_encode_mlp_bf16_forward
    f.argtypes = [V, V, V, C.c_int, V, C.c_int, C.c_int, V, V]
    L.ddnerf_debug_set_stamps_g2e.argtypes = [V]
    n_cu = torch.cuda.get_device_properties(device).multi_processor_count
    stamps = torch.zeros(n_cu * 6, dtype=torch.int64, device=device)
    if L.ddnerf_debug_set_stamps_g2e(stamps.data_ptr()):
        return None
    launch = lambda: f(tab.data_ptr(), t.data_ptr(), packed.data_ptr(), 0, raw.data_ptr(), n, S, scratch.data_ptr(), st)
    t0, nl = time.time(), 0
    while time.time() - t0 < seconds:
        for _ in range(50):
            launch()
        torch.cuda.synchronize()
        nl += 50
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        launch()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 50
    s = stamps.cpu().numpy().reshape(n_cu, 6).astype(np.float64)
    s = s[s[:, 4] > 0]
    clk = float(np.median((s[:, 2] - s[:, 0]) / (s[:, 3] - s[:, 1]) * 100.0))
    cyc = float(np.median((s[:, 2] - s[:, 0]) / s[:, 4]))
    ideal = 9640 * 16
    return {"kernel": "two groups per weight pass with the encoder inside (mlp_bf16_g2e.hip)",
            "in_kernel_clock_mhz": round(clk), "nominal_clock_mhz": 2400, "tile_samples": 512, "cycles_per_tile": round(cyc),
            "ideal_cycles_per_tile": ideal, "mfma_busy": round(ideal / cyc, 4), "frac_bound_at_this_clock": round(clk / 2400.0, 4),
            "back_to_back_launch_ms": round(ms, 4), "back_to_back_frac": round(FLOP_FINE * M / (ms * 1e-3) / 1e12 / PEAK["bf16"], 4),
            "warm_launches": nl, "how": "diagnostic stamp build of the same kernel sources on 4096 synthetic rays x 128 sorted fenceposts, after the "
            "timed region; cycles_per_tile = a workgroup's tile-loop cycles / its tiles: the MLP's MFMAs, the next tile's encoder in their "
            "gaps and a quarter of the first tile's straight-line encoder (four tiles per workgroup)"}


def kernel_source_digest(mlp):
    """md5 of the kernel's sources; for the generated assembly body of the two-group bf16 kernel: of the generator's OUTPUT (the files
    the build compiles), so that an experiment switch added to the generator does not disown a profile of the unchanged kernel"""
    h = hashlib.md5()
    for f in KERNEL_SOURCES[mlp]:
        h.update(open(os.path.join(ROOT, "ddnerf_amd", "csrc", f), "rb").read())
    return h.hexdigest()


def measured_traffic(args):
    """HBM bytes per launch of the fine-MLP kernel from the committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes
    (profiles/rNN_hbm_traffic_<mlp>.json, the newest round's).  Only quoted when that profile was taken on THIS kernel source (digest recorded
    in the file) and at this launch size; otherwise null."""
    import glob

    # (the 16-bit tiers' fine launch is the fused encoder + MLP kernel unless DDNERF_FUSE_ENCODER=0: the two-launch form has its own file)
    tag = args.mlp + ("_unfused" if args.mlp in ("bf16", "fp16") and os.environ.get("DDNERF_FUSE_ENCODER", "all") == "0" else "")
    found = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_hbm_traffic_%s.json" % tag)))
    if not (found and (args.rays, args.fine) == (4096, 128)):
        return None, None
    tf = found[-1]
    d = json.load(open(tf))
    if d.get("kernel_source_md5") != kernel_source_digest(args.mlp):
        return None, None
    return d.get("fine_mlp_%s_fwd_hbm_bytes_per_launch" % args.mlp, d.get("fine_mlp_%su_fwd_hbm_bytes_per_launch" % args.mlp)), os.path.relpath(tf, ROOT)


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse(argv)
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        launch_ranks(args, argv)
    world = int(env_world or "1")
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (start it as `python bench.py --gpus N` or under a launcher "
                         "with --nproc-per-node N)" % (args.gpus, world))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = world > 1 or args.single_rank_rccl   # (--single-rank-rccl: the N > 1 code path on a one-rank group, for one-GPU boxes)
    # stdout carries ONE JSON line.  Native libraries write there too (RCCL prints a version banner on file descriptor 1 when the
    # process group comes up): from here on descriptor 1 IS stderr, and the line goes out through a private copy of the real stdout.
    sys.stdout.flush()
    real_stdout = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    # The other tiers are measured by child runs BEFORE this process touches the GPU: a child that shares the card with an idle
    # parent context measures the launch-heavy bf16 step up to 1.6x slower than a run of its own (0.89 -> 1.39 ms).
    tiers = {}
    if world == 1 and not dist and args.mode == "render" and args.mlp == "fp32" and args.config == "config_blender.yml" and not args.no_bf16_tier:
        tiers["x3_tier"] = extra_tier(args, "x3")
        tiers["bf16_tier"] = extra_tier(args, "bf16")
        tiers["fp16_tier"] = extra_tier(args, "fp16")
        tiers["train_tier"] = {name: train_tier(args, name) for name in TRAIN_TIERS}
        if not args.image:     # the product-level figure (eval_nerf.py:103-111 times one whole validation image), from child runs
            tiers["image"] = {mlp: image_tier(args, mlp) for mlp in ("fp32", "fp16", "bf16")}

    import torch

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (the HIP path has no CPU fallback)")
    if args.share_gpu:
        local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    backend = None
    if dist:
        import torch.distributed as td

        import datetime

        if env_world is None:   # --single-rank-rccl without a launcher: a rendezvous of one
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", str(free_port()))
        # "nccl" IS RCCL on ROCm; a bounded collective timeout: a rank whose peer died fails instead of waiting 10 minutes
        kw = {"device_id": device} if args.backend == "nccl" else {}
        td.init_process_group(args.backend, rank=rank, world_size=world, timeout=datetime.timedelta(seconds=int(os.environ.get("DDNERF_PG_TIMEOUT_S", "300"))), **kw)
        backend = "%s (RCCL)" % td.get_backend() if args.backend == "nccl" else "%s (rehearsal: not the measured configuration)" % td.get_backend()
    from ddnerf_amd import ops, synthetic

    model, cfg, sd_c, sd_f = build_model(args, device)
    ro, rd, rad, tgt = (torch.from_numpy(x).to(device) for x in synthetic.make_rays(args.ray_kind, args.rays, 1 + rank))
    torch.manual_seed(1234 + rank)

    timer = KernelTimer(only=args.rays * args.fine, every=4 if (args.mlp in ("bf16", "fp16") and args.mode == "render") else 1)
    ops.MLP_LAUNCH_HOOK = timer

    def fence():
        if dist:
            td.barrier()
        torch.cuda.synchronize()

    def timed(step):
        """W untimed steps, then EXACTLY K steps between barrier + synchronize fences; max over ranks."""
        for _ in range(args.ramp + args.warmup):
            step()
        fence()
        timer.pairs.clear()
        timer.seen = 0
        timer.active = True
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        fence()
        dt = time.perf_counter() - t0
        timer.active = False
        if dist:
            t = torch.tensor([dt], dtype=torch.float64, device=device)
            td.all_reduce(t, op=td.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    M_fine = args.rays * args.fine
    res = {}
    if args.mode in ("render", "both"):
        model.eval()

        def render_step():
            with torch.no_grad():
                return model.run_iter(ro, rd, rad, mode="validation", rgb_target=tgt)

        res["render"] = (timed(render_step), timer.mean_ms(M_fine))
    train_skipped = None
    if args.mode in ("train", "both") and args.mlp in ("bf16", "fp16"):
        # there is no bf16 training tier (the reference trains in fp32; ddnerf_amd/functions.py raises)
        if args.mode == "train":
            raise SystemExit("bench.py: training runs on the fp32 / x3 MLP kernels (--mlp fp32|x3)")
        train_skipped = "no training leg: --mlp %s is an inference-only kernel (training runs on --mlp fp32 | x3)" % args.mlp
    elif args.mode in ("train", "both"):
        from ddnerf_amd import train_step

        stepper = train_step.TrainStepper(model, cfg, dist=dist, single_rank_collectives=args.single_rank_rccl)
        res["train"] = (timed(lambda: stepper.step(ro, rd, rad, tgt)), timer.mean_ms(M_fine))
    image = image_pass(args, model, device, fence) if (args.image and "render" in res) else None

    if rank == 0:
        head = "render" if "render" in res else "train"
        dt, (ms, launches) = res[head]
        roof = None
        if ms and head == "render":
            traffic, src = measured_traffic(args)
            # `achieved` / `frac` count ALGORITHMIC FLOP (1,220,608 per sample); the x3 kernel issues three bf16 MFMAs per
            # product: its issued-work figures stand beside them
            ach = M_fine * FLOP_FINE / (ms * 1e-3) / 1e12
            roof = {"bound": "mfma", "kernel": "mlp_%s_fwd_kernel<fine> (%d samples/launch)" % (args.mlp, M_fine),
                    "achieved": round(ach, 2), "peak": PEAK[args.mlp], "unit": "TFLOP/s",
                    "frac": round(ach / PEAK[args.mlp], 4), "traffic": traffic, "traffic_source": src, "traffic_measured_in_run": False,
                    "launch_ms": round(ms, 4), "launches_timed": launches, "launches_in_timed_region": timer.seen}
            if args.mlp == "x3":
                roof["issued_tflops"] = round(3 * ach, 2)
                roof["frac_issued"] = round(3 * ach / PEAK["x3"], 4)
            fused = False
            if args.mlp in ("fp32", "x3"):
                from ddnerf_amd import models as _m
                from ddnerf_amd import ops as _o

                roof["view_dirs_per_ray"] = bool(_m.RAY_DIRS and _o.mlp_rays_supported(args.fine, M_fine))
                if roof["view_dirs_per_ray"]:
                    roof["kernel"] = "mlp_%s_fwd_rays_kernel<fine> (view-direction columns from a per-ray table; %d samples/launch)" % (args.mlp, M_fine)
            if args.mlp in ("bf16", "fp16"):
                from ddnerf_amd import models as _m
                from ddnerf_amd import ops as _o

                fused = (_m.FUSE_ENCODER != "0" and str(cfg.nerf.ray_shape) == "cone" and _o.encode_mlp_bf16_supported(args.fine, M_fine))
                roof["encoder_in_kernel"] = fused
                if fused:
                    # (the launch that is timed also ENCODES its samples -- cast_rays + integrated_pos_enc, models/models.py:117-142 -- and
                    # the unfused path's encode launch is gone from the step; `achieved` / `frac` still count the MLP's FLOP only)
                    roof["kernel"] = "mlp_%sg2e_fwd_kernel<fine> (encoder inside the MLP kernel; %d samples/launch)" % ({"bf16": "bf16", "fp16": "f16"}[args.mlp], M_fine)
            if args.mlp == "fp32" and (args.rays, args.fine) == (4096, 128) and not args.no_clock and cfg.nerf.type == "DDNerfModel":
                roof["clock"] = f32_in_kernel_clock(model, device)
            if args.mlp == "bf16" and (args.rays, args.fine) == (4096, 128) and not args.no_clock:
                roof["clock"] = bf16_fused_in_kernel_clock(model, device) if fused else bf16_in_kernel_clock(model.fine.flat_params().detach(), device)
                if fused:
                    roof["clock_unfused_mlp"] = bf16_in_kernel_clock(model.fine.flat_params().detach(), device, seconds=1.5)
                    roof["unfused_mlp_launch"] = bf16_unfused_mlp_launch(model, device)
                roof["ceiling_same_box"] = bf16_same_box_ceiling(device, roof["frac"])
        elif head == "train":
            roof = train_roofline(args, cfg, dt / args.steps)
        net = "DDNerfModel" if cfg.nerf.type == "DDNerfModel" else "GeneralMipNerfModel (one shared MLP)"
        line = {
            "metric": "rays/sec (4096 rays x 128 samples, 8x256 MLP)",
            "value": round(world * args.rays * args.steps / dt, 1), "unit": "rays/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": DTYPE[args.mlp] if head == "render" else TRAIN_DTYPE[args.mlp], "data": "synthetic",
            "config": {"workload": "%s %s, %d rays/GPU x (%d coarse + %d fine), %s rays, run_iter %s pass"
                                   % (args.config, net, args.rays, args.coarse, args.fine, args.ray_kind, head),
                       "rays_per_gpu": args.rays, "global_rays": world * args.rays, "mode": head, "weights": "seeded, fc_alpha x20",
                       "parallelism": "dp%d (independent ray batches)" % world},
            "roofline": roof,
        }
        if dist:
            line["rccl_ranks"] = td.get_world_size()
            line["backend"] = backend
        if train_skipped:
            line["train"] = {"skipped": train_skipped}
        if image:
            line["image"] = image
        if head == "render" and "train" in res:
            tdt = res["train"][0]
            line["train"] = {"value": round(world * args.rays * args.steps / tdt, 1), "unit": "rays/s",
                             "ms_per_step": round(tdt / args.steps * 1e3, 4), "dtype": TRAIN_DTYPE[args.mlp],
                             "collective": "one all-reduce of the flat fp32 gradient buffer per network per step" if dist else None,
                             "roofline": train_roofline(args, cfg, tdt / args.steps)}
        line.update(tiers)
        if world == 1 and not args.no_cpu_baseline and head == "render":
            line["cpu_baseline"] = cpu_baseline(args, cfg, sd_c, sd_f, check_model=model)
        real_stdout.write(json.dumps(line) + "\n")
        real_stdout.flush()
    if dist:
        td.destroy_process_group()


def image_pass(args, model, device, fence):
    """One full validation image through run_iter, all its ray chunks (eval_nerf.py:103-111 times exactly this call): the product-level
    figure beside the per-chunk one.  Rays of the image are resident in HBM; 2 untimed images, then 3 timed ones."""
    import torch

    from ddnerf_amd import synthetic

    h, w = args.image
    ro, rd, rad, _ = (torch.from_numpy(x).to(device) for x in synthetic.make_rays(args.ray_kind, h * w, 7))
    ro, rd, rad = ro.view(h, w, 3), rd.view(h, w, 3), rad.view(h, w, 1)

    def one():
        with torch.no_grad():
            return model.run_iter(ro, rd, rad, mode="validation")

    for _ in range(2):
        one()
    fence()
    t0 = time.perf_counter()
    n = 3
    for _ in range(n):
        out = one()
    fence()
    dt = (time.perf_counter() - t0) / n
    chunk = int(getattr(model.cfg.nerf.validation, "chunksize", 0) or 0)
    return {"size": "%dx%d" % (h, w), "rays": h * w, "s_per_image": round(dt, 5), "rays_per_s": round(h * w / dt, 1),
            "chunks": -(-h * w // chunk) if chunk else None, "chunk_rays": chunk or None, "images_timed": n,
            "rgb_shape": list(out[len(out) - 1]["rgb"].shape)}


def bf16_same_box_ceiling(device, kernel_frac, seconds=0.45):
    """What bare bf16 MFMA loops sustain on THIS device, right behind the timed region (the chip is power-limited on them and devices
    differ by several per cent): the diagnostic library's ddnerf_debug_mfma_ceiling (csrc/mfma_ceiling.hip) -- 16x16x32 MFMAs at the
    MLP kernels' per-wave tile with operands in registers, with the A fragments from LDS at the kernels' ratio, and with the kernels'
    weight staging (LDS-DMA at their rate, a barrier per period) on top -- each for `seconds` of back-to-back launches on uniform(-1, 1)
    bf16 operands; fractions of the nominal 2.5 PFLOP/s."""
    import ctypes as C

    import numpy as np
    import torch

    from ddnerf_amd import build as hip_build

    if not os.path.exists(hip_build.DIAG_SO):
        return None
    L = C.CDLL(hip_build.DIAG_SO)
    if not hasattr(L, "ddnerf_debug_mfma_ceiling"):
        return None
    V = C.c_void_p
    L.ddnerf_debug_mfma_ceiling_src_bytes.restype = C.c_size_t
    nb = L.ddnerf_debug_mfma_ceiling_src_bytes()
    f = L.ddnerf_debug_mfma_ceiling
    f.argtypes = [C.c_int, V, V, C.c_int, V, V]
    n_cu = torch.cuda.get_device_properties(device).multi_processor_count
    src = (torch.rand(nb // 2, device=device) * 2 - 1).to(torch.bfloat16).contiguous()
    out = torch.empty(n_cu * 256, device=device)
    stamps = torch.zeros(n_cu * 2, dtype=torch.int64, device=device)
    st = torch.cuda.current_stream().cuda_stream
    iters = 20000                                   # x 96 MFMAs x 16 cycles = 30.7 M cycles ~ 15 ms per launch
    flop = n_cu * 4 * iters * 96 * (16 * 16 * 32 * 2)
    res = {}
    for mode, name in ((0, "registers"), (1, "lds_fed"), (2, "lds_fed_staged")):
        t0 = time.time()
        while time.time() - t0 < seconds:
            for _ in range(4):
                if f(mode, src.data_ptr(), out.data_ptr(), iters, stamps.data_ptr(), st):
                    return None
            torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(4):
            f(mode, src.data_ptr(), out.data_ptr(), iters, stamps.data_ptr(), st)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 4
        s = stamps.cpu().numpy().reshape(n_cu, 2).astype(np.float64)
        res[name] = {"frac": round(flop / (ms * 1e-3) / 1e12 / PEAK["bf16"], 4), "in_kernel_clock_mhz": round(float(np.median(s[:, 0] / s[:, 1] * 100.0))),
                     "mfma_busy": round(iters * 96 * 16 / float(np.median(s[:, 0])), 4)}
    res["kernel_frac_over_staged_ceiling"] = round(kernel_frac / res["lds_fed_staged"]["frac"], 4)
    res["how"] = ("bare v_mfma_f32_16x16x32_bf16 loops (csrc/mfma_ceiling.hip, diagnostic library), one wave per SIMD on every CU, uniform(-1, 1) bf16 "
                  "operands, %.2f s of back-to-back launches each, on this device right after the timed region; fractions of 2500 TFLOP/s" % seconds)
    return res


def train_roofline(args, cfg, step_s):
    """Whole-step MFMA roofline of a training step: the time the step's algorithmic FLOP would take at the dense MFMA peak of
    the unit each pass runs on, over the measured step time.  Forward and backward-data (no input gradient for the first
    layer, the skip's xyz columns and the dir columns) run on the fp32 matrix cores (--mlp fp32) or as three bf16 MFMAs per
    product (--mlp x3); the weight gradients always run as three bf16 MFMAs per product."""
    dd = cfg.nerf.type == "DDNerfModel"
    m_c, m_f = args.rays * args.coarse, args.rays * args.fine
    f_c, f_f = (FLOP_COARSE_DD if dd else FLOP_FINE), FLOP_FINE
    first = 2 * (96 * 256 + 96 * 256 + 27 * 128)
    fwd = m_c * f_c + m_f * f_f
    bwd = m_c * (f_c - first) + m_f * (f_f - first)
    wgrad = fwd
    # algorithmic: every product once, on the unit it runs on; issued: the hi/lo-split passes issue three bf16 MFMAs per product
    if args.mlp == "x3":
        ideal = (fwd + bwd + wgrad) / (PEAK["x3"] * 1e12)
        issued = (3 * (fwd + bwd) + wgrad) / (PEAK["x3"] * 1e12)
    else:
        ideal = (fwd + bwd) / (PEAK["fp32"] * 1e12) + wgrad / (PEAK["x3"] * 1e12)
        issued = (fwd + bwd) / (PEAK["fp32"] * 1e12) + 3 * wgrad / (PEAK["x3"] * 1e12)
    from ddnerf_amd import ops as _o

    records = ("bf16 row pairs and 16-bit sign words (the x3 tier's own kernels)" if args.mlp == "x3" else
               {"x3": "blocked records of the fp32 values, split into bf16 hi / lo by the weight-gradient kernel; 1-bit ReLU masks through scalar memory (round 5)",
                "x3words": "blocked records of hi/lo words split by the recording kernels (round 4's form)",
                "pairs": "bf16 row pairs (opt-in speed mode, not fp32-class)", "f32": "fp32 [feature][sample] matrices"}[_o.WGRAD_MODE])
    return {"bound": "mfma", "kernel": "whole training step (forward + backward-data + weight gradients of both networks)",
            "records": records, "wgrad_mode": _o.WGRAD_MODE if args.mlp == "fp32" else None,
            "algorithmic_flop_per_step": fwd + bwd + wgrad, "ideal_ms": round(ideal * 1e3, 4), "step_ms": round(step_s * 1e3, 4),
            "frac": round(ideal / step_s, 4), "frac_issued": round(issued / step_s, 4), "issued_ideal_ms": round(issued * 1e3, 4),
            "unit": "fraction of the step time the (algorithmic / issued) FLOP need at the dense MFMA peaks",
            "peaks_tflops": {"fp32_mfma": PEAK["fp32"], "bf16_mfma": PEAK["x3"]},
            "note": ("forward / backward-data as three bf16 MFMAs per product, weight gradients as one (bf16-rounded records)" if args.mlp == "x3" else
                     "forward / backward-data on the fp32 matrix cores, weight gradients as three bf16 MFMAs per product")}


if __name__ == "__main__":
    main()
