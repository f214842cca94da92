"""bench.py's one-line JSON contract (driver-facing) on a real GPU: a short run must print exactly one JSON line with the
metric fields, a `roofline` object for the fine-MLP kernel and, at N = 1, a `cpu_baseline` object and the tier objects."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*extra):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", *extra],
                       capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    return json.loads(lines[0])


def test_render_line_has_the_contract_fields():
    d = _bench("--cpu-rays", "64")
    for k, t in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int),
                 ("ms_per_step", float), ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str),
                 ("config", dict), ("roofline", dict), ("cpu_baseline", dict)):
        assert isinstance(d[k], t), (k, d[k])
    assert d["vs_baseline"] is None and d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1
    assert d["unit"] == "rays/s" and d["scaling"] == "weak" and d["higher_is_better"] is True and "workload" in d["config"]
    assert abs(d["value"] - 4096 / (d["ms_per_step"] * 1e-3)) <= 1e-3 * d["value"]
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["peak"] == 157.3
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and 0.3 < r["frac"] < 1.0 and r["launches_timed"] == 3
    assert r["traffic"] is None or r["traffic"] > 2.1e8        # at least the algorithmic bytes (384-byte rows + the per-ray table + outputs)
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == "rays/s" and c["cores"] >= 1 and c["value"] > 0 and "sample" in c
    for tier, lo in (("x3_tier", 0.1), ("bf16_tier", 0.2)):   # `frac` counts algorithmic FLOP: the x3 kernel issues 3x that
        t = d[tier]
        assert t["value"] > d["value"] and t["roofline"]["peak"] == 2500.0 and lo < t["roofline"]["frac"] < 1.0, tier
    x3 = d["x3_tier"]["roofline"]
    assert abs(x3["frac_issued"] - 3 * x3["frac"]) < 2e-3 and 0.3 < x3["frac_issued"] < 1.0
    # the bf16 line carries what its fraction is made of: the in-kernel clock and the matrix-pipe busy share (diagnostic build)
    clk = d["bf16_tier"]["roofline"]["clock"]
    assert 1000 < clk["in_kernel_clock_mhz"] <= 2500 and 0.5 < clk["mfma_busy"] <= 1.0
    assert abs(clk["mfma_busy"] - clk["ideal_cycles_per_tile"] / clk["cycles_per_tile"]) < 1e-3
    tt = d["train_tier"]
    assert tt["x3"]["value"] > tt["x3_exact"]["value"] > tt["fp32_pairs"]["value"] > tt["fp32"]["value"] > 0
    assert tt["x3"]["roofline"]["frac"] < tt["x3"]["roofline"]["frac_issued"] < 3 * tt["x3"]["roofline"]["frac"]
    assert "NOT fp32-class" in tt["fp32_pairs"]["dtype"] and "fp32-class" in tt["x3_exact"]["dtype"]
    # round 4: the fp16 tier; the same-box bare-loop ceiling beside the bf16 fraction; the image-level figure; the self-check of the
    # timed path against the CPU oracle
    f16 = d["fp16_tier"]
    assert f16["value"] > 2 * d["x3_tier"]["value"] and 0.3 < f16["roofline"]["frac"] < 1.0 and "fp16" in f16["dtype"]
    ceil = d["bf16_tier"]["roofline"]["ceiling_same_box"]
    assert 0.5 < ceil["lds_fed_staged"]["frac"] <= ceil["lds_fed"]["frac"] + 0.03 and ceil["lds_fed"]["frac"] <= ceil["registers"]["frac"] + 0.03 < 1.03
    assert abs(ceil["kernel_frac_over_staged_ceiling"] - d["bf16_tier"]["roofline"]["frac"] / ceil["lds_fed_staged"]["frac"]) < 2e-3
    assert d["roofline"]["traffic_measured_in_run"] is False
    # round 5: the headline line proves its own fraction from in-kernel clock stamps; the bf16 step runs the fused encoder + MLP kernel and
    # carries the same MLP as a launch of its own beside it
    hc = d["roofline"]["clock"]
    assert 2000 < hc["in_kernel_clock_mhz"] <= 2500 and 0.9 < hc["mfma_share_of_workgroup_cycles"] <= 1.0 and hc["closes_within"] < 0.02, hc
    br = d["bf16_tier"]["roofline"]
    assert br["encoder_in_kernel"] is True and "g2e" in br["kernel"] and "encoder inside" in clk["kernel"]
    # (a 16-bit tier times every fourth fine launch of its timed region: an event pair is 12 us of idle GPU around the launch it brackets)
    assert br["launches_in_timed_region"] == d["steps"] and br["launches_timed"] == (d["steps"] + 3) // 4
    assert d["roofline"]["launches_in_timed_region"] == d["roofline"]["launches_timed"] == d["steps"]
    um = br["unfused_mlp_launch"]
    assert 0.3 < um["frac"] < 1.0 and um["launch_ms"] > 0 and um["encode_launch_ms"] > 0 and um["launch_ms"] + um["encode_launch_ms"] > br["launch_ms"]
    # ... the fp16 tier its twin; the fp32 training step the values records with the 1-bit ReLU masks (the word records are an A/B mode)
    assert d["fp16_tier"]["roofline"]["encoder_in_kernel"] is True and "f16g2e" in d["fp16_tier"]["roofline"]["kernel"]
    tr = tt["fp32"]["roofline"]
    assert tr["wgrad_mode"] == "x3" and "fp32 values" in tr["records"] and "scalar memory" in tr["records"], tr
    assert tt["fp32_pairs"]["roofline"]["wgrad_mode"] == "pairs" and tt["x3"]["roofline"]["wgrad_mode"] is None
    for tier in ("fp32", "fp16", "bf16"):
        im = d["image"][tier]
        assert im["size"] == "800x800" and im["chunks"] == 40 and im["rays_per_s"] > 0 and 0.9 <= im["image_over_chunk_rate"] <= 1.1, (tier, im)
    pc = c["parity_check"]
    assert pc["within_1e-4"] is True and max(pc["max_abs_err_vs_cpu_oracle"].values()) <= 1e-4, pc


def test_train_line():
    d = _bench("--mode", "train", "--mlp", "x3", "--no-cpu-baseline")
    assert d["config"]["mode"] == "train" and d["value"] > 0 and "cpu_baseline" not in d and "x3_tier" not in d


def test_multi_rank_branch_on_a_one_rank_rccl_group():
    """bench.py's N > 1 code path -- init_process_group("nccl", device_id=...), the barrier fences, the MAX all-reduce of the timed
    region, the train leg with its in-backward gradient all-reduce, `rccl_ranks` / `backend` / `train` in the line -- executed on the
    box's one GPU as a one-rank RCCL group (--single-rank-rccl), in --mode both: the first 8-GPU run must not be the first execution
    of these lines."""
    d = _bench("--single-rank-rccl", "--no-cpu-baseline", "--mlp", "x3")
    assert d["rccl_ranks"] == 1 and "nccl" in d["backend"] and "RCCL" in d["backend"] and d["n_gpus"] == 1
    assert d["config"]["mode"] == "render" and d["roofline"]["launches_timed"] == 3
    t = d["train"]
    assert t["value"] > 0 and t["ms_per_step"] > d["ms_per_step"] and "all-reduce" in t["collective"]
    assert 0 < t["roofline"]["frac"] < t["roofline"]["frac_issued"] < 1.0


def test_bf16_config_in_both_mode_reports_the_render_leg():
    """config_ff.yml defaults to the inference-only bf16 kernel: `--mode both` (the N > 1 default) must print the render line and say
    that there is no training leg, not exit after the render leg was measured."""
    d = _bench("--single-rank-rccl", "--no-cpu-baseline", "--config", "config_ff.yml", "--no-clock")
    assert d["config"]["mode"] == "render" and d["value"] > 0 and d["roofline"]["peak"] == 2500.0
    assert "skipped" in d["train"] and "inference-only" in d["train"]["skipped"]


def test_two_ranks_through_the_launcher_on_one_gpu():
    """`bench.py --gpus 2` starts and supervises two ranks itself; on a one-GPU box both use GPU 0 over gloo (--share-gpu: RCCL
    refuses two ranks on one device).  The launcher, the rendezvous, the fences and the MAX reduce across two real processes."""
    d = _bench("--gpus", "2", "--backend", "gloo", "--share-gpu", "--rays", "1024", "--mlp", "x3")
    assert d["n_gpus"] == 2 and d["rccl_ranks"] == 2 and "gloo" in d["backend"] and "rehearsal" in d["backend"]
    assert d["config"]["global_rays"] == 2048 and d["config"]["rays_per_gpu"] == 1024
    assert abs(d["value"] - 2048 / (d["ms_per_step"] * 1e-3)) <= 1e-3 * d["value"]
    assert d["train"]["value"] > 0 and "all-reduce" in d["train"]["collective"] and "cpu_baseline" not in d


def test_image_level_number():
    """--image HxW: one full validation image through run_iter (all its ray chunks), seconds per image beside the per-chunk
    figure; the chunk loop must not cost the image more than a few per cent of the per-chunk rate."""
    d = _bench("--image", "400x400", "--no-cpu-baseline", "--no-bf16-tier", "--rays", "16384")
    im = d["image"]
    assert im["size"] == "400x400" and im["rays"] == 160000 and im["chunks"] == 10 and im["rgb_shape"] == [400, 400, 3]
    assert abs(im["rays_per_s"] - 160000 / im["s_per_image"]) <= 1e-2 * im["rays_per_s"]
    assert im["rays_per_s"] >= 0.9 * d["value"], (im, d["value"])


def test_two_ranks_under_torch_distributed_run_on_one_gpu():
    """the driver's N > 1 command line -- `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P bench.py --gpus N ...` -- with N = 2 ranks sharing the box's one GPU over gloo: bench.py must take RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_* from the launcher's environment (not start ranks of its own) and rank 0 must print exactly one JSON line"""
    import socket

    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--backend", "gloo", "--share-gpu", "--rays", "1024", "--mlp", "x3"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["rccl_ranks"] == 2 and d["config"]["global_rays"] == 2048 and d["train"]["value"] > 0


def test_config_360_strong_scaling_command_line_on_one_gpu():
    """BASELINE configs[3] -- config_360.yml, 8192 rays split over the ranks (--global-rays: strong scaling), data-parallel with the
    gradient all-reduce -- as two supervised ranks sharing the box's one GPU over gloo: the first real 8-GPU run of that configuration
    must not be the first execution of this argument path (round-4 review, item 9)."""
    d = _bench("--gpus", "2", "--global-rays", "8192", "--config", "config_360.yml", "--backend", "gloo", "--share-gpu")
    assert d["n_gpus"] == 2 and d["rccl_ranks"] == 2 and d["scaling"] == "strong"
    assert d["config"]["global_rays"] == 8192 and d["config"]["rays_per_gpu"] == 4096 and "config_360.yml" in d["config"]["workload"]
    assert abs(d["value"] - 8192 / (d["ms_per_step"] * 1e-3)) <= 1e-3 * d["value"]
    assert d["train"]["value"] > 0 and "all-reduce" in d["train"]["collective"]
