"""CPU-side checks of bench.py's launcher (no GPU): --gpus N builds N rank environments, refuses to pretend when the node
has fewer GPUs, and the BASELINE configs map to the right workload."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_rank_environments_one_per_gpu():
    envs = bench.rank_environments(8, 29999)
    assert len(envs) == 8
    assert [e["RANK"] for e in envs] == [str(r) for r in range(8)] == [e["LOCAL_RANK"] for e in envs]
    assert all(e["WORLD_SIZE"] == "8" and e["MASTER_ADDR"] == "127.0.0.1" and e["MASTER_PORT"] == "29999" for e in envs)
    assert all(e["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" for e in envs)     # dmabuf IPC only on this pool (RCCL needs it)


def test_gpus_flag_is_not_silently_ignored():
    # no GPU in this container: asking for two must fail loudly, not run one rank and print n_gpus 1
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"], capture_output=True, text=True, env=env, timeout=300)
    assert p.returncode != 0 and "only" in p.stderr and "GPU" in p.stderr and not p.stdout.strip()
    # under a launcher the world size must agree with --gpus
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "1"], capture_output=True, text=True,
                       env=dict(env, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"), timeout=300)
    assert p.returncode != 0 and "WORLD_SIZE=2" in p.stderr


def test_config_workloads():
    a = bench.parse([])
    assert (a.config, a.rays, a.mlp, a.mode, a.scaling, a.ray_kind) == ("config_blender.yml", 4096, "fp32", "render", "weak", "blender")
    a = bench.parse(["--config", "config_ff.yml"])
    assert (a.mlp, a.ray_kind, a.rays) == ("bf16", "llff", 4096)                      # BASELINE configs[2]: NDC rays, bf16 MFMA MLP
    a = bench.parse(["--config", "config_360.yml", "--gpus", "8", "--global-rays", "8192"])
    assert (a.rays, a.scaling, a.mode, a.ray_kind) == (1024, "strong", "both", "real360")   # configs[3], strong scaling
    a = bench.parse(["--config", "config_360.yml", "--gpus", "8"])
    assert (a.rays, a.scaling) == (8192, "weak")
    a = bench.parse(["--config", "config_blender_mipnerf.yml", "--gpus", "2"])
    assert (a.rays, a.mode, a.mlp) == (4096, "both", "fp32")
    # configs[2] under the driver's N > 1 launch: the bf16 kernel, both legs asked for (bench.main prints the render leg and
    # says that the inference-only kernel has no training leg: tests/test_bench_contract.py runs it)
    a = bench.parse(["--config", "config_ff.yml", "--gpus", "8"])
    assert (a.mlp, a.mode, a.rays, a.scaling) == ("bf16", "both", 4096, "weak")
    # the one-GPU rehearsals of the N > 1 path
    a = bench.parse(["--single-rank-rccl"])
    assert (a.gpus, a.mode, a.backend) == (1, "both", "nccl")
    a = bench.parse(["--gpus", "2", "--backend", "gloo", "--share-gpu"])
    assert a.share_gpu and a.backend == "gloo" and a.mode == "both"
    a = bench.parse(["--image", "800x800"])
    assert a.image == (800, 800)


def _child(code):
    return subprocess.Popen([sys.executable, "-c", code], stdout=subprocess.PIPE, text=True)


def test_supervise_kills_the_siblings_of_a_failed_rank():
    # rank 1 dies before "rendezvous"; ranks 0 and 2 would wait for it for ten minutes
    import time

    t0 = time.monotonic()
    procs = [_child("import time; print('rank0 up', flush=True); time.sleep(600)"),
             _child("import sys; sys.exit(3)"),
             _child("import time; time.sleep(600)")]
    codes, out, reason = bench.supervise(procs, timeout_s=120)
    assert time.monotonic() - t0 < 30
    assert codes[1] == 3 and codes[0] != 0 and codes[2] != 0 and all(c is not None for c in codes)
    assert "rank 1 exited with code 3" in reason and "rank0 up" in out


def test_supervise_time_limit_and_success():
    import time

    t0 = time.monotonic()
    procs = [_child("import time; time.sleep(600)"), _child("import time; time.sleep(600)")]
    codes, out, reason = bench.supervise(procs, timeout_s=1.0)
    assert time.monotonic() - t0 < 30 and "no result after" in reason and all(c not in (None, 0) for c in codes)
    # a child that ignores SIGTERM is killed
    procs = [_child("import signal, time; signal.signal(signal.SIGTERM, signal.SIG_IGN); print('x', flush=True); time.sleep(600)")]
    time.sleep(0.5)
    codes, out, reason = bench.supervise(procs, timeout_s=0.5)
    assert codes[0] == -9 and reason
    # the good case: every rank exits 0, rank 0's output is returned whole (more than a pipe buffer of it)
    procs = [_child("print('{\"metric\": 1}'); print('y' * 200000)"), _child("pass")]
    codes, out, reason = bench.supervise(procs, timeout_s=60)
    assert codes == [0, 0] and reason is None and out.startswith('{"metric": 1}') and len(out) > 200000


def test_free_port_is_bindable():
    import socket

    port = bench.free_port()
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", port))
