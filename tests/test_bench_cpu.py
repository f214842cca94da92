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
