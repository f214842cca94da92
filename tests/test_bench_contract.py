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
    assert r["traffic"] is None or r["traffic"] > 2.7e8        # at least the algorithmic bytes
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
    assert tt["x3"]["value"] > tt["fp32"]["value"] > 0
    assert tt["x3"]["roofline"]["frac"] < tt["x3"]["roofline"]["frac_issued"] < 3 * tt["x3"]["roofline"]["frac"]


def test_train_line():
    d = _bench("--mode", "train", "--mlp", "x3", "--no-cpu-baseline")
    assert d["config"]["mode"] == "train" and d["value"] > 0 and "cpu_baseline" not in d and "x3_tier" not in d
