"""GPU box: bench.py itself, end to end on a small grid -- the JSON line the driver parses must carry every field of the
contract, with live PMC traffic (rocprofv3 child processes), clocks and the CPU baseline."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*args):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, f"bench.py must print ONE line, got {len(lines)}"
    return json.loads(lines[0])


def test_bench_line_carries_the_contract_fields():
    d = _run("--grid", "4096", "--steps", "20", "--warmup", "5", "--repeats", "5", "--no-secondary")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline", "gpu_state"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 20 and d["warmup"] == 5 and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert d["dtype"] == "f32" and d["data"] == "synthetic" and "workload" in d["config"] and d["repeats"] == 5
    assert d["value"] > 1e5 and abs(d["ms_per_step"] * d["value"] * 1e6 / 1e3 - 4096 * 4096) / (4096 * 4096) < 0.02
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_ms", "steady_state", "overfetch", "kernel"):
        assert k in r, k
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and r["unit"] == "GB/s"
    if r["traffic"] is not None:          # live PMC worked: the fraction is the per-launch bytes over the launch time
        assert abs(r["traffic"] / (r["avg_launch_ms"] * 1e-3) / 1e9 / 8000.0 - r["frac"]) < 0.01
        assert 0.9 < r["overfetch"] < 3 and 0.05 < r["frac"] < 1.0
    else:
        assert "traffic_note" in r
    assert r["steady_state"]["steps_per_launch"] == 16 and r["steady_state"]["avg_launch_ms"] > 0
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] == "port" and c["cores"] == 1 and c["value"] > 1


def test_bench_other_dtype_boundary_and_no_pmc():
    d = _run("--grid", "2048", "--cols", "4096", "--steps", "16", "--warmup", "0", "--repeats", "3", "--dtype", "f64", "--pmc", "off",
             "--no-secondary", "--no-cpu-baseline")
    assert d["dtype"] == "f64" and d["roofline"]["traffic"] is None and d["value"] > 1e4
    d = _run("--grid", "2048", "--cols", "4096", "--steps", "32", "--warmup", "16", "--repeats", "3", "--boundary", "pml", "--pmc", "off",
             "--no-secondary", "--no-cpu-baseline")
    assert d["config"]["boundary"] == "pml" and d["value"] > 1e4
