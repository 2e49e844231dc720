"""GPU: the one-line JSON contract of bench.py (metric / value / ms_per_step / roofline / cpu_baseline ...) on a small configuration."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*extra):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "small", "--steps", "3", "--warmup", "1", *extra],
                         capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]                      # exactly ONE JSON line
    return json.loads(lines[0])


def test_bench_line_contract():
    d = run_bench("--batch", "4", "--group", "2", "--lanes", "2")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
              "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["dtype"] == "f64" and d["data"] == "synthetic" and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"]
    # value = instances * steps / time of the timed region
    assert abs(d["value"] - 4 * 3 / (d["ms_per_step"] * 3e-3)) <= 1e-6 * d["value"]
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["peak"] > 0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert r["achieved"] > 0 and "traffic" in r
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 1 and c["value"] > 0 and c["unit"] == d["unit"] and isinstance(c["sample"], str)
    assert d["value"] > c["value"]                                   # the device path is faster than the single-core port


def test_bench_single_units_and_no_baseline():
    d = run_bench("--batch", "2", "--group", "1", "--lanes", "2", "--no-cpu-baseline")
    assert d["cpu_baseline"] is None and d["config"]["instances_per_group"] == 1 and d["value"] > 0
