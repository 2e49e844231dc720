"""GPU: the one-line JSON contract of bench.py (metric / value / ms_per_step / roofline / cpu_baseline ...) on a small configuration,
single rank and two ranks (torch.distributed.run, gloo, both ranks on device 0: the multi-rank path of bench.py on one GPU)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def parse(out):
    assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-3000:])
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]                      # exactly ONE JSON line
    return json.loads(lines[0])


def run_bench(*extra):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "small", "--steps", "3", "--warmup", "1", *extra],
                         capture_output=True, text=True, timeout=900, cwd=ROOT)
    return parse(out)


def run_bench_2ranks(*extra):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--config", "small", "--steps", "3",
                          "--warmup", "1", "--dist-backend", "gloo", "--force-device", "0", *extra],
                         capture_output=True, text=True, timeout=1200, cwd=ROOT, env=env)
    return parse(out)


def test_bench_line_contract():
    d = run_bench("--batch", "4", "--group", "2", "--lanes", "2", "--batched-passes", "3", "--cpu-samples", "2")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
              "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["dtype"] == "f64" and d["data"] == "synthetic" and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"]
    # headline: ONE system stepped sequentially => value = steps / time of the timed region
    assert abs(d["value"] - 3 / (d["ms_per_step"] * 3e-3)) <= 1e-6 * d["value"]
    b = d["config"]["batched"]
    assert b["instances_per_gpu"] == 4 and b["instances_per_group"] == 2 and b["passes"] == 3
    assert abs(b["newton_steps_per_s"] - 4 * 3 / (b["ms_per_pass"] * 3e-3)) <= 1e-6 * b["newton_steps_per_s"]
    assert abs(b["problems_per_s_of_10_steps"] - b["newton_steps_per_s"] / 10.0) < 1e-9
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["peak"] > 0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert r["achieved"] > 0 and "traffic" in r and r["instances_per_launch"] == 1
    assert 10.0 < r["peak_measured"] < r["peak"]                      # the measured fp64 MFMA ceiling of this chip
    assert r["group_launch"]["instances_per_launch"] == 2 and r["group_launch"]["achieved"] > 0
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 1 and c["value"] > 0 and c["unit"] == d["unit"] and isinstance(c["sample"], str)
    assert len(c["samples_s"]) == 2
    assert 0 < c["B0_ii_reference_refactorisation"]["value"] < c["value"] and c["B0_ii_reference_refactorisation"]["factorizations_per_step"] >= 3
    assert c["B1_lapack_all_cores"]["value"] > 0 and c["B1_lapack_all_cores"]["cores"] >= 1
    assert d["value"] > c["value"]                                   # the device path is faster than the single-core port


def test_bench_single_units_and_no_baseline():
    d = run_bench("--batch", "2", "--group", "1", "--lanes", "2", "--no-cpu-baseline", "--batched-passes", "2")
    assert d["cpu_baseline"] is None and d["config"]["batched"]["instances_per_group"] == 1 and d["value"] > 0


def test_bench_two_ranks_on_one_gpu():
    """the multi-rank path of bench.py (barrier, max-over-ranks time, gather of status rows, all-reduce of counters) on ONE GPU: two
    ranks under torch.distributed.run with the gloo backend, both on device 0.  value = sum over ranks; the two ranks share the GPU,
    so the aggregate is between 1x and ~2x the single-rank figure (sharing penalty), never more"""
    one = run_bench("--batch", "4", "--group", "2", "--lanes", "2", "--no-cpu-baseline", "--batched-passes", "3")
    two = run_bench_2ranks("--batch", "4", "--group", "2", "--lanes", "2", "--batched-passes", "3")
    assert two["n_gpus"] == 2 and two["cpu_baseline"] is None and two["steps"] == 3
    assert abs(two["value"] - 2 * 3 / (two["ms_per_step"] * 3e-3)) <= 1e-6 * two["value"]         # both ranks' steps over the max time
    b1, b2 = one["config"]["batched"], two["config"]["batched"]
    assert b2["instances_per_gpu"] == 4 and abs(b2["newton_steps_per_s"] - 2 * 4 * 3 / (b2["ms_per_pass"] * 3e-3)) <= 1e-6 * b2["newton_steps_per_s"]
    # (bench.py itself asserts that the gathered status table has 2 x B rows, all ok, and that the counters sum to 2 B P)
    assert 0.2 * b1["newton_steps_per_s"] <= b2["newton_steps_per_s"] <= 2.3 * b1["newton_steps_per_s"]      # (a timing ratio: loose, the box may be shared with other test processes)
    assert 0.5 * one["value"] <= two["value"] <= 2.3 * one["value"]
