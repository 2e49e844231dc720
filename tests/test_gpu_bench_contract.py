"""GPU: the one-line JSON contract of bench.py (metric / value / ms_per_step / roofline / cpu_baseline ...) on a small configuration,
single rank and two ranks (both ranks on device 0 with the gloo backend: the multi-rank path of bench.py on one GPU), started both ways:
under torch.distributed.run and by `python bench.py --gpus 2` alone (bench.py then spawns the ranks itself).  Structural asserts only: rates
are reported, never compared (a wall-clock ratio does not belong in a correctness suite)."""
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


def run_bench_2ranks(*extra, torchrun=True):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    args = [os.path.join(ROOT, "bench.py"), "--gpus", "2", "--config", "small", "--steps", "3", "--warmup", "1", "--dist-backend", "gloo",
            "--force-device", "0", *extra]
    if torchrun:
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port)] + args
    else:
        cmd = [sys.executable] + args                                # the driver's plain command form: bench.py starts the ranks itself
        for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
            env.pop(k, None)
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=1200, cwd=ROOT, env=env)
    return parse(out)


def check_roofline_entry(e, inst):
    assert e["bound"] == "mfma" and e["unit"] == "TFLOP/s" and e["peak"] > 0 and abs(e["frac"] - e["achieved"] / e["peak"]) < 1e-12
    assert e["achieved"] > 0 and "traffic" in e and e["instances_per_launch"] == inst and e["launches_per_step"] >= 1
    assert abs(e["avg_launch_ms"] * e["launches_per_step"] - e["ms_per_step"]) <= 1e-9 * e["ms_per_step"]
    assert abs(e["achieved"] - e["flops_per_launch"] * e["launches_per_step"] / (e["ms_per_step"] * 1e-3) * 1e-12) <= 1e-9 * e["achieved"]


def test_bench_line_contract():
    d = run_bench("--batch", "4", "--group", "2", "--lanes", "2", "--batched-passes", "3", "--cpu-samples", "2",
                  "--c4-configs", "smallT", "--c4-batch", "4", "--c4-group", "2")
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
    assert "calipso_hip_comm" in b["post_round_exchange"]              # one rank: the gather goes through the product's RCCL entry points
    # roofline = the kernel with the largest share of the step; the other matrix-core kernel is listed under `secondary`
    r = d["roofline"]
    check_roofline_entry(r, 1)
    lfac = r["kernel"].startswith("k_lfac")      # one dense system alone (NP >= 1024): the Schur complement's products ride in the panel launches, ONE matrix-core kernel (csrc/lfac.hip)
    if lfac:
        assert len(r["secondary"]) == 1 and r["flops_schur"] > 0 and r["flops_ldl"] > 0 and r["schedule_buffers_bytes"] > 0
        assert abs(r["flops_per_launch"] * r["launches_per_step"] - (r["flops_schur"] + r["flops_ldl"])) <= 1e-6 * r["flops_schur"]
    else:
        assert len(r["secondary"]) == 2                               # the other matrix-core kernel, then the HBM-bound mat-vec of the refinement residual
        check_roofline_entry(r["secondary"][0], 1)
        assert r["ms_per_step"] >= r["secondary"][0]["ms_per_step"]
        assert {r["kernel"].split(" ")[0], r["secondary"][0]["kernel"].split(" ")[0]} == {"k_ldl_step", "k_schur"}
    hb = r["secondary"][-1]
    assert hb["bound"] == "hbm" and hb["unit"] == "GB/s" and hb["peak"] == 8000.0 and hb["kernel"].startswith("k_gemv_t2_and_n") and "traffic" in hb
    assert abs(hb["achieved"] - hb["bytes_per_launch"] / (hb["avg_launch_ms"] * 1e-3) * 1e-9) <= 1e-9 * hb["achieved"] and abs(hb["frac"] - hb["achieved"] / 8000.0) < 1e-12
    assert "cpu_baseline_rows" in d["config"] and "NOT in this line" in d["config"]["cpu_baseline_rows"]
    assert 10.0 < r["peak_measured"] < r["peak"]                      # the measured fp64 MFMA ceiling of this chip
    for name in ("k_ldl_step", "k_schur"):
        check_roofline_entry(r["group_launch"][name], 2)
    ph = d["config"]["roofline_phases"]["single_system"]
    assert ph["solve_and_refine"]["bytes_executed"] > 0 and ph["solve_and_refine"]["frac_executed"] > 0 and ph["factor"]["frac_executed"] > 0
    # config.c4: the batched figures of BASELINE config 4 ride in the same line (here: a small stage-structured stand-in)
    c4 = d["config"]["c4"]
    assert c4["instances_per_gpu"] == 4 and c4["instances_per_group"] == 2
    e = c4["smallT"]
    assert e["batched_newton_steps_per_s"] > 0 and e["single_system_steps_per_s"] > 0 and e["device_bytes_per_instance"] > 0
    assert abs(e["batched_problems_per_s_of_10_steps"] - e["batched_newton_steps_per_s"] / 10.0) < 1e-9
    cb4 = e["cpu_baseline"]                                           # the oracle on the box's host beside the GPU figure (reported, never compared)
    assert cb4["kind"] == "port" and cb4["cores"] == 1 and cb4["value"] > 0 and cb4["status"] == 0 and cb4["gpu_batched_over_cpu"] > 0
    assert d["config"]["rccl_ranks"] == 1                             # the size the product's RCCL communicator itself reports (ncclCommCount)
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 1 and c["value"] > 0 and c["unit"] == d["unit"] and isinstance(c["sample"], str)
    assert len(c["samples_s"]) == 2
    b0ii = c["B0_ii_reference_refactorisation"]                      # not derived: absent from the line unless measured (--cpu-baseline-full)
    assert b0ii["value"] is None and b0ii["measured"] is False and b0ii["factorizations_per_step"] >= 3 and b0ii["one_factorisation_s"] > 0
    assert c["B1_lapack_all_cores"]["value"] > 0 and c["B1_lapack_all_cores"]["cores"] >= 1


def test_bench_single_units_and_no_baseline():
    d = run_bench("--batch", "2", "--group", "1", "--lanes", "2", "--no-cpu-baseline", "--batched-passes", "2")
    assert d["cpu_baseline"] is None and d["config"]["batched"]["instances_per_group"] == 1 and d["value"] > 0 and d["config"]["c4"] is None


@pytest.mark.parametrize("torchrun", [True, False], ids=["under-torchrun", "self-spawned"])
def test_bench_two_ranks_on_one_gpu(torchrun):
    """the multi-rank path of bench.py (barrier, max-over-ranks time, gather of status rows, all-reduce of counters) on ONE GPU: two
    ranks with the gloo backend, both on device 0 — once launched by torch.distributed.run, once by `python bench.py --gpus 2` alone.
    value = sum over ranks of the steps over the max time (bench.py itself asserts that the gathered status table has 2 x B rows, all ok,
    and that the counters sum to 2 B P)."""
    two = run_bench_2ranks("--batch", "4", "--group", "2", "--lanes", "2", "--batched-passes", "3", torchrun=torchrun)
    assert two["n_gpus"] == 2 and two["cpu_baseline"] is None and two["steps"] == 3
    assert abs(two["value"] - 2 * 3 / (two["ms_per_step"] * 3e-3)) <= 1e-6 * two["value"]         # both ranks' steps over the max time
    b2 = two["config"]["batched"]
    assert b2["instances_per_gpu"] == 4 and abs(b2["newton_steps_per_s"] - 2 * 4 * 3 / (b2["ms_per_pass"] * 3e-3)) <= 1e-6 * b2["newton_steps_per_s"]
    assert "torch.distributed (gloo)" in b2["post_round_exchange"] and two["config"]["rccl_ranks"] is None      # (no RCCL communicator between two ranks on one device)
    print("two ranks on one GPU (%s): %.1f steps/s single, %.1f batched" % ("torchrun" if torchrun else "self-spawned", two["value"], b2["newton_steps_per_s"]))
