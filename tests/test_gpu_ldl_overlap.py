"""GPU: the schedule of the LDL^T of one instance does not change its bits.  One dense system alone takes the left-looking schedule of csrc/lfac.hip (the products of
the Schur complement as slices of the panel launches, deferred trailing updates; CALIPSO_HIP_LFAC=0: k_schur + the right-looking panel steps a group takes).  By default the finish of the factorisation (factor columns + merges of the
inverse blocks) of the completed solve blocks runs on a second stream while the pivot chain goes on, fed by the host from a progress word, and the
inertia counts are published right behind the chain (csrc/ldl.hip: launch_ldl); CALIPSO_HIP_LDL_OVERLAP=0 / CALIPSO_HIP_LDL_PUBLISH=0 /
CALIPSO_HIP_GRAPH_LDL=1 select the one-stream schedules.  The switches are read once per process, so every variant runs in a process of its own;
the Newton steps they take must agree bit for bit (same kernels, same operands, only the order in time of independent launches differs)."""
import hashlib
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import hashlib, sys, os
import numpy as np
sys.path[:0] = [%(root)r, os.path.join(%(root)r, "tests")]
from helpers import load_pkg
from test_gpu_group import build
pkg = load_pkg()
out = []
for shape, pid in (((1500, 300, 60, 30, 3), 41), ((2100, 200, 40, 20, 3), 42)):      # NP = 1536 (1024 + 512) and 2112 (two of 1024 + 64): ranges, tails, a narrow last block
    s = build(pkg, pid, shape)
    if os.environ.get("CHILD_SOLVE_BLOCK"):
        s.set_option("solve_block", int(os.environ["CHILD_SOLVE_BLOCK"]))
    for it in range(2):
        info = s.newton_step(advance=True)
        assert info["status"] >= 0, info
        out.append(hashlib.sha256(np.ascontiguousarray(s.data("step").all).tobytes()).hexdigest())
        out.append(hashlib.sha256(np.ascontiguousarray(s.solution.all).tobytes()).hexdigest())
        out.append(repr(sorted((k, v) for k, v in info.items() if k in ("status", "refinement_rounds", "factorizations", "step_size"))))
print("DIGEST " + hashlib.sha256("\n".join(out).encode()).hexdigest())
'''


def run_variant(env):
    e = dict(os.environ)
    e.update(env)
    r = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT}], env=e, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("DIGEST ")]
    assert len(lines) == 1, r.stdout[-2000:]
    return lines[0]


def test_newton_steps_do_not_depend_on_the_schedule_of_the_factorisation():
    ref = run_variant({})
    for env in ({"CALIPSO_HIP_LFAC": "0"},                # k_schur + the right-looking panel steps instead of the left-looking schedule of csrc/lfac.hip: the same operations per entry in the same order
                {"CALIPSO_HIP_LFAC": "0", "CALIPSO_HIP_LDL_OVERLAP": "0"},
                {"CALIPSO_HIP_LDL_OVERLAP": "0"}, {"CALIPSO_HIP_LDL_PUBLISH": "0"}, {"CALIPSO_HIP_GRAPH_LDL": "1"}, {"CALIPSO_HIP_LDL_FEED": "64"},
                {"CALIPSO_HIP_WFORM_WGS": "64"},
                {"CALIPSO_HIP_RHS_AHEAD": "0"}):          # the operands of the first condensed solve on the main stream behind the factorisation instead of on the second stream beside k_schur
        assert run_variant(env) == ref, env


@pytest.mark.parametrize("solve_block", [2048, 512])
def test_schedule_independence_holds_for_other_solve_block_widths(solve_block):
    """opt.solve_block = 2048: a solve block of two 1024-wide halves whose joining merge is split over two hand-overs (first phase with the left half,
    second with the right one); 512: more, narrower blocks.  NP = 2112 is 2048 + 64 / 4 x 512 + 64: the last block is a single panel."""
    sb = {"CHILD_SOLVE_BLOCK": str(solve_block)}
    ref = run_variant(dict(sb, CALIPSO_HIP_LDL_OVERLAP="0"))
    for env in ({}, {"CALIPSO_HIP_LDL_FEED": "64"}, {"CALIPSO_HIP_LDL_FEED": "1024"}):
        assert run_variant(dict(sb, **env)) == ref, (solve_block, env)
