"""GPU: the schedule of the LDL^T of one instance does not change its bits.  One dense system alone takes the left-looking schedule of csrc/lfac.hip (the products of
the Schur complement as slices of the panel launches, deferred trailing updates; CALIPSO_HIP_LFAC=0: k_schur + the right-looking panel steps a group takes).  By default the finish of the factorisation (factor columns + merges of the
inverse blocks) of the completed solve blocks runs on a second stream while the pivot chain goes on, fed by the host from a progress word, and the
inertia counts are published right behind the chain (csrc/ldl.hip: launch_ldl); CALIPSO_HIP_LDL_OVERLAP=0 / CALIPSO_HIP_LDL_PUBLISH=0 /
CALIPSO_HIP_GRAPH_LDL=1 select the one-stream schedules.  These, CALIPSO_HIP_RHS_AHEAD, CALIPSO_HIP_SPEC_REFINE, CALIPSO_HIP_SPEC_STEP, CALIPSO_HIP_TAIL_PT (here), CALIPSO_HIP_LASTBLOCK_SYM (test_gpu_wform.py) and
CALIPSO_HIP_SOLVE_TAIL (below, to rounding) are the environment switches of one handle's step (INTEGRATION.md lists all thirteen of the library).  The switches are read once per process, so every variant runs in a process of its own;
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
used = []
facts = []
SHAPES = (((1500, 300, 60, 30, 3), 41), ((2100, 200, 40, 20, 3), 42))      # NP = 1536 (1024 + 512) and 2112 (two of 1024 + 64): ranges, tails, a narrow last block
if os.environ.get("CHILD_SHAPES"):
    SHAPES = tuple((tuple(int(v) for v in t.split(",")), 50 + k) for k, t in enumerate(os.environ["CHILD_SHAPES"].split(";")))
for shape, pid in SHAPES:
    s = build(pkg, pid, shape, indefinite=float(os.environ.get("CHILD_INDEFINITE", "0")))
    if os.environ.get("CHILD_SOLVE_BLOCK"):
        s.set_option("solve_block", int(os.environ["CHILD_SOLVE_BLOCK"]))
    for it in range(2):
        info = s.newton_step(advance=True)
        assert info["status"] >= 0, info
        facts.append(info["factorizations"])
        out.append(hashlib.sha256(np.ascontiguousarray(s.data("step").all).tobytes()).hexdigest())
        out.append(hashlib.sha256(np.ascontiguousarray(s.solution.all).tobytes()).hexdigest())
        out.append(repr(sorted((k, v) for k, v in info.items() if k in ("status", "refinement_rounds", "factorizations", "step_size"))))
    used.append(int(s.kernel_times()[6]))          # 1: the last factorisation took the left-looking schedule
print("FACT " + " ".join(str(s_) for s_ in facts))
print("LFAC " + "".join(str(u) for u in used))
print("DIGEST " + hashlib.sha256("\n".join(out).encode()).hexdigest())
'''


def run_variant(env):
    e = dict(os.environ)
    e.update(env)
    r = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT}], env=e, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("DIGEST ")]
    assert len(lines) == 1, r.stdout[-2000:]
    run_variant.lfac = [l for l in r.stdout.splitlines() if l.startswith("LFAC ")][0][5:]
    run_variant.facts = [int(v) for v in [l for l in r.stdout.splitlines() if l.startswith("FACT ")][0][5:].split()]
    return lines[0]


def test_newton_steps_do_not_depend_on_the_schedule_of_the_factorisation():
    ref = run_variant({})
    for env in ({"CALIPSO_HIP_LFAC": "0"},                # k_schur + the right-looking panel steps instead of the left-looking schedule of csrc/lfac.hip: the same operations per entry in the same order
                {"CALIPSO_HIP_LFAC": "0", "CALIPSO_HIP_LDL_OVERLAP": "0"},
                {"CALIPSO_HIP_LDL_OVERLAP": "0"}, {"CALIPSO_HIP_LDL_PUBLISH": "0"}, {"CALIPSO_HIP_GRAPH_LDL": "1"},
                {"CALIPSO_HIP_RHS_AHEAD": "0"},           # the operands of the first condensed solve on the main stream behind the factorisation instead of on the second stream
                {"CALIPSO_HIP_SPEC_REFINE": "0"},         # refinement rounds one by one, a host wait each, instead of queued ahead behind a device-side gate: same kernels, same order
                {"CALIPSO_HIP_TAIL_PT": "16"},            # k_solve_tail with 256 threads (two of the 32 column parts per thread: what group launches take) instead of 512: the same parts, columns and order of every sum
                {"CALIPSO_HIP_SPEC_STEP": "0"}):          # every decision of the step waited for in place instead of IC-1 queued ahead of the exit tests and the cone search / first candidate / merit behind the unread refinement report
        assert run_variant(env) == ref, env


def test_refactorisations_of_the_inertia_correction_under_every_schedule():
    """a non-convex Hessian: IC-1 fails its inertia test and inertia_correction! (inertia.jl:30-80) factors again with a larger primal regularisation — the scalars of the
    SECOND factorisation of a step differ from those of the first.  Every schedule must rebuild S with the current scalars: under CALIPSO_HIP_GRAPH_LDL=1 (panel steps
    replayed from a captured graph) the left-looking schedule, whose kernel carries the scalars by value, must not be captured (round-5 advisor finding)."""
    sh = {"CHILD_SHAPES": "1100,200,40,20,3", "CHILD_INDEFINITE": "6.0"}
    ref = run_variant(dict(sh, CALIPSO_HIP_LFAC="0", CALIPSO_HIP_LDL_OVERLAP="0"))
    assert max(run_variant.facts) >= 2, run_variant.facts            # the regularisation loop really ran
    for env in ({}, {"CALIPSO_HIP_GRAPH_LDL": "1"}, {"CALIPSO_HIP_LFAC": "0", "CALIPSO_HIP_GRAPH_LDL": "1"}, {"CALIPSO_HIP_LDL_PUBLISH": "0"}):
        assert run_variant(dict(sh, **env)) == ref, env
        if env.get("CALIPSO_HIP_GRAPH_LDL") == "1":
            assert set(run_variant.lfac) == {"0"}                       # no left-looking schedule inside a captured graph


@pytest.mark.parametrize("solve_block", [2048, 512])
def test_schedule_independence_holds_for_other_solve_block_widths(solve_block):
    """opt.solve_block = 2048: a solve block of two 1024-wide halves whose joining merge is split over two hand-overs (first phase with the left half,
    second with the right one); 512: more, narrower blocks.  NP = 2112 is 2048 + 64 / 4 x 512 + 64: the last block is a single panel."""
    sb = {"CHILD_SOLVE_BLOCK": str(solve_block)}
    ref = run_variant(dict(sb, CALIPSO_HIP_LDL_OVERLAP="0", CALIPSO_HIP_LFAC="0"))
    for env in ({}, {"CALIPSO_HIP_LFAC": "0"}, {"CALIPSO_HIP_LDL_OVERLAP": "0"}):
        assert run_variant(dict(sb, **env)) == ref, (solve_block, env)


@pytest.mark.parametrize("shapes", ["1000,40,0,0,3;1024,0,64,0,3", "1100,5,0,4,3;1300,700,100,50,4", "3000,1500,200,100,3", "4000,2000,600,200,3", "6000,3000,1000,300,3"])
def test_left_looking_schedule_on_edge_shapes(shapes):
    """the planner of csrc/lfac.hip on shapes its scan was not tuned on: two stages of constraints and no cone (m = 40), no equality (m = 64), a single partial stage (m = 17),
    cones of dimension 4, NP = 1024 (the smallest it takes) to 4096 (65 panels: the coarse scan) and 6016 (94 panels of 154 constraint stages: beyond the scan's grid, the plan
    comes from its fall-back over longer launches and heads) — against the right-looking schedule, bit for bit"""
    sh = {"CHILD_SHAPES": shapes}
    a = run_variant(dict(sh)); la = run_variant.lfac
    b = run_variant(dict(sh, CALIPSO_HIP_LFAC="0")); lb = run_variant.lfac
    assert set(la) == {"1"} and set(lb) == {"0"}, (la, lb)          # the default really took the left-looking schedule on every shape
    assert a == b, shapes


def test_separate_solve_tail_kernels_agree_with_the_fused_launch():
    """CALIPSO_HIP_SOLVE_TAIL=0: t2 = [gx; hx] dx, the back-substitution, the recovery and the local rows of the next refinement residual as separate launches
    instead of k_solve_tail: the same quantities in another summation order — the steps agree to rounding, the round counts are the same"""
    child = CHILD.replace('out.append(hashlib.sha256(np.ascontiguousarray(s.data("step").all).tobytes()).hexdigest())', 'np.save(os.environ["CHILD_OUT"] + "_" + str(pid) + "_" + str(it) + ".npy", s.data("step").all)')
    import tempfile
    import numpy as np
    with tempfile.TemporaryDirectory() as d:
        outs = {}
        for tag, env in (("fused", {}), ("separate", {"CALIPSO_HIP_SOLVE_TAIL": "0"})):
            e = dict(os.environ); e.update(env); e["CHILD_OUT"] = os.path.join(d, tag)
            r = subprocess.run([sys.executable, "-c", child % {"root": ROOT}], env=e, capture_output=True, text=True, timeout=600)
            assert r.returncode == 0, r.stderr[-2000:]
            outs[tag] = r.stdout
        for pid in (41, 42):
            for it in range(2):
                a = np.load(os.path.join(d, "fused_%d_%d.npy" % (pid, it))); b = np.load(os.path.join(d, "separate_%d_%d.npy" % (pid, it)))
                assert np.abs(a - b).max() <= 1e-9 * max(1.0, np.abs(a).max()), (pid, it)
