"""GPU: the W-form of the triangular solves ("opt.solve_wform", csrc/ldl.hip: a solve = one launch per solve block and direction with the stacked
blocks [Tinv_b; W_b], W_b = L[below, b] Tinv_b) — linear_solve! (src/solver/linear_solver.jl:52-60, qdldl.jl:592-622) against the oracle, against the
four-launches-per-block form it replaces, for every block layout (two to five solve blocks, a narrower last block), alone and in a group."""
import numpy as np
import pytest

import problems as pr
from helpers import interior_point, load_pkg, make_pair
from test_gpu_shapes import soc_layout

pytestmark = pytest.mark.gpu

# nx, ne, n_nn, soc dims, solve_block  ->  NP, blocks
LAYOUTS = [
    (700, 120, 30, [3] * 10, 512),       # NP = 1024: 512 + 512
    (1100, 200, 40, [3] * 20, 1024),     # NP = 1536: 1024 + 512 (narrower last block)
    (1100, 200, 40, [3] * 20, 512),      # NP = 1536: three blocks of 512
    (1700, 300, 40, [2] * 10, 1024),     # NP = 2048: 1024 + 1024
    (2100, 100, 20, [4] * 5, 2048),      # NP = 2560: 2048 + 512
]


@pytest.mark.parametrize("layout", LAYOUTS, ids=lambda l: "nx%d_block%d" % (l[0], l[4]))
def test_wform_search_direction_against_oracle_and_plain_form(oracle_mod, layout):
    nx, ne, n_nn, dims, blockw = layout
    nonneg, soc, nc = soc_layout(n_nn, dims)
    prob = pr.random_qp(nx, ne, nc, seed=nx + ne + blockw, nonnegative_indices=nonneg, second_order_indices=soc)
    pt, lam = interior_point(prob, seed=3)
    o, g = make_pair(oracle_mod, prob, pt, lam, kappa=0.3, rho=7.0, ep=0.0, ed=0.0)
    o.set_int("linear_solve_refactor", 0)
    o.cone(product=True, jacobian=True, target=True)
    o.residual()
    assert o.search_direction() == 0
    so = o.buf("step")
    steps, rounds = {}, {}
    for wform in (1, 0):
        g.set_option("solve_block", blockw)
        g.set_option("solve_wform", wform)
        g.cone(product=True, target=True)
        g.residual()
        assert g.search_direction() == 0
        steps[wform] = g.data("step").all.copy()
        rounds[wform] = g.stats()["last_refinement_rounds"]
        assert np.abs(steps[wform] - so).max() <= 1e-8 * max(1.0, np.abs(so).max()), wform
    assert rounds[0] == rounds[1] == o.stats()["last_refinement_rounds"]
    assert np.abs(steps[1] - steps[0]).max() <= 1e-9 * max(1.0, np.abs(so).max())
    assert not np.array_equal(steps[1], steps[0]) or nx < 600       # (the two forms sum in different orders: identical bits would mean the option did nothing)
    with pytest.raises(load_pkg().CalipsoHipError, match="0 or 1"):
        g.set_option("solve_wform", 2)


def test_wform_linear_solve_residual_is_at_rounding_level():
    """K x = b with the W-form solve alone (no refinement): the residual against the dense condensed matrix is at the level of the plain form's"""
    pkg = load_pkg()
    nonneg, soc, nc = soc_layout(40, [3] * 20)
    prob = pr.random_qp(1100, 200, nc, seed=77, nonnegative_indices=nonneg, second_order_indices=soc)
    pt, lam = interior_point(prob, seed=9)
    s = pkg.Solver(prob, prob.nx, 0, prob.ne, prob.nc, nonnegative_indices=nonneg, second_order_indices=soc)
    s.set("solution", np.concatenate([pt[k] for k in "xrsyzt"]))
    s.set("dual", lam)
    for name, v in (("central_path", 0.3), ("penalty", 7.0), ("primal_regularization", 1e-7), ("dual_regularization", 1e-7)):
        s.set(name, [v])
    s.evaluate(pr.ALL_VARIABLE_FLAGS, 0)
    s.cone(product=True, target=True)
    s.residual()
    K = s.jacobian_variables_symmetric()
    K = np.triu(K) + np.triu(K, 1).T
    b = np.random.default_rng(1).standard_normal(prob.nx + prob.ne + prob.nc)
    res = {}
    for wform in (1, 0):
        s.set_option("solve_wform", wform)
        inertia, _ = s.factorize()
        assert inertia == (prob.nx, prob.ne + prob.nc, 0)
        s.set("residual_symmetric", b)
        s.linear_solve()
        x = s.data("step_symmetric")
        res[wform] = np.abs(K @ x - b).max() / max(1.0, np.abs(x).max())
    assert res[1] <= 1e-9 and res[0] <= 1e-9, res
    assert res[1] <= 20 * res[0] + 1e-13, res


def test_wform_group_members_get_the_bits_of_a_handle_stepped_alone():
    pkg = load_pkg()
    from test_gpu_group import build, same
    shape = (1700, 300, 40, 20, 3)
    for blockw, wform in ((1024, 1), (512, 1)):
        singles = [build(pkg, p, shape) for p in (41, 42)]
        members = [build(pkg, p, shape) for p in (41, 42)]
        for h in singles + members:
            h.set_option("solve_block", blockw)
            h.set_option("solve_wform", wform)
        g = pkg.Group(members)
        ref = [s.newton_step(advance=False) for s in singles]
        got = g.newton_step(advance=False)
        for r, q, s, m in zip(ref, got, singles, members):
            assert r == q and r["status"] == 0
            assert same(s.data("step").all, m.data("step").all)
        g.close()
    members = [build(pkg, p, shape) for p in (41, 42)]
    members[1].set_option("solve_wform", 0)
    g = pkg.Group(members)
    with pytest.raises(pkg.CalipsoHipError, match="agree on opt.solve_wform"):
        g.newton_step(advance=False)
    g.close()


def test_last_block_through_its_symmetric_inverse_agrees_with_the_two_triangular_launches(tmp_path):
    """the last solve block has nothing below it: by default its forward and backward steps are ONE mat-vec with Msym = Tinv' D^-1 Tinv (ldl.hip: k_lastblock_sym,
    k_block_sym); CALIPSO_HIP_LASTBLOCK_SYM=0 (read once per process) keeps the two triangular launches.  Same Newton step to rounding, same refinement rounds."""
    import os
    import subprocess
    import sys
    from helpers import ROOT
    child = r'''
import sys, os
import numpy as np
sys.path[:0] = [%(root)r, os.path.join(%(root)r, "tests")]
from helpers import load_pkg
from test_gpu_group import build
pkg = load_pkg()
out = []
for shape, pid in (((1500, 300, 60, 30, 3), 41), ((2100, 200, 40, 20, 3), 42)):      # NP = 1536 (1024 + 512) and 2112 (1024 + 1024 + 64)
    s = build(pkg, pid, shape)
    info = s.newton_step(advance=False)
    assert info["status"] == 0, info
    out.append(np.array(s.data("step").all))
    out.append(np.array([info["refinement_rounds"]], dtype=np.float64))
np.save(sys.argv[1], np.concatenate(out))
'''
    res = {}
    for v in ("1", "0"):
        f = str(tmp_path / ("sym%s.npy" % v))
        e = dict(os.environ, CALIPSO_HIP_LASTBLOCK_SYM=v)
        r = subprocess.run([sys.executable, "-c", child % {"root": ROOT}, f], env=e, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        res[v] = np.load(f)
    assert res["1"].shape == res["0"].shape
    assert np.abs(res["1"] - res["0"]).max() <= 1e-9 * max(1.0, np.abs(res["0"]).max())
    assert not np.array_equal(res["1"], res["0"])          # (identical bits would mean the switch did nothing)
