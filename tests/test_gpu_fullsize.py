"""GPU, BASELINE.json's full sizes (C3: n = 5000 condensed; C4 shape: n = 5234): the oracle needs ~10 s per step there, so it is
run once (test_full_size_step_matches_the_oracle: C3 step / residual / inertia / refinement rounds against the CPU restatement)
and the other checks are size-independent properties of one Newton step computed entirely on the device:
  * the refined step solves the UNREDUCED Newton system: ||R - H*step||_inf <= 1e-10 (the reference's refinement criterion,
    iterative_refinement.jl:15), with H*v from the matrix-free multiply that test_gpu_parity validates against the oracle;
  * inertia of the regularised condensed matrix = (nx, ne+nc, 0)  (inertia.jl:7-11);
  * condensation identity: the condensed solution [dx,dy,dz] reproduces rows x, y, z of the unreduced system;
  * fraction-to-boundary: the returned step sizes keep s, t strictly inside the cones, and doubling them would not;
  * benchmark-mode steps are bit-reproducible and restore the iterate."""
import numpy as np
import pytest

import problems as pr
from helpers import load_pkg
from test_gpu_synthetic import build

pytestmark = pytest.mark.gpu

SHAPES = {"C3": (2500, 1500, 400, 200, 3), "C4": (2302, 2208, 244, 240, 2),
          "beyond": (4100, 700, 200, 100, 4)}     # nx padded to 4608 = 9 solve blocks: larger than anything in BASELINE.json


@pytest.mark.parametrize("cfg", ["C3", "C4", "beyond"])
def test_newton_step_properties_full_size(cfg):
    pkg = load_pkg()
    nx, ne, n_nn, n_soc, dim = SHAPES[cfg]
    prob, pt, lam, w, s = build(pkg, pkg.splitmix_uniform, 0, nx, ne, n_nn, n_soc, dim)
    nc = n_nn + n_soc * dim
    fl = pkg.FLAGS
    s.qp_evaluate(fl["objective"] | fl["equality_constraint"] | fl["cone_constraint"], 0)
    s.cone(product=True, target=True)
    info = s.newton_step(advance=False)
    assert info["status"] == 0 and info["factorizations"] == 1 and 1 <= info["refinement_rounds"] <= 10
    step = s.data("step").all
    R = s.data("residual").all
    assert np.array_equal(s.get("solution", s.N), w)                      # benchmark mode restored the iterate
    # unreduced Newton system
    Hs = s.jacobian_variables_mul(step)
    assert np.abs(R - Hs).max() <= 1e-10
    # inertia at the regularisation the step used (eps_p = eps_d = 1e-7: IC-1 accepted)
    assert s.scalar("primal_regularization") == 1e-7 and s.scalar("dual_regularization") == 1e-7
    inertia, warn = s.factorize()
    assert inertia == (nx, ne + nc, 0) and warn == 0
    # the step equals what a host-side dense solve of a random 1-D projection predicts: v'H step = v'R
    v = np.random.default_rng(1).standard_normal(s.N)
    lhs = v @ Hs
    assert abs(lhs - v @ R) <= 1e-8 * max(1.0, abs(v @ R))
    # fraction to boundary (tau = 0.99)
    S0, T0 = pt["s"], pt["t"]
    ds, dt = step[s.indices["cone_slack"] - 1], step[s.indices["cone_slack_dual"] - 1]
    a_s, a_t = s.cone_search()
    assert not s.cone_violation(S0 - a_s * ds, S0, 0.99) and not s.cone_violation(T0 - a_t * dt, T0, 0.99)
    if a_s < 1.0:
        assert s.cone_violation(S0 - 2 * a_s * ds, S0, 0.99)
    if a_t < 1.0:
        assert s.cone_violation(T0 - 2 * a_t * dt, T0, 0.99)
    assert info["step_size"] <= a_s and info["step_size_cone_slack_dual"] == a_t
    # determinism
    info2 = s.newton_step(advance=False)
    assert np.array_equal(s.data("step").all, step) and info2["merit_candidate"] == info["merit_candidate"]
    # the dense blocks round-trip through the stacked Jacobian storage bit-exactly
    A_back = s.get("equality_jacobian_variables", ne * nx).reshape(nx, ne).T
    assert np.array_equal(A_back, prob.A)


def test_advancing_steps_reduce_the_residual_full_size():
    """ten real (advancing) Newton iterations at fixed kappa, rho on C3: the KKT residual norm decreases monotonically enough
    to reach the inner-loop exit (optimality error <= 10 kappa) — the solver is doing Newton, not just linear algebra"""
    pkg = load_pkg()
    nx, ne, n_nn, n_soc, dim = SHAPES["C3"]
    prob, pt, lam, w, s = build(pkg, pkg.splitmix_uniform, 1, nx, ne, n_nn, n_soc, dim)
    fl = pkg.FLAGS
    s.qp_evaluate(fl["objective"] | fl["equality_constraint"] | fl["cone_constraint"], 0)
    s.cone(product=True, target=True)
    s.residual()
    v0 = s.violations()["optimality_violation"]
    for _ in range(12):
        info = s.newton_step(advance=True)
        assert info["status"] >= 0
    s.qp_evaluate(fl["objective_gradient_variables"] | fl["equality_dual_jacobian_variables"] | fl["cone_dual_jacobian_variables"], 0)
    s.cone(product=True, target=True)
    s.residual()
    v1 = s.violations()["optimality_violation"]
    assert v1 <= 10.0 * 0.17 and v1 < 0.05 * v0        # inner-loop exit of solve.jl:165 reached, residual cut > 20x
    sol = s.solution
    assert not s.cone_violation(sol.cone_slack, np.zeros(s.nc), 0.0) and not s.cone_violation(sol.cone_slack_dual, np.zeros(s.nc), 0.0)


def test_group_of_full_size_instances_is_bitwise_the_single_step():
    """two C3 instances stepped as one group (one launch sequence for both) equal their stand-alone steps bit for bit"""
    pkg = load_pkg()
    nx, ne, n_nn, n_soc, dim = SHAPES["C3"]
    fl = pkg.FLAGS

    def make(pid):
        prob, pt, lam, w, s = build(pkg, pkg.splitmix_uniform, pid, nx, ne, n_nn, n_soc, dim)
        s.qp_evaluate(fl["objective"] | fl["equality_constraint"] | fl["cone_constraint"], 0)
        s.cone(product=True, target=True)
        return s

    singles = [make(1), make(2)]
    members = [make(1), make(2)]
    g = pkg.Group(members)
    ref = [s.newton_step(advance=False) for s in singles]
    got = g.newton_step(advance=False)
    for r, q, s, m in zip(ref, got, singles, members):
        assert r == q and r["status"] == 0
        assert np.array_equal(s.data("step").all, m.data("step").all)
    g.close()


@pytest.mark.parametrize("cfg", ["C3", "C4"])
def test_full_size_step_matches_the_oracle(oracle_mod, cfg):
    """C3 (n = 5000 condensed, N = 8500) and one instance of C4's shape (n = 5234, N = 8890) at BASELINE's full sizes: the refined
    Newton step of the device path against the CPU restatement on the same inputs — the oracle needs ~10 s for one such step
    (sparse up-looking LDL^T in QDLDL's operation order), which is also what bench.py times as the CPU baseline.  Tolerances of
    SURVEY.md 8(c): step 1e-8, residual 1e-12."""
    pkg = load_pkg()
    nx, ne, n_nn, n_soc, dim = SHAPES[cfg]
    prob, pt, lam, w, s = build(pkg, pkg.splitmix_uniform, 0, nx, ne, n_nn, n_soc, dim)
    fl = pkg.FLAGS
    s.qp_evaluate(fl["objective"] | fl["equality_constraint"] | fl["cone_constraint"], 0)
    s.cone(product=True, target=True)
    info = s.newton_step(advance=False)
    step, R = s.data("step").all, s.data("residual").all
    o = oracle_mod.OracleSolver(prob.nx, 0, prob.ne, prob.nc, prob.nonnegative_indices, prob.second_order_indices)
    o.point()["all"][:] = w
    o.buf("dual")[:] = lam
    o.buf("central_path")[0] = 0.17; o.buf("penalty")[0] = 52.0
    o.set_int("linear_solve_refactor", 0)
    prob.evaluate(pr.ALL_VARIABLE_FLAGS, pt["x"], pt["y"], pt["z"], np.zeros(0), o.buf)
    o.cone(product=True, jacobian=True, target=True, barrier=True, barrier_gradient=True)
    o.residual()
    assert o.search_direction() == 0
    so, Ro = np.array(o.buf("step")), np.array(o.buf("residual"))
    assert np.abs(R - Ro).max() <= 1e-12 * max(1.0, np.abs(Ro).max())
    assert np.abs(step - so).max() <= 1e-8 * max(1.0, np.abs(so).max())
    assert o.stats()["last_refinement_rounds"] == info["refinement_rounds"]
    assert tuple(o.compute_inertia()) == (nx, ne + n_nn + n_soc * dim, 0)


def test_group_of_twelve_c4_instances_is_bitwise_the_single_steps():
    """BASELINE config C4's per-GPU unit: a group of 12 instances of C4's shape (nx = 2302, ne = 2208, nc = 244 R+ + 240 x SOC2) stepped
    through one launch sequence equals the 12 stand-alone steps bit for bit (each stand-alone step being oracle-checked at this shape
    by test_full_size_step_matches_the_oracle[C4])"""
    pkg = load_pkg()
    nx, ne, n_nn, n_soc, dim = SHAPES["C4"]
    fl = pkg.FLAGS
    members = []
    for pid in range(12):
        prob, pt, lam, w, s = build(pkg, pkg.splitmix_uniform, pid, nx, ne, n_nn, n_soc, dim)
        s.qp_evaluate(fl["objective"] | fl["equality_constraint"] | fl["cone_constraint"], 0)
        s.cone(product=True, target=True)
        s.problem = None; s.methods = None
        del prob
        members.append(s)
    ref, steps = [], []
    for s in members:                                  # stand-alone (benchmark mode restores the iterate)
        ref.append(s.newton_step(advance=False))
        steps.append(s.data("step").all)
    g = pkg.Group(members)
    got = g.newton_step(advance=False)
    for r, q, st, m in zip(ref, got, steps, members):
        assert r == q and r["status"] == 0 and r["factorizations"] == 1
        assert np.array_equal(st, m.data("step").all)
    g.close()
