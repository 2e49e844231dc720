"""GPU: the full solve! loop (calipso_hip_solve over the HIP kernels) on the reference's own test problems — the same
known answers / convergence criteria the reference asserts (and that pin the oracle in test_oracle_solve.py), plus agreement
of the converged solution with the oracle's."""
import numpy as np
import pytest

import problems as pr
from helpers import load_pkg
from test_oracle_solve import run as run_oracle

pytestmark = pytest.mark.gpu


def run_hip(prob, **opts):
    pkg = load_pkg()
    s = pkg.Solver(prob, prob.nx, prob.np, prob.ne, prob.nc, parameters=prob.parameters, nonnegative_indices=prob.nonnegative_indices,
                   second_order_indices=prob.second_order_indices, options=opts)
    pkg.initialize_b(s, prob.x0)
    ok = pkg.solve_b(s)
    return s, ok


def criteria(s, tol=1e-4):
    """the four checks of every reference solver test, e.g. test/solver/test1.jl:21-30"""
    res = s.data("residual")
    assert np.abs(res.all).sum() / s.N < tol
    slack = max(np.abs(res.equality_dual).max() if s.ne else 0.0, np.abs(res.cone_dual).max() if s.nc else 0.0)
    assert slack < tol
    if s.ne:
        assert np.abs(s.get("equality_constraint", s.ne)).max() <= tol
    if s.nc:
        assert np.abs(s.get("cone_product", s.nc)).max() <= tol


def test_wachter_c1():
    """BASELINE config C1: README.md:97-121 / test/solver/wachter.jl:47: x* = [1, 0, 0.5] +- 1e-3"""
    s, ok = run_hip(pr.wachter())
    assert ok
    criteria(s)
    assert np.abs(s.solution.variables - np.array([1.0, 0.0, 0.5])).max() < 1e-3


def test_pendulum_c2(oracle_mod):
    """BASELINE config C2: pendulum swing-up T = 11 (test/examples/pendulum.jl:3-73), single instance on one MI355X"""
    prob = pr.pendulum(action_guess=np.zeros(10))
    s, ok = run_hip(prob)
    assert ok
    criteria(s)
    x = s.solution.variables
    assert np.abs(x[-2:] - np.array([np.pi, 0.0])).max() < 1e-3 and np.abs(x[:2]).max() < 1e-3
    o, st = run_oracle(oracle_mod, prob)
    assert st == 1
    assert np.abs(x - o.point()["x"]).max() < 1e-3    # same local solution as the oracle
    assert abs(s.stats()["total_iterations"] - o.stats()["total_iterations"]) <= 2


@pytest.mark.parametrize("name", ["maratos", "test1", "knitro"])
def test_reference_problems(oracle_mod, name):
    prob = getattr(pr, name)()
    s, ok = run_hip(prob)
    assert ok
    criteria(s)
    o, st = run_oracle(oracle_mod, prob)
    assert np.abs(s.solution.variables - o.point()["x"]).max() < 1e-3


@pytest.mark.parametrize("v,mu,gamma", [([0.0, 1.0, 1.0], 0.5, 1.0), ([0.0, 10.0, 1.0], 1.0, 1.0), ([0.0, 1.0, 0.0], 0.0, 1.0)])
def test_friction_cone(v, mu, gamma):
    """test/solver/friction_cone.jl:19-63 (one second-order cone, no nonnegative cones)"""
    prob = pr.friction_cone(v, mu, gamma, np.random.default_rng(11).standard_normal(3))
    s, ok = run_hip(prob)
    assert ok
    criteria(s)
    x = s.solution.variables
    assert not s.cone_violation(x, np.zeros(3), 0.0)
    if mu > 0:
        v_dir = np.array(v[1:]) / np.linalg.norm(v[1:])
        assert np.abs(v_dir + x[1:] / np.linalg.norm(x[1:])).max() < 1e-3 and np.linalg.norm(x[1:]) <= mu * gamma + 1e-12


def test_portfolio():
    """test/solver/portfolio.jl:6-62: 2 nonnegative + a second-order cone of dimension 12"""
    prob = pr.portfolio(seed=2)
    s, ok = run_hip(prob)
    assert ok
    criteria(s)
    sl = s.solution.cone_slack
    assert np.all(sl[:2] > -1e-5) and np.linalg.norm(sl[3:14]) < sl[2] + 1e-5
    assert np.abs(prob.b_cone - prob.A_cone @ s.solution.variables - sl).max() < 1e-4


def test_qp_equality_sensitivity_c5_style(oracle_mod):
    """test/solver/qp_equality.jl:37-122: differentiate! — sensitivities vs the analytic KKT inverse (1e-2) and vs the oracle"""
    prob = pr.qp_equality_parametric(seed=5)
    s, ok = run_hip(prob, residual_tolerance=1e-8, equality_tolerance=1e-6, complementarity_tolerance=1e-6, differentiate=1)
    assert ok
    criteria(s, tol=1e-6)
    nx, ne = prob.nx, prob.ne
    rz = np.block([[np.diag(prob.Pd), prob.A.T], [prob.A, np.zeros((ne, ne))]])
    fxp = s.problem["objective_jacobian_variables_parameters"].reshape(prob.np, nx).T
    gyxp = s.problem["equality_dual_jacobian_variables_parameters"].reshape(prob.np, nx).T
    gp = s.problem["equality_jacobian_parameters"].reshape(prob.np, ne).T
    sens = -np.linalg.solve(rz, np.vstack([fxp + gyxp, gp]))
    S = s.data("solution_sensitivity")
    assert np.abs(sens[:nx] - S[:nx]).max() < 1e-2
    o, st = run_oracle(oracle_mod, prob, residual_tolerance=1e-8, equality_tolerance=1e-6, complementarity_tolerance=1e-6, differentiate=1)
    assert np.abs(S - o.mat("solution_sensitivity", o.N, prob.np)).max() < 1e-5


def test_error_paths():
    pkg = load_pkg()
    prob = pr.wachter()
    with pytest.raises(pkg.CalipsoHipError):     # layout the reference itself is inconsistent for
        pkg.Solver(prob, 3, 0, 2, 2, nonnegative_indices=[2], second_order_indices=[[]])
    s = pkg.Solver(prob, 3, 0, 2, 2)
    with pytest.raises(pkg.CalipsoHipError):
        s.set("no_such_field", [1.0])
    with pytest.raises(pkg.CalipsoHipError):
        s.set("solution", np.zeros(5))


@pytest.mark.parametrize("which,expect", [("test2", [2.0 / 3.0, 1.0 / np.sqrt(3.0)]), ("test3", None), ("test4", [-1 / np.sqrt(6), 2 / np.sqrt(6), -1 / np.sqrt(6)])])
def test_small_nonconvex(oracle_mod, which, expect):
    """test/solver/test2.jl, test3.jl, test4.jl: the reference's convergence criteria, the known minimisers, and the oracle's answer"""
    prob = getattr(pr, which)(np.random.default_rng(4).random(3 if which == "test4" else 2))
    s, ok = run_hip(prob)
    assert ok
    criteria(s)
    if expect is not None:
        assert np.abs(s.solution.variables - np.array(expect)).max() < 2e-3
    o, st = run_oracle(oracle_mod, prob)
    assert st == 1 and np.abs(s.solution.variables - o.point()["x"]).max() < 1e-3
    assert s.stats()["total_iterations"] == o.stats()["total_iterations"]


def test_qp_nonnegative(oracle_mod):
    """test/solver/qp_nonnegative.jl: parametric QP with x >= 0, differentiate=true; criteria of :52-67 and parity of the solution and
    of the sensitivities with the oracle"""
    prob = pr.qp_nonnegative_parametric(seed=3)
    s, ok = run_hip(prob, differentiate=1)
    assert ok
    criteria(s)
    x = s.solution.variables
    assert np.all(x > -1e-4) and np.abs(prob.A @ x - prob.b).max() < 1e-4
    o, st = run_oracle(oracle_mod, prob, differentiate=1)
    assert st == 1 and s.stats()["total_iterations"] == o.stats()["total_iterations"]
    assert np.abs(s.solution.all - o.point()["all"]).max() <= 1e-6 * max(1.0, np.abs(o.point()["all"]).max())
    # dR/dtheta agrees; the sensitivities solve H S = -dR/dtheta at the converged point.  (S itself is NOT compared entry-wise with
    # the oracle's: with active bounds the slacks are ~1e-5, so iterates that agree to 1e-9 give K_zz entries that differ in the
    # third digit — the reference comments its own sensitivity asserts out for this problem, qp_nonnegative.jl:122-124.  Entry-wise
    # parity of differentiate! at a common point is test_gpu_multirhs.py.)
    J = s.data("jacobian_parameters")
    assert np.abs(J - o.mat("jacobian_parameters", o.N, prob.np)).max() <= 1e-6
    S_gpu = s.data("solution_sensitivity")
    assert np.isfinite(S_gpu).all()
    for j in (0, prob.nx, prob.np - 1):
        Hs = s.jacobian_variables_mul(S_gpu[:, j])
        assert np.abs(Hs + J[:, j]).max() <= 1e-6 * max(1.0, np.abs(J[:, j]).max(), np.abs(S_gpu[:, j]).max())


def test_double_integrator_sensitivities_on_the_device(oracle_mod):
    """test/examples/double_integrator.jl:3-164 through the HIP path: same iterations as the oracle, sensitivities of the variables within the
    reference's 1e-3 of the analytic -L_zz \\ L_z,theta and within 1e-6 of the oracle's"""
    from test_oracle_solve import analytic_sensitivity, run
    pkg = load_pkg()
    prob = pr.double_integrator(action_guess=[0.3, -0.2, 0.1, 0.05])
    opts = dict(residual_tolerance=1e-12, equality_tolerance=1e-8, complementarity_tolerance=1e-8, differentiate=1)
    s = pkg.Solver(prob, prob.nx, prob.np, prob.ne, prob.nc, parameters=prob.parameters, options=opts)
    pkg.initialize_b(s, prob.x0)
    assert pkg.solve_b(s)
    o, st = run(oracle_mod, prob, **opts)
    assert st == 1 and s.stats()["total_iterations"] == o.stats()["total_iterations"]
    S_gpu = s.data("solution_sensitivity")
    S_cpu = o.mat("solution_sensitivity", o.N, prob.np)
    assert np.abs(S_gpu - S_cpu).max() <= 1e-6 * max(1.0, np.abs(S_cpu).max())
    sol = s.solution
    sens = analytic_sensitivity(prob, sol.variables, sol.equality_dual)
    assert np.abs(sens[:prob.nx] - S_gpu[:prob.nx]).max() < 1e-3


def test_pendulum_c2_with_the_stage_parallel_factorisation(oracle_mod):
    """C2 again, host-callback evaluated, with the structure analysed after the first evaluate! and the Schur complement factored by the multifrontal
    path.  Whatever the later iterates do to the pattern (an upload outside the analysed structure puts the handle back on the dense treatment,
    structure.hip), the solve must end at the reference's solution with the reference's criteria."""
    pkg = load_pkg()
    prob = pr.pendulum(action_guess=np.zeros(10))
    s = pkg.Solver(prob, prob.nx, prob.np, prob.ne, prob.nc, parameters=prob.parameters, nonnegative_indices=prob.nonnegative_indices,
                   second_order_indices=prob.second_order_indices)
    pkg.initialize_b(s, prob.x0)
    s.evaluate(pkg.ALL_VARIABLE_FLAGS if hasattr(pkg, "ALL_VARIABLE_FLAGS") else pr.ALL_VARIABLE_FLAGS, 0)
    info = s.analyze_structure()
    sp = s.set_stage_parallel(True)
    assert sp["largest_front"] <= 196
    assert pkg.solve_b(s)
    criteria(s)
    x = s.solution.variables
    assert np.abs(x[-2:] - np.array([np.pi, 0.0])).max() < 1e-3 and np.abs(x[:2]).max() < 1e-3
    o, st = run_oracle(oracle_mod, prob)
    assert st == 1 and np.abs(x - o.point()["x"]).max() < 1e-3
    assert abs(s.stats()["total_iterations"] - o.stats()["total_iterations"]) <= 2


def test_pendulum_c2_with_stage_blocks(oracle_mod):
    """C2 once more, host-callback evaluated (every evaluate! uploads the dense blocks: the pack kernels of csrc/blocks.hip follow each upload), with the
    stage blocks on top of the stage-parallel factorisation: a REAL trajectory problem (3 x 3 stages of state + action, dynamics rows over two stages) through
    the block mat-vecs and the Schur complement by segment pairs.  An iterate whose pattern leaves the blocks puts the handle back on the dense treatment;
    either way the solve ends at the reference's solution with the reference's criteria."""
    pkg = load_pkg()
    prob = pr.pendulum(action_guess=np.zeros(10))
    s = pkg.Solver(prob, prob.nx, prob.np, prob.ne, prob.nc, parameters=prob.parameters, nonnegative_indices=prob.nonnegative_indices,
                   second_order_indices=prob.second_order_indices)
    pkg.initialize_b(s, prob.x0)
    s.evaluate(pr.ALL_VARIABLE_FLAGS, 0)
    s.analyze_structure()
    s.set_stage_parallel(True)
    info = s.set_stage_blocks(True)
    assert info["hessian_blocks"] >= 2 and info["z_blocks"] >= 2
    assert pkg.solve_b(s)
    criteria(s)
    x = s.solution.variables
    assert np.abs(x[-2:] - np.array([np.pi, 0.0])).max() < 1e-3 and np.abs(x[:2]).max() < 1e-3
    o, st = run_oracle(oracle_mod, prob)
    assert st == 1 and np.abs(x - o.point()["x"]).max() < 1e-3
    assert abs(s.stats()["total_iterations"] - o.stats()["total_iterations"]) <= 2
