"""Pins the oracle's full solve! loop (solve.jl:8-377) to the known answers of the reference's own tests."""
import numpy as np
import pytest

import problems as pr


def run(oracle_mod, prob, **opts):
    o = oracle_mod.OracleSolver(prob.nx, prob.np, prob.ne, prob.nc, prob.nonnegative_indices, prob.second_order_indices)
    for k, v in opts.items():
        if isinstance(v, float):
            o.set_opt(k, v)
        else:
            o.set_int(k, v)
    o.buf("parameters")[:] = prob.parameters
    o.point()["x"][:] = prob.x0            # initialize!(solver, x0)  initialize.jl:9-13
    status = o.solve(prob)
    return o, status


def criteria(o, tol=1e-4):
    """the four convergence checks every reference solver test makes, e.g. test1.jl:21-30"""
    res = o.buf("residual")
    assert np.abs(res).sum() / o.N < tol
    ry = res[o.index("equality_dual") - 1]
    rz = res[o.index("cone_dual") - 1]
    slack = max(np.abs(ry).max() if ry.size else 0.0, np.abs(rz).max() if rz.size else 0.0)
    assert slack < tol
    g = o.buf("equality_constraint")
    assert (np.abs(g).max() if g.size else 0.0) <= tol
    cp = o.buf("cone_product")
    assert (np.abs(cp).max() if cp.size else 0.0) <= tol


def test_wachter_known_answer(oracle_mod):
    """test/solver/wachter.jl:14-47 = README.md:97-121 (BASELINE config C1): x* = [1, 0, 0.5] +- 1e-3"""
    o, status = run(oracle_mod, pr.wachter())
    assert status == 1
    criteria(o)
    assert np.abs(o.point()["x"] - np.array([1.0, 0.0, 0.5])).max() < 1e-3
    st = o.stats()
    assert st["total_iterations"] < 60 and st["lu_fallbacks"] == 0


def test_maratos(oracle_mod):
    o, status = run(oracle_mod, pr.maratos())
    assert status == 1
    criteria(o)
    assert np.abs(o.point()["x"] - np.array([1.0, 0.0])).max() < 1e-3


def test_test1(oracle_mod):
    o, status = run(oracle_mod, pr.test1())
    assert status == 1
    criteria(o)
    x = o.point()["x"]
    assert np.allclose(x[:30] ** 2, 1.2, atol=1e-3) and np.abs(x[30:]).max() < 1e-3


@pytest.mark.parametrize("which,expect", [("test2", [2.0 / 3.0, 1.0 / np.sqrt(3.0)]), ("test3", None), ("test4", [-1 / np.sqrt(6), 2 / np.sqrt(6), -1 / np.sqrt(6)])])
def test_small_nonconvex(oracle_mod, which, expect):
    rng = np.random.default_rng(4)
    prob = getattr(pr, which)(rng.random(3 if which == "test4" else 2))
    o, status = run(oracle_mod, prob)
    assert status == 1
    criteria(o)
    if expect is not None:
        assert np.abs(o.point()["x"] - np.array(expect)).max() < 2e-3


def test_knitro_mpcc(oracle_mod):
    """test/solver/knitro.jl: converges; the commented asserts :39-44 give x = [1,0,2,0,0,0,3,6]"""
    o, status = run(oracle_mod, pr.knitro())
    assert status == 1
    criteria(o)
    assert np.abs(o.point()["x"] - np.array([1, 0, 2, 0, 0, 0, 3, 6.0])).max() < 1e-2


@pytest.mark.parametrize("v", [[0.0, 1.0, 0.0], [0.0, 1.0, 1.0], [0.0, 10.0, 1.0]])
@pytest.mark.parametrize("mu,gamma", [(0.5, 1.0), (1.0, 1.0), (0.0, 1.0), (0.5, 0.0)])
def test_friction_cone(oracle_mod, v, mu, gamma):
    """test/solver/friction_cone.jl:19-63"""
    rng = np.random.default_rng(11)
    prob = pr.friction_cone(v, mu, gamma, rng.standard_normal(3))
    o, status = run(oracle_mod, prob)
    assert status == 1
    criteria(o)
    x = o.point()["x"]
    assert not o.cone_violation(x, np.zeros(3), 0.0)                     # :56-57 (cone(x) = x)
    if np.linalg.norm(v[1:]) > 0 and gamma > 0 and mu > 0:
        v_dir = np.array(v[1:]) / np.linalg.norm(v[1:])
        b_dir = x[1:] / np.linalg.norm(x[1:])
        assert np.abs(v_dir + b_dir).max() < 1e-3                          # :58-62
        assert np.linalg.norm(x[1:]) <= mu * gamma + 1e-12


def test_portfolio(oracle_mod):
    """test/solver/portfolio.jl:6-62: 2 nonnegative + one SOC of dimension 12"""
    prob = pr.portfolio(seed=2)
    o, status = run(oracle_mod, prob)
    assert status == 1
    criteria(o)
    s = o.point()["s"]
    assert np.all(s[:2] > -1e-5)
    assert np.linalg.norm(s[3:14]) < s[2] + 1e-5
    assert np.abs(prob.b_cone - prob.A_cone @ o.point()["x"] - s).max() < 1e-4


def test_random_qp_solves(oracle_mod):
    """test/solver/qp_nonnegative.jl-style: convex QP with equality and nonnegative cone constraints"""
    prob = pr.random_qp(10, 5, 5, seed=21)
    o, status = run(oracle_mod, prob)
    assert status == 1
    criteria(o)


def test_qp_nonnegative(oracle_mod):
    """test/solver/qp_nonnegative.jl:52-67: convergence criteria, x >= -1e-4, A x = b (the reference's sensitivity asserts there are
    commented out, :122-124)"""
    prob = pr.qp_nonnegative_parametric(seed=3)
    o, status = run(oracle_mod, prob, differentiate=1)
    assert status == 1
    criteria(o)
    x = o.point()["x"]
    assert np.all(x > -1e-4) and np.abs(prob.A @ x - prob.b).max() < 1e-4
    S = o.mat("solution_sensitivity", o.N, prob.np)
    assert np.isfinite(S).all() and np.abs(S[:prob.nx]).max() > 0


def test_qp_equality_sensitivity(oracle_mod):
    """test/solver/qp_equality.jl:37-122: solution sensitivities vs the analytic KKT inverse and vs -H \\ dR/dtheta (1e-2)"""
    prob = pr.qp_equality_parametric(seed=5)
    o, status = run(oracle_mod, prob, residual_tolerance=1e-8, equality_tolerance=1e-6, complementarity_tolerance=1e-6, differentiate=1)
    assert status == 1
    criteria(o, tol=1e-6)
    nx, ne, npar = prob.nx, prob.ne, prob.np
    x, y = o.point()["x"], o.point()["y"]
    assert np.abs(prob.A @ x - prob.b).max() < 1e-6
    fxp = o.mat("objective_jacobian_variables_parameters", nx, npar)
    gyxp = o.mat("equality_dual_jacobian_variables_parameters", nx, npar)
    gp = o.mat("equality_jacobian_parameters", ne, npar)
    rz = np.block([[np.diag(prob.Pd), prob.A.T], [prob.A, np.zeros((ne, ne))]])
    rth = np.vstack([fxp + gyxp, gp])
    sens = -np.linalg.solve(rz, rth)
    H = o.H_dense()
    Jp = o.mat("jacobian_parameters", o.N, npar)
    sens_full = -np.linalg.solve(H, Jp)
    sens_solver = o.mat("solution_sensitivity", o.N, npar)
    assert np.abs(sens[:nx] - sens_full[:nx]).max() < 1e-2
    assert np.abs(sens[:nx] - sens_solver[:nx]).max() < 1e-2
    assert np.abs(sens_full[:nx] - sens_solver[:nx]).max() < 1e-2


def test_pendulum_swing_up(oracle_mod):
    """test/examples/pendulum.jl:3-73 = README.md:123-189 (BASELINE config C2): nx=32, ne=24, reaches x_T = (pi, 0)"""
    prob = pr.pendulum(action_guess=np.zeros(10))
    assert (prob.nx, prob.ne, prob.nc) == (32, 24, 0)
    o, status = run(oracle_mod, prob)
    assert status == 1
    criteria(o)
    x = o.point()["x"]
    assert np.abs(x[-2:] - np.array([np.pi, 0.0])).max() < 1e-3 and np.abs(x[:2]).max() < 1e-3


def analytic_sensitivity(prob, x, y):
    """-L_zz \\ L_z,theta of the Lagrangian in (x, y) at the solution (test/examples/double_integrator.jl:111-158), by finite-difference-free
    evaluation of the problem's own second derivatives: L_zz = [[f_xx + (g'y)_xx, g_x'], [g_x, 0]], L_z,theta = [[f_x,theta + (g'y)_x,theta], [g_theta]]"""
    nx, ne, npar = prob.nx, prob.ne, prob.np
    bufs = {}
    size = dict(objective=1, objective_gradient_variables=nx, equality_constraint=ne, cone_constraint=0, equality_dual_jacobian_variables=nx,
                cone_dual_jacobian_variables=nx, objective_jacobian_variables_variables=nx * nx, equality_dual_jacobian_variables_variables=nx * nx,
                cone_dual_jacobian_variables_variables=nx * nx, equality_jacobian_variables=ne * nx, cone_jacobian_variables=0,
                objective_jacobian_variables_parameters=nx * npar, equality_jacobian_parameters=ne * npar,
                equality_dual_jacobian_variables_parameters=nx * npar, cone_jacobian_parameters=0, cone_dual_jacobian_variables_parameters=nx * npar)
    prob.evaluate((1 << 16) - 1, x, y, np.zeros(0), prob.parameters, lambda nm: bufs.setdefault(nm, np.zeros(size[nm])))
    cm = lambda nm, r, c: bufs[nm].reshape(c, r).T
    Lxx = cm("objective_jacobian_variables_variables", nx, nx) + cm("equality_dual_jacobian_variables_variables", nx, nx)
    gx = cm("equality_jacobian_variables", ne, nx)
    Lzz = np.block([[Lxx, gx.T], [gx, np.zeros((ne, ne))]])
    Lzth = np.vstack([cm("objective_jacobian_variables_parameters", nx, npar) + cm("equality_dual_jacobian_variables_parameters", nx, npar),
                      cm("equality_jacobian_parameters", ne, npar)])
    return -np.linalg.solve(Lzz, Lzth)


def test_double_integrator_sensitivities(oracle_mod):
    """test/examples/double_integrator.jl:3-164: horizon 5, 42 parameters; solve to the example's tolerances with differentiate = true and compare
    the solver's sensitivities of the variables with the analytic -L_zz \\ L_z,theta (the reference asserts 1e-3, :162-164)"""
    prob = pr.double_integrator(action_guess=[0.3, -0.2, 0.1, 0.05])
    assert (prob.nx, prob.ne, prob.nc, prob.np) == (14, 12, 0, 42)
    o, status = run(oracle_mod, prob, residual_tolerance=1e-12, equality_tolerance=1e-8, complementarity_tolerance=1e-8, differentiate=1)
    assert status == 1
    res = o.buf("residual")
    assert np.abs(res[o.index("variables") - 1]).max() < 1e-4 and np.abs(res[o.index("equality_dual") - 1]).max() < 1e-4   # :97-108
    assert np.abs(o.buf("equality_constraint")).max() <= 1e-8
    x, y = o.point()["x"].copy(), o.point()["y"].copy()
    assert np.abs(x[:2]).max() < 1e-8 and np.abs(x[-2:] - np.array([1.0, 0.0])).max() < 1e-8       # initial / goal state constraints
    sens = analytic_sensitivity(prob, x, y)
    S = o.mat("solution_sensitivity", o.N, prob.np)
    assert np.abs(sens[:prob.nx] - S[:prob.nx]).max() < 1e-3                      # the reference's assertion (the regularised H of the last iterate
                                                                                   # differs from the exact KKT matrix by O(1e-4) here)
    # dR/dtheta rows as the reference checks them (:160-161, 1e-5)
    Jp = o.mat("jacobian_parameters", o.N, prob.np)
    bufs_gp = o.mat("equality_jacobian_parameters", prob.ne, prob.np)
    assert np.abs(Jp[o.index("equality_dual") - 1] - bufs_gp).max() < 1e-12
