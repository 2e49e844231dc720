"""Golden fixtures (tests/golden/*.npz, produced by tests/golden/make_golden.py from the pinned oracle): the oracle must
still reproduce them bit-for-bit on the CPU, and the HIP path must match them on the GPU to the stated tolerances."""
import os

import numpy as np
import pytest

import problems as pr
from helpers import load_pkg

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return dict(np.load(os.path.join(G, name + ".npz")))


def qp_from(g):
    soc = [list(g["soc"][g["soc_ptr"][k]:g["soc_ptr"][k + 1]]) for k in range(len(g["soc_ptr"]) - 1)]
    return pr.ConicQP(g["P"], g["q"], g["A"], g["b"], g["G"], g["h"], nonnegative_indices=list(g["nonneg"]), second_order_indices=soc,
                      objective_scale=1.0)


@pytest.mark.parametrize("name", ["kat_qp_10_5_5", "kat_soc_6_3_9"])
def test_oracle_reproduces_golden(oracle_mod, name):
    g = load(name)
    prob = qp_from(g)
    o = oracle_mod.OracleSolver(prob.nx, 0, prob.ne, prob.nc, prob.nonnegative_indices, prob.second_order_indices)
    o.point()["all"][:] = g["w"]
    o.buf("dual")[:] = g["lam"]
    for nm, v in (("central_path", 0.17), ("penalty", 52.0), ("primal_regularization", 0.12), ("dual_regularization", 0.21)):
        o.buf(nm)[0] = v
    p = o.point()
    prob.evaluate(pr.ALL_VARIABLE_FLAGS, p["x"], p["y"], p["z"], prob.parameters, o.buf)
    o.cone(barrier=True, barrier_gradient=True, product=True, jacobian=True, target=True)
    o.residual_jacobian_variables(); o.residual_jacobian_variables_symmetric(); o.residual(); o.residual_symmetric(0)
    assert np.array_equal(o.buf("residual"), g["residual"]) and np.array_equal(o.K_dense(), g["K"]) and np.array_equal(o.H_dense(), g["H"])
    assert np.array_equal(o.buf("residual_symmetric"), g["residual_symmetric"])
    o.factorize(update=False)
    assert np.array_equal(np.array(o.compute_inertia()), g["inertia"])
    o.search_direction_symmetric(0, fact=False)
    assert np.array_equal(o.buf("step"), g["step_first"])
    assert o.iterative_refinement()
    assert np.array_equal(o.buf("step"), g["step"])
    # the fixture itself satisfies the unreduced Newton system
    assert np.abs(g["H"] @ g["step"] - g["residual"]).max() <= 1e-10


@pytest.mark.parametrize("name,maker", [("c1_wachter_trace", lambda: pr.wachter()), ("c2_pendulum_trace", lambda: pr.pendulum(action_guess=np.zeros(10)))])
def test_oracle_reproduces_trace(oracle_mod, name, maker):
    g = load(name)
    prob = maker()
    o = oracle_mod.OracleSolver(prob.nx, prob.np, prob.ne, prob.nc, prob.nonnegative_indices, prob.second_order_indices)
    o.point()["x"][:] = prob.x0
    assert o.solve(prob) == int(g["status"][0]) == 1
    tr = o.trace()
    assert tr.shape == g["trace"].shape and np.allclose(tr, g["trace"], rtol=0, atol=1e-12)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["kat_qp_10_5_5", "kat_soc_6_3_9"])
def test_hip_matches_golden_step(name):
    g = load(name)
    prob = qp_from(g)
    pkg = load_pkg()
    s = pkg.Solver(prob, prob.nx, 0, prob.ne, prob.nc, nonnegative_indices=prob.nonnegative_indices, second_order_indices=prob.second_order_indices)
    s.set("solution", g["w"])
    if prob.ne:
        s.set("dual", g["lam"])
    for nm, v in (("central_path", 0.17), ("penalty", 52.0), ("primal_regularization", 0.12), ("dual_regularization", 0.21), ("fraction_to_boundary", 0.99)):
        s.set(nm, [v])
    s.evaluate(pr.ALL_VARIABLE_FLAGS, 0)
    s.cone(barrier=True, barrier_gradient=True, product=True, target=True)
    s.residual()
    rel = lambda a, b: np.abs(a - b).max() / max(1.0, np.abs(b).max())
    assert rel(s.data("residual").all, g["residual"]) <= 1e-12
    assert rel(s.jacobian_variables_symmetric(), g["K"]) <= 1e-12
    assert rel(s.get("cone_product", s.nc), g["cone_product"]) <= 1e-14 and np.array_equal(s.get("cone_target", s.nc), g["cone_target"])
    assert abs(s.scalar("barrier") - g["barrier"][0]) <= 1e-12
    s.residual_symmetric(0)
    assert rel(s.data("residual_symmetric"), g["residual_symmetric"]) <= 1e-12
    inertia, warn = s.factorize()
    assert np.array_equal(np.array(inertia), g["inertia"])
    s.search_direction_symmetric(0)
    assert rel(s.data("step").all, g["step_first"]) <= 1e-8
    ok, rounds, nrm = s.iterative_refinement()
    assert ok and rel(s.data("step").all, g["step"]) <= 1e-8
    a_s, a_t = s.cone_search()
    assert np.array_equal(np.array([a_s, a_t]), g["alpha"])           # identical halving counts
    assert abs(s.merit(0) - g["merit"][0]) <= 1e-12 * max(1.0, abs(g["merit"][0]))
    assert abs(s.constraint_violation(0) - g["theta"][0]) <= 1e-13
    s.merit_gradient()
    assert rel(s.data("merit_gradient"), g["merit_gradient"]) <= 1e-13
    assert abs(s.violations()["optimality_violation"] - g["optimality_error"][0]) <= 1e-12


@pytest.mark.gpu
@pytest.mark.parametrize("name,maker", [("c1_wachter_trace", lambda: pr.wachter()), ("c2_pendulum_trace", lambda: pr.pendulum(action_guess=np.zeros(10)))])
def test_hip_matches_golden_trace(name, maker):
    """BASELINE configs C1 / C2: every accepted iterate of solve! against the stored oracle trace (callback_inner hook)"""
    g = load(name)
    prob = maker()
    pkg = load_pkg()
    s = pkg.Solver(prob, prob.nx, prob.np, prob.ne, prob.nc)
    rows = []
    s.set_callbacks(inner=lambda sv: rows.append(sv.get("solution", sv.N)))
    pkg.initialize_b(s, prob.x0)
    assert pkg.solve_b(s)
    tr = np.array(rows)
    assert tr.shape == g["trace"].shape                                 # same number of Newton iterations
    scale = np.maximum(1.0, np.abs(g["trace"]).max(axis=1, keepdims=True))
    assert (np.abs(tr - g["trace"]) / scale).max() <= 1e-6               # iterates agree (rounding differs with elimination order)
    assert np.abs(s.get("solution", s.N) - g["solution"]).max() <= 1e-6 * max(1.0, np.abs(g["solution"]).max())


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["kat_sd_nonconvex_12_3_4", "kat_sd_portfolio_soc12"])
def test_hip_matches_golden_search_direction(name):
    """search_direction! as a whole from the default regularisation start (search_direction.jl:1-23, inertia.jl:30-80): the same walk through IC-1 .. IC-6 (the same
    final regularisation to the bit), the same inertia, the step to 1e-8 — on a non-convex Hessian and through a second-order cone of dimension 12"""
    from test_reference_fixtures import sd_problem
    g = load(name)
    prob, t = sd_problem(name)
    pkg = load_pkg()
    s = pkg.Solver(prob, prob.nx, 0, prob.ne, prob.nc, nonnegative_indices=prob.nonnegative_indices, second_order_indices=prob.second_order_indices)
    s.set("solution", g["w"])
    if prob.ne:
        s.set("dual", g["lam"])
    for nm in ("central_path", "penalty", "primal_regularization", "dual_regularization", "fraction_to_boundary"):
        s.set(nm, [t[nm][0, 0]])
    s.evaluate(pr.ALL_VARIABLE_FLAGS, 0)
    s.cone(product=True, target=True)
    s.residual()
    rc = s.search_direction()
    assert rc == g["status"][0]
    for nm in ("primal_regularization", "primal_regularization_last", "dual_regularization"):
        assert s.scalar(nm) == g[nm][0], nm
    st = s.data("step").all
    assert np.abs(st - g["step"]).max() <= 1e-8 * max(1.0, np.abs(g["step"]).max())
