"""GPU: the synthetic dense conic QP of SURVEY.md 8(d) (the benchmark workload) — device-resident evaluator vs the host
callback path, one full Newton step vs the oracle at a size the oracle finishes in seconds, and size-independent
properties at larger sizes."""
import numpy as np
import pytest

import problems as pr
from helpers import load_pkg

pytestmark = pytest.mark.gpu


def build(pkg, uniform, pid, nx, ne, n_nn, n_soc, dim):
    prob, pt, lam = pr.synthetic_conic_qp(uniform, pid, nx, ne, n_nn, n_soc, dim)
    s = pkg.Solver(prob, prob.nx, 0, prob.ne, prob.nc, nonnegative_indices=prob.nonnegative_indices, second_order_indices=prob.second_order_indices)
    w = np.concatenate([pt["x"], pt["r"], pt["s"], pt["y"], pt["z"], pt["t"]])
    s.set("solution", w)
    s.set("dual", lam)
    for name, v in (("central_path", 0.17), ("penalty", 52.0), ("fraction_to_boundary", 0.99)):
        s.set(name, [v])
    s.qp_attach(prob.P, prob.q, prob.A, prob.b, prob.G, prob.h, 0.5)
    return prob, pt, lam, w, s


def test_splitmix_stream_matches_oracle(oracle_mod):
    pkg = load_pkg()
    for pid, sid in ((0, 1), (3, 7), (255, 19)):
        assert np.array_equal(pkg.splitmix_uniform(pid, sid, -1.0, 1.0, 1000), oracle_mod.splitmix_uniform(pid, sid, -1.0, 1.0, 1000))


def test_device_evaluator_matches_callback():
    pkg = load_pkg()
    prob, pt, lam, w, s = build(pkg, pkg.splitmix_uniform, 1, 200, 90, 30, 20, 3)
    fl = pkg.FLAGS
    flags = fl["objective"] | fl["objective_gradient_variables"] | fl["equality_constraint"] | fl["equality_dual_jacobian_variables"] | \
        fl["cone_constraint"] | fl["cone_dual_jacobian_variables"]
    s.qp_evaluate(flags, 0)
    dev = {k: s.get(k, n) for k, n in (("objective", 1), ("objective_gradient_variables", 200), ("equality_constraint", 90),
                                        ("equality_dual_jacobian_variables", 200), ("cone_constraint", 90), ("cone_dual_jacobian_variables", 200))}
    host = {}
    prob.evaluate(flags, pt["x"], pt["y"], pt["z"], np.zeros(0), lambda name: host.setdefault(name, np.zeros(dev[name].size if name in dev else 1)))
    for k in dev:
        assert np.abs(dev[k] - host[k]).max() <= 1e-12 * max(1.0, np.abs(host[k]).max()), k


def test_newton_step_vs_oracle(oracle_mod):
    """one inner iteration of solve! (solve.jl:98-353) on a 600/300/(100 + 50 x SOC3) synthetic problem"""
    pkg = load_pkg()
    prob, pt, lam, w, s = build(pkg, pkg.splitmix_uniform, 2, 600, 300, 100, 50, 3)
    fl = pkg.FLAGS
    s.qp_evaluate(fl["objective"] | fl["equality_constraint"] | fl["cone_constraint"], 0)
    s.cone(product=True, target=True)
    info = s.newton_step(advance=False)
    assert info["status"] == 0 and info["factorizations"] == 1 and info["refinement_rounds"] >= 1
    step = s.data("step").all
    # oracle: same state, one search direction + cone search + candidate merit
    o = oracle_mod.OracleSolver(prob.nx, 0, prob.ne, prob.nc, prob.nonnegative_indices, prob.second_order_indices)
    o.point()["all"][:] = w
    o.buf("dual")[:] = lam
    o.buf("central_path")[0] = 0.17; o.buf("penalty")[0] = 52.0
    o.set_int("linear_solve_refactor", 0)
    prob.evaluate(pr.ALL_VARIABLE_FLAGS, pt["x"], pt["y"], pt["z"], np.zeros(0), o.buf)
    o.cone(product=True, jacobian=True, target=True, barrier=True, barrier_gradient=True)
    o.residual()
    assert o.search_direction() == 0
    so = o.buf("step")
    assert np.abs(step - so).max() <= 1e-8 * max(1.0, np.abs(so).max())
    assert s.scalar("primal_regularization") == 1e-7 and s.scalar("dual_regularization") == 1e-7
    # fraction-to-boundary: identical halving counts
    for vec, dv, a_g in ((pt["s"], so[o.index("cone_slack") - 1], info["step_size"]), (pt["t"], so[o.index("cone_slack_dual") - 1], info["step_size_cone_slack_dual"])):
        a = 1.0
        while o.cone_violation(vec - a * dv, vec, 0.99):
            a *= 0.5
        assert a_g <= a   # the shared step size may be halved further by the filter line search
    # benchmark mode restored the iterate
    assert np.array_equal(s.get("solution", s.N), w)
    # a second identical step gives bit-identical results (deterministic reductions, no float atomics)
    info2 = s.newton_step(advance=False)
    assert np.array_equal(s.data("step").all, step) and info2["merit_candidate"] == info["merit_candidate"]


def test_solve_synthetic_to_tolerance():
    """size-independent property: solve! on a synthetic conic QP meets the reference's four convergence criteria and the cone
    membership of the slacks (cyberdrift.jl:306-307 style)"""
    pkg = load_pkg()
    prob, pt, lam, w, s = build(pkg, pkg.splitmix_uniform, 4, 300, 120, 40, 30, 3)
    s.set("solution", np.concatenate([pt["x"], np.zeros(s.N - 300)]))
    assert pkg.solve_b(s)
    res = s.data("residual")
    assert np.abs(res.all).sum() / s.N < 1e-4
    assert max(np.abs(res.equality_dual).max(), np.abs(res.cone_dual).max()) < 1e-4
    assert np.abs(s.get("cone_product", s.nc)).max() <= 1e-4
    sol = s.solution
    assert not s.cone_violation(sol.cone_slack, np.zeros(s.nc), 0.0) and not s.cone_violation(sol.cone_slack_dual, np.zeros(s.nc), 0.0)
    assert np.abs(prob.A @ sol.variables - prob.b).max() < 1e-4
