"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C ABI, against the CPU oracle on the
same seeded inputs.  Tolerances are the ones SURVEY.md 8(c) states: residuals 1e-12 relative, steps 1e-8 relative,
integer results (inertia, halving counts, index sets) exact."""
import numpy as np
import pytest

import problems as pr
from helpers import interior_point, load_pkg, make_pair

pytestmark = pytest.mark.gpu


def rel(a, b):
    return np.abs(np.asarray(a) - np.asarray(b)).max() / max(1.0, np.abs(np.asarray(b)).max()) if np.asarray(b).size else 0.0


CASES = {
    "qp_nonneg_10_5_5": lambda: pr.random_qp(10, 5, 5, seed=3),
    "qp_soc_6_3_9": lambda: pr.random_qp(6, 3, 9, seed=10, nonnegative_indices=[1, 2], second_order_indices=[[3, 4, 5], [6, 7, 8, 9]]),
    "qp_soc12_20_4_14": lambda: pr.random_qp(20, 4, 14, seed=5, nonnegative_indices=[1, 2], second_order_indices=[list(range(3, 15))]),
    # wide cones (dimension > 4) run one wavefront per cone (csrc/soc_wide.hip): the largest supported dimension, and wide + small + nonnegative mixed
    "qp_soc64_70_5_66": lambda: pr.random_qp(70, 5, 66, seed=12, nonnegative_indices=[1, 2], second_order_indices=[list(range(3, 67))]),
    "qp_soc_mixed_widths_40_6_35": lambda: pr.random_qp(40, 6, 35, seed=13, nonnegative_indices=[1, 2, 3],
                                                          second_order_indices=[[4, 5, 6], list(range(7, 19)), [19, 20], list(range(21, 30)), [], list(range(30, 36))]),
    "qp_noeq_7_0_4": lambda: pr.random_qp(7, 0, 4, seed=6),
    "qp_nocone_9_4_0": lambda: pr.random_qp(9, 4, 0, seed=7),
    "qp_mixed_300_120_130": lambda: pr.random_qp(300, 120, 130, seed=8, nonnegative_indices=list(range(1, 41)),
                                                  second_order_indices=[list(range(41 + 3 * k, 44 + 3 * k)) for k in range(30)]),
}


@pytest.mark.parametrize("case", list(CASES))
def test_newton_step_parity(oracle_mod, case):
    prob = CASES[case]()
    pt, lam = interior_point(prob, seed=1, tail=0.05 if "soc64" in case else 0.3)
    o, g = make_pair(oracle_mod, prob, pt, lam)
    nx, ne, nc, N, n = o.nx, o.ne, o.nc, o.N, o.n
    # a1: Indices — bit-exact integer work
    for name in ("variables", "equality_slack", "cone_slack", "equality_dual", "cone_dual", "cone_slack_dual", "symmetric_equality",
                 "symmetric_cone", "primals", "duals", "violation_equality", "violation_cone", "cone_nonnegative", "cone_second_order"):
        assert np.array_equal(g.index(name), o.index(name)), name
    # a3: cone!
    o.cone(barrier=True, barrier_gradient=True, product=True, jacobian=True, target=True)
    g.cone(barrier=True, barrier_gradient=True, product=True, jacobian=True, target=True)
    if nc:
        assert rel(g.get("cone_product", nc), o.buf("cone_product")) <= 1e-14
        assert np.array_equal(g.get("cone_target", nc), o.buf("cone_target"))
        assert rel(g.get("barrier_gradient", nc), o.buf("barrier_gradient")[:nc]) <= 1e-14
        assert abs(g.scalar("barrier") - o.buf("barrier")[0]) <= 1e-12 * max(1.0, abs(o.buf("barrier")[0]))
    # a4: residual!
    o.residual(); g.residual()
    R = o.buf("residual").copy()
    assert rel(g.data("residual").all, R) <= 1e-12
    # a5/a6: condensed K (dense, both triangles) and the matrix-free H*v
    o.residual_jacobian_variables(); o.residual_jacobian_variables_symmetric()
    assert rel(g.jacobian_variables_symmetric(), o.K_dense()) <= 1e-12
    v = np.random.default_rng(9).standard_normal(N)
    assert rel(g.jacobian_variables_mul(v), o.H_mul(v)) <= 1e-12
    # a10: condensed right-hand side
    o.residual_symmetric(0); g.residual_symmetric(0)
    assert rel(g.data("residual_symmetric"), o.buf("residual_symmetric")) <= 1e-12
    # a7-a9: factorisation + inertia (identical triple)
    o.factorize(update=False)
    inertia, warn = g.factorize()
    assert inertia == o.compute_inertia() == (nx, ne + nc, 0) and warn == 0
    # a11/a12: condensed solve + recovery; compared with the oracle's (order-dependent rounding => 1e-8 relative)
    o.search_direction_symmetric(0, fact=False)
    g.search_direction_symmetric(0)
    first_o = o.buf("step").copy()
    first_g = g.data("step").all
    assert rel(first_g, first_o) <= 1e-8
    # a13: refinement against the unreduced system
    assert o.iterative_refinement()
    ok, rounds, nrm = g.iterative_refinement()
    assert ok and nrm <= 1e-10
    step_o = o.buf("step").copy(); step_g = g.data("step").all
    assert rel(step_g, step_o) <= 1e-8
    H = o.H_dense()
    assert np.abs(R - H @ step_g).max() <= 1e-9 * max(1.0, np.abs(R).max())
    # a14: cone fraction-to-boundary search — identical halving counts
    a_s, a_t = g.cone_search()
    w = o.point()["all"]; s, t = o.point()["s"], o.point()["t"]
    Ds, Dt = step_o[o.index("cone_slack") - 1], step_o[o.index("cone_slack_dual") - 1]
    for vec, dv, a_g in ((s, Ds, a_s), (t, Dt, a_t)):
        a = 1.0
        while nc and o.cone_violation(vec - a * dv, vec, 0.99):
            a *= 0.5
        assert a == a_g
    # a15: merit, merit gradient, constraint violation on the solution point
    o.cone(barrier=True, barrier_gradient=True); g.cone(barrier=True, barrier_gradient=True)
    M_o = o.merit(o.buf("objective")[0], o.point()["r"], o.buf("barrier")[0])
    assert abs(g.merit(0) - M_o) <= 1e-12 * max(1.0, abs(M_o))
    o.merit_gradient(); g.merit_gradient()
    assert rel(g.data("merit_gradient"), o.buf("merit_gradient")) <= 1e-13
    th_o = o.constraint_violation(o.buf("equality_constraint"), o.point()["r"], o.buf("cone_constraint"), o.point()["s"])
    assert abs(g.constraint_violation(0) - th_o) <= 1e-13 * max(1.0, th_o)
    # a16: violations / optimality error
    vio = g.violations()
    assert abs(vio["optimality_violation"] - o.optimality_error()) <= 1e-12 * max(1.0, o.optimality_error())
    assert abs(vio["residual_violation"] - np.abs(R).sum() / N) <= 1e-12 * max(1.0, np.abs(R).sum() / N)


def test_search_direction_with_inertia_correction(oracle_mod):
    """a7: non-convex Hessian => IC-1 fails, eps_p restarts at 1e-20 and grows x100 (quirk B-1); both sides must walk the same
    regularisation sequence and end with the same eps_p, inertia and (to 1e-8) step"""
    prob = pr.random_qp(12, 3, 4, seed=4)
    prob.P = -prob.P
    prob.Psym = prob.c * (prob.P + prob.P.T)
    pt, lam = interior_point(prob, seed=2)
    o, g = make_pair(oracle_mod, prob, pt, lam, kappa=1.0, rho=1.0, ep=0.0, ed=0.0)
    o.cone(product=True, jacobian=True, target=True); g.cone(product=True, target=True)
    o.residual(); g.residual()
    assert o.search_direction() in (0, 2)
    rc = g.search_direction()
    assert rc in (0, 2)
    assert g.scalar("primal_regularization") == o.buf("primal_regularization")[0] > 1e-7
    assert g.scalar("primal_regularization_last") == o.buf("primal_regularization_last")[0]
    assert g.scalar("dual_regularization") == o.buf("dual_regularization")[0]
    assert rel(g.data("step").all, o.buf("step")) <= 1e-8


def test_cone_violation_api(oracle_mod):
    prob = CASES["qp_soc_6_3_9"]()
    pt, lam = interior_point(prob, seed=3)
    o, g = make_pair(oracle_mod, prob, pt, lam)
    rng = np.random.default_rng(0)
    for k in range(20):
        xh = pt["s"] - rng.random() * 2.0 * rng.standard_normal(prob.nc)
        assert g.cone_violation(xh, pt["s"], 0.99) == o.cone_violation(xh, pt["s"], 0.99)
    assert not g.cone_violation(pt["s"], np.zeros(prob.nc), 0.0)
