"""GPU: shape sweep of search_direction! (inertia correction + condensed solve + refinement) against the oracle — padding
boundaries of the 512-wide solve blocks and 64/128-wide tiles, more constraints than variables, no equalities, no cones,
large and many second-order cones, ragged cone dimensions."""
import numpy as np
import pytest

import problems as pr
from helpers import interior_point, load_pkg, make_pair

pytestmark = pytest.mark.gpu


def soc_layout(n_nn, dims):
    idx, start = [], n_nn + 1
    for d in dims:
        idx.append(list(range(start, start + d)))
        start += d
    return list(range(1, n_nn + 1)), (idx if idx else [[]]), n_nn + sum(dims)


SHAPES = [
    # nx, ne, n_nn, soc dims
    (1, 1, 1, []),
    (63, 10, 5, [2, 3]),
    (64, 0, 7, [4]),
    (65, 30, 0, [3] * 10),
    (127, 200, 50, [2] * 20),            # m > nx
    (511, 100, 20, [5, 7, 9]),
    (512, 64, 64, [16]),
    (513, 77, 13, [33, 2, 64]),          # crosses the 512 padding boundary; the largest cone of ONE element per lane (64)
    (700, 300, 0, []),                   # equality-only
    (300, 0, 0, []),                     # unconstrained Newton system
    (200, 50, 400, []),                  # many nonnegative cones
]


@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: "nx%d_ne%d_nn%d_soc%s" % (s[0], s[1], s[2], "x".join(map(str, s[3])) or "0"))
def test_search_direction_shapes(oracle_mod, shape):
    nx, ne, n_nn, dims = shape
    nonneg, soc, nc = soc_layout(n_nn, dims)
    prob = pr.random_qp(nx, ne, nc, seed=nx + 7 * ne + nc, nonnegative_indices=nonneg, second_order_indices=soc)
    pt, lam = interior_point(prob, seed=5)
    o, g = make_pair(oracle_mod, prob, pt, lam, kappa=0.3, rho=7.0, ep=0.0, ed=0.0)
    o.set_int("linear_solve_refactor", 0)
    o.cone(product=True, jacobian=True, target=True)
    g.cone(product=True, target=True)
    o.residual(); g.residual()
    rc_o = o.search_direction()
    rc_g = g.search_direction()
    assert rc_o == 0 and rc_g == 0
    so, sg = o.buf("step"), g.data("step").all
    assert np.abs(sg - so).max() <= 1e-8 * max(1.0, np.abs(so).max())
    assert g.scalar("primal_regularization") == o.buf("primal_regularization")[0]
    inertia, _ = g.factorize()
    assert inertia == (nx, ne + nc, 0)
    if nc:
        a_s, a_t = g.cone_search()
        for vec, dv, a_g in ((pt["s"], so[o.index("cone_slack") - 1], a_s), (pt["t"], so[o.index("cone_slack_dual") - 1], a_t)):
            a = 1.0
            while o.cone_violation(vec - a * dv, vec, 0.99):
                a *= 0.5
            assert a == a_g


@pytest.mark.parametrize("dims", [[100], [130, 3, 70], [200, 12], [600, 5], [1024]], ids=lambda d: "soc" + "x".join(map(str, d)))
def test_second_order_cones_wider_than_a_wavefront(oracle_mod, dims):
    """cones of dimension > 64 (the reference has no limit: cones/second_order.jl:1-69): two / four elements per lane of the cone's wavefront
    (csrc/soc_wide.hip); dimension 130 and 200 keep the d x d block of the cone outside the LDS; 600 and 1024 (the maximum) take sixteen elements per lane.
    Whole search_direction! against the oracle."""
    nx, ne, n_nn = (260 if sum(dims) < 300 else 1100), 40, 6
    nonneg, soc, nc = soc_layout(n_nn, dims)
    prob = pr.random_qp(nx, ne, nc, seed=sum(dims), nonnegative_indices=nonneg, second_order_indices=soc)
    pt, lam = interior_point(prob, seed=2, tail=0.04)
    o, g = make_pair(oracle_mod, prob, pt, lam, kappa=0.3, rho=7.0, ep=0.0, ed=0.0)
    o.set_int("linear_solve_refactor", 0)
    o.cone(product=True, jacobian=True, target=True, barrier=True, barrier_gradient=True)
    g.cone(product=True, target=True, barrier=True, barrier_gradient=True)
    assert np.abs(g.get("cone_product", nc) - o.buf("cone_product")).max() <= 1e-13 * max(1.0, np.abs(o.buf("cone_product")).max())
    o.residual(); g.residual()
    assert o.search_direction() == 0 and g.search_direction() == 0
    so, sg = o.buf("step"), g.data("step").all
    assert np.abs(sg - so).max() <= 1e-8 * max(1.0, np.abs(so).max())
    assert g.stats()["last_refinement_rounds"] == o.stats()["last_refinement_rounds"]
    inertia, _ = g.factorize()
    assert inertia == (nx, ne + nc, 0)
    a_s, a_t = g.cone_search()
    for vec, dv, a_g in ((pt["s"], so[o.index("cone_slack") - 1], a_s), (pt["t"], so[o.index("cone_slack_dual") - 1], a_t)):
        a = 1.0
        while o.cone_violation(vec - a * dv, vec, 0.99):
            a *= 0.5
        assert a == a_g
    with pytest.raises(load_pkg().CalipsoHipError):
        n2, s2, c2 = soc_layout(0, [1025])
        p2 = pr.random_qp(20, 0, c2, seed=1, nonnegative_indices=n2, second_order_indices=s2)
        load_pkg().Solver(p2, p2.nx, 0, p2.ne, p2.nc, nonnegative_indices=n2, second_order_indices=s2)
