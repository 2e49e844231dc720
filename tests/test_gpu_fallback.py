"""GPU: search_direction_nonsymmetric! (src/solver/search_direction.jl:106-119), the `H \\ residual` fallback the reference takes
when iterative refinement fails (search_direction.jl:22).  Device: dense H from the block closed forms + pivoted LU (csrc/fallback.hip)."""
import numpy as np
import pytest

import problems as pr
from helpers import interior_point, make_pair, near_boundary_point

pytestmark = pytest.mark.gpu


def prepared_pair(oracle_mod, prob, pt, lam, **kw):
    o, g = make_pair(oracle_mod, prob, pt, lam, **kw)
    o.cone(product=True, jacobian=True, target=True, barrier=True, barrier_gradient=True)
    g.cone(product=True, jacobian=True, target=True, barrier=True, barrier_gradient=True)
    o.residual(); g.residual()
    return o, g


@pytest.mark.parametrize("shape", [(12, 5, 4, 0, 3), (30, 8, 4, 6, 3), (200, 60, 20, 10, 4)])
def test_nonsymmetric_solve_matches_lapack(oracle_mod, shape):
    nx, ne, n_nn, n_soc, dim = shape
    prob = pr.parametric_conic_qp(nx, ne, n_nn, n_soc, dim, seed=nx)
    pt, lam = interior_point(prob, 5)
    o, g = prepared_pair(oracle_mod, prob, pt, lam)
    o.residual_jacobian_variables()
    H = o.H_dense()
    R = np.array(o.buf("residual"))
    ref = np.linalg.solve(H, R)
    assert g.search_direction_nonsymmetric() == 0
    step = g.data("step").all
    assert np.abs(step - ref).max() <= 1e-9 * max(1.0, np.abs(ref).max())
    assert np.abs(H @ step - R).max() <= 1e-9 * max(1.0, np.abs(R).max())


@pytest.mark.parametrize("seed", [3, 5, 7, 9])
def test_refinement_failure_takes_the_fallback(oracle_mod, seed):
    prob = pr.parametric_conic_qp(30, 8, 4, 6, 3, seed=seed)
    pt, lam = near_boundary_point(prob, seed)
    o, g = prepared_pair(oracle_mod, prob, pt, lam, ep=0.0, ed=0.0)
    assert o.search_direction() == 2                       # the oracle's refinement fails -> its dense-LU stand-in
    rc = g.search_direction()
    assert rc == 2 and g.stats()["fallbacks"] == 1
    so, sg = np.array(o.buf("step")), g.data("step").all
    assert np.abs(so - sg).max() <= 1e-8 * max(1.0, np.abs(so).max())
    Hs = g.jacobian_variables_mul(sg)
    R = g.data("residual").all
    assert np.abs(Hs - R).max() <= 1e-8 * max(1.0, np.abs(R).max())
