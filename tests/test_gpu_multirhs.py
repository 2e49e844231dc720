"""differentiate! with many right-hand sides (src/solver/differentiate.jl:1-61): the device path pushes all np columns through the
factors at once (GEMM + block TRSM, csrc/gemm.hip); the oracle loops over columns like the reference."""
import numpy as np
import pytest

import problems as pr
from helpers import interior_point, make_pair

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shape", [(40, 12, 9, 0, 3), (70, 20, 6, 5, 3), (150, 60, 20, 8, 4), (600, 150, 33, 21, 3), (10, 0, 4, 0, 3), (12, 5, 0, 0, 3)])
def test_sensitivities_all_columns_match_oracle(oracle_mod, shape):
    nx, ne, n_nn, n_soc, dim = shape
    prob = pr.parametric_conic_qp(nx, ne, n_nn, n_soc, dim, seed=nx)
    pt, lam = interior_point(prob, 3)
    o, g = make_pair(oracle_mod, prob, pt, lam, ep=1e-5, ed=1e-5)
    o.cone(product=True, jacobian=True, target=True)      # differentiate! uses the cone Jacobians of the last cone! call (differentiate.jl:13)
    g.cone(product=True, jacobian=True, target=True)
    assert o.differentiate(prob) >= 0
    g.differentiate()
    S_cpu = o.mat("solution_sensitivity", o.N, prob.np)
    S_gpu = g.data("solution_sensitivity")
    J = g.data("jacobian_parameters")
    assert np.abs(J - o.mat("jacobian_parameters", o.N, prob.np)).max() <= 1e-12
    scale = max(1.0, np.abs(S_cpu).max())
    assert np.abs(S_gpu - S_cpu).max() <= 1e-8 * scale, (np.abs(S_gpu - S_cpu).max(), scale)
    # the column-at-a-time entry points give the same columns: H-condensed solve of column j through search_direction_symmetric
    if n_soc == 0:   # R+ only: the condensed solve is exact for the unreduced system, so  H S = -dR/dtheta  up to round-off
        for j in (0, prob.np // 2, prob.np - 1):
            Hs = g.jacobian_variables_mul(S_gpu[:, j])
            assert np.abs(Hs + J[:, j]).max() <= 1e-7 * max(1.0, np.abs(J[:, j]).max(), np.abs(S_gpu[:, j]).max())


@pytest.mark.parametrize("shape", [(70, 20, 6, 5, 3), (150, 60, 20, 8, 4)])
def test_sensitivities_with_the_stage_parallel_factorisation(oracle_mod, shape):
    """differentiate! when S is held by the multifrontal factorisation (calipso_hip_set_stage_parallel): the block triangular solves of gemm.hip have no
    dense factor to read, the right-hand sides go through the tree instead — same sensitivities"""
    nx, ne, n_nn, n_soc, dim = shape
    prob = pr.parametric_conic_qp(nx, ne, n_nn, n_soc, dim, seed=nx)
    pt, lam = interior_point(prob, 3)
    o, g = make_pair(oracle_mod, prob, pt, lam, ep=1e-5, ed=1e-5)
    o.cone(product=True, jacobian=True, target=True)
    g.cone(product=True, jacobian=True, target=True)
    g.analyze_structure()
    info = g.set_stage_parallel(True)           # a dense S of this size is one chain of <= 64-column fronts
    assert info["largest_front"] <= 196
    assert o.differentiate(prob) >= 0
    g.differentiate()
    S_cpu = o.mat("solution_sensitivity", o.N, prob.np)
    S_gpu = g.data("solution_sensitivity")
    scale = max(1.0, np.abs(S_cpu).max())
    assert np.abs(S_gpu - S_cpu).max() <= 1e-8 * scale, (np.abs(S_gpu - S_cpu).max(), scale)
