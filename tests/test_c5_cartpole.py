"""BASELINE config C5: cart-pole MPC sensitivities dw*/dtheta (examples/autotuning/cartpole.jl:85-146,179-227): nx = 49, ne = 40,
p = 102 right-hand sides through differentiate! (src/solver/differentiate.jl:1-61)."""
import functools

import numpy as np
import pytest

import problems as pr
from helpers import load_pkg
from test_oracle_solve import run as run_oracle

OPTS = dict(residual_tolerance=1e-3, optimality_tolerance=1e-3, equality_tolerance=1e-3, complementarity_tolerance=1e-3,
            slack_tolerance=1e-3, differentiate=1)
TIGHT = dict(residual_tolerance=1e-9, optimality_tolerance=1e-9, equality_tolerance=1e-9, complementarity_tolerance=1e-9,
             slack_tolerance=1e-9, differentiate=1)


@functools.lru_cache(maxsize=1)
def problem():
    return pr.cartpole_mpc()


def test_c5_shape_and_oracle_sensitivities_vs_finite_differences(oracle_mod):
    prob = problem()
    assert (prob.nx, prob.ne, prob.nc, prob.np) == (49, 40, 0, 102)
    o, st = run_oracle(oracle_mod, prob, **TIGHT)
    assert st == 1
    S = o.mat("solution_sensitivity", o.N, prob.np)
    x_star = o.point()["x"].copy()
    theta0 = prob.parameters.copy()
    # same check as the reference's double_integrator.jl:162-164 (1e-3), here by central differences on re-solves
    for j in (5, 9, 10, 12, 40, 95):                                           # weights, initial state, a later stage, terminal weight
        d = 1e-5
        xs = []
        for sgn in (+1, -1):
            prob.parameters = theta0.copy(); prob.parameters[j] += sgn * d
            oj, stj = run_oracle(oracle_mod, prob, **TIGHT)
            assert stj == 1
            xs.append(oj.point()["x"].copy())
        prob.parameters = theta0
        fd = (xs[0] - xs[1]) / (2 * d)
        assert np.abs(fd - S[:prob.nx, j]).max() <= 1e-3 * max(1.0, np.abs(fd).max()), j


@pytest.mark.gpu
def test_c5_hip_sensitivities_match_oracle(oracle_mod):
    prob = problem()
    pkg = load_pkg()
    s = pkg.Solver(prob, prob.nx, prob.np, prob.ne, prob.nc, parameters=prob.parameters, options=OPTS)
    pkg.initialize_b(s, prob.x0)
    assert pkg.solve_b(s)
    o, st = run_oracle(oracle_mod, prob, **OPTS)
    assert st == 1 and s.stats()["total_iterations"] == o.stats()["total_iterations"]
    assert np.abs(s.solution.all - o.point()["all"]).max() <= 1e-6 * max(1.0, np.abs(o.point()["all"]).max())
    S_gpu = s.data("solution_sensitivity")
    S_cpu = o.mat("solution_sensitivity", o.N, prob.np)
    assert np.abs(S_gpu - S_cpu).max() <= 1e-6 * max(1.0, np.abs(S_cpu).max())
    # the slice the auto-tuning loop consumes: d u_1 / d [cost weights; initial state]  (examples/autotuning/cartpole.jl:152-153,200)
    action1 = 4                                                                  # z = [x1(4); u1; ...]
    weights = [5, 6, 7, 8, 9]
    init = [10, 11, 12, 13]
    assert np.isfinite(S_gpu[action1, weights + init]).all() and np.abs(S_gpu[action1, init]).max() > 0
    # dR/dtheta assembled on the device equals the oracle's (residual_jacobian_parameters.jl:1-40)
    J = s.data("jacobian_parameters")
    assert np.abs(J - o.mat("jacobian_parameters", o.N, prob.np)).max() <= 1e-9


@pytest.mark.gpu
def test_c5_as_a_trajectory_problem_on_a_structured_handle(oracle_mod):
    """C5 is a trajectory problem (examples/autotuning/cartpole.jl:85-146: 10 stages of 4 states + 1 action, dynamics between neighbours): declared as such it
    runs on a structured handle — stage blocks, multifrontal LDL^T of S — and differentiate! (differentiate.jl:1-61) takes its 102 columns through the blocks and
    the fronts together.  Same iterations and sensitivities as the oracle (1e-6) and as the dense handle."""
    import time
    prob = problem()
    pkg = load_pkg()
    st = pr.structure_from_pattern(prob)
    assert len(st["hessian_block_start"]) >= 10                              # at least one Hessian block per stage (the cart position couples with nothing: blocks of its own)
    s = pkg.Solver(prob, prob.nx, prob.np, prob.ne, prob.nc, parameters=prob.parameters, options=OPTS, structure=st)
    dense = pkg.Solver(prob, prob.nx, prob.np, prob.ne, prob.nc, parameters=prob.parameters, options=OPTS)
    for h in (s, dense):
        pkg.initialize_b(h, prob.x0)
        assert pkg.solve_b(h)
    o, sto = run_oracle(oracle_mod, prob, **OPTS)
    assert sto == 1 and s.stats()["total_iterations"] == o.stats()["total_iterations"] == dense.stats()["total_iterations"]
    assert np.abs(s.solution.all - o.point()["all"]).max() <= 1e-6 * max(1.0, np.abs(o.point()["all"]).max())
    S_cpu = o.mat("solution_sensitivity", o.N, prob.np)
    S_str, S_dense = s.data("solution_sensitivity"), dense.data("solution_sensitivity")
    assert np.abs(S_str - S_cpu).max() <= 1e-6 * max(1.0, np.abs(S_cpu).max())
    assert np.abs(S_str - S_dense).max() <= 1e-8 * max(1.0, np.abs(S_dense).max())
    assert s.device_bytes() < dense.device_bytes()
    # timed side by side: differentiate! alone on the two handles (the batched LDS-resident path for thousands of such systems: tests/test_gpu_small.py)
    for name, h in (("structured", s), ("dense", dense)):
        h.differentiate(); h.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            h.differentiate()
        h.synchronize()
        print("differentiate! on the %s handle: %.3f ms" % (name, (time.perf_counter() - t0) / 20 * 1e3))
