"""GPU (-m gpu): the HIP path against the identities the REFERENCE'S OWN TESTS hold — directly, with numpy closed forms, WITHOUT the oracle.

The oracle (oracle/) is "parity unpinned": nothing it computes has met an output of the Julia code.  What the reference does pin in closed form
is the arithmetic of the assemble / residual / condensation / recovery kernels:
  * test/solver/problem.jl:112-142  every block of the unreduced KKT matrix H
  * test/solver/problem.jl:145-159  every block of the condensed matrix K (R+ cones): K_yy = -1/(rho + ep) - ed, K_zz = -(s - ed)/(t + (s - ed) ep) - ed
  * test/solver/problem.jl:162-189  the residual R and the condensed right-hand side b
  * test/solver/problem.jl:192-211  condensed step == unreduced solve; iterative refinement reaches 1e-10 from a noisy step
  * src/solver/cones/second_order.jl:19-22,50-65 + search_direction.jl:59-101  the arrow blocks and the closed-form arrow inverse of the recovery
These identities are exact; the reference asserts them at 1e-6 (its inputs are unseeded `randn`), here they are asserted at 1e-13 relative on the
device results, through the C ABI.  Same constants as problem.jl:56-61 (kappa = .17, rho = 52, ep = .12, ed = .21)."""
import numpy as np
import pytest

import problems as pr
from helpers import interior_point, load_pkg

pytestmark = pytest.mark.gpu
KAPPA, RHO, EP, ED = 0.17, 52.0, 0.12, 0.21


def rel(a, b):
    a, b = np.asarray(a, dtype=float), np.asarray(b, dtype=float)
    return np.abs(a - b).max() / max(1.0, np.abs(b).max()) if b.size else 0.0


def arrow(v):
    d = len(v)
    M = v[0] * np.eye(d)
    M[0, 1:] = v[1:]
    M[1:, 0] = v[1:]
    return M


def cone_blocks(prob, u):
    """d(s o t)/d(other) for u = the other operand: diag(u) on the nonnegative entries, arrow(u) per second-order cone
    (cones/nonnegative.jl:19-26, cones/second_order.jl:19-22)"""
    M = np.zeros((prob.nc, prob.nc))
    for i in prob.nonnegative_indices:
        M[i - 1, i - 1] = u[i - 1]
    for c in prob.second_order_indices:
        if c:
            i = np.array(c) - 1
            M[np.ix_(i, i)] = arrow(u[i])
    return M


def device_solver(prob, seed):
    pkg = load_pkg()
    pt, lam = interior_point(prob, seed=seed, tail=0.05 if max([len(c) for c in prob.second_order_indices] + [0]) > 16 else 0.3)
    g = pkg.Solver(prob, prob.nx, prob.np, prob.ne, prob.nc, parameters=prob.parameters, nonnegative_indices=prob.nonnegative_indices,
                   second_order_indices=prob.second_order_indices)
    w = np.concatenate([pt[k] for k in "xrsyzt"])
    g.set("solution", w)
    if prob.ne:
        g.set("dual", lam)
    for name, v in (("central_path", KAPPA), ("penalty", RHO), ("primal_regularization", EP), ("dual_regularization", ED), ("fraction_to_boundary", 0.99)):
        g.set(name, [v])
    g.evaluate(pr.ALL_VARIABLE_FLAGS, 0)
    g.cone(product=True, target=True)
    g.residual()
    return g, pt, lam


def closed_forms(prob, g, pt, lam):
    """H, R (unreduced) and K, b (condensed) written out from the problem data the user functions produced — numpy only"""
    nx, ne, nc = prob.nx, prob.ne, prob.nc
    P = g.problem
    M = lambda name, r, c: np.asarray(P[name]).reshape(c, r).T          # column-major host ProblemData
    Lxx = M("objective_jacobian_variables_variables", nx, nx) + M("equality_dual_jacobian_variables_variables", nx, nx) + M("cone_dual_jacobian_variables_variables", nx, nx)
    gx, hx = M("equality_jacobian_variables", ne, nx), M("cone_jacobian_variables", nc, nx)
    fx, gval, hval = P["objective_gradient_variables"], P["equality_constraint"], P["cone_constraint"]
    x, r, s, y, z, t = (pt[k] for k in "xrsyzt")
    T, Sbar = cone_blocks(prob, t), cone_blocks(prob, s) - ED * np.eye(nc)
    Ix, Ie, Ic = np.eye(nx), np.eye(ne), np.eye(nc)
    Z = np.zeros
    H = np.block([
        [Lxx + EP * Ix, Z((nx, ne)), Z((nx, nc)), gx.T, hx.T, Z((nx, nc))],
        [Z((ne, nx)), (RHO + EP) * Ie, Z((ne, nc)), -Ie, Z((ne, nc)), Z((ne, nc))],
        [Z((nc, nx)), Z((nc, ne)), EP * Ic, Z((nc, ne)), -Ic, -Ic],
        [gx, -Ie, Z((ne, nc)), -ED * Ie, Z((ne, nc)), Z((ne, nc))],
        [hx, Z((nc, ne)), -Ic, Z((nc, ne)), -ED * Ic, Z((nc, nc))],
        [Z((nc, nx)), Z((nc, ne)), T, Z((nc, ne)), Z((nc, nc)), Sbar]])
    target = np.zeros(nc)
    for i in prob.nonnegative_indices:
        target[i - 1] = 1.0
    prod = np.zeros(nc)
    for i in prob.nonnegative_indices:
        prod[i - 1] = s[i - 1] * t[i - 1]
    for c in prob.second_order_indices:
        if c:
            i = np.array(c) - 1
            target[i[0]] = 1.0
            prod[i[0]] = s[i] @ t[i]
            prod[i[1:]] = s[i[0]] * t[i[1:]] + t[i[0]] * s[i[1:]]
    R = np.concatenate([fx + gx.T @ y + hx.T @ z, lam + RHO * r - y, -z - t, gval - r, hval - s, prod - KAPPA * target])
    # condensation (residual_jacobian_variables.jl:110-167, residual.jl:53-101): eliminate r, s, t
    D = T + Sbar * EP                                                      # T + Sbar P with P = ep I
    Dinv = np.linalg.inv(D) if nc else np.zeros((0, 0))
    K = np.block([
        [Lxx + EP * Ix, gx.T, hx.T],
        [gx, (-1.0 / (RHO + EP) - ED) * Ie, Z((ne, nc))],
        [hx, Z((nc, ne)), -Dinv @ Sbar - ED * Ic]])
    Rr, Rs, Rt = R[nx:nx + ne], R[nx + ne:nx + ne + nc], R[nx + 2 * ne + 2 * nc:]
    b = np.concatenate([R[:nx], (gval - r) + Rr / (RHO + EP), (hval - s) + Dinv @ (Rt + Sbar @ Rs)])
    return dict(H=H, R=R, K=K, b=b, T=T, Sbar=Sbar)


CASES = {
    "problem_jl_10_5_5_nonnegative": lambda: pr.random_qp(10, 5, 5, seed=3),
    "soc3_soc4_6_3_9": lambda: pr.random_qp(6, 3, 9, seed=10, nonnegative_indices=[1, 2], second_order_indices=[[3, 4, 5], [6, 7, 8, 9]]),
    "soc12_portfolio_size_20_4_14": lambda: pr.random_qp(20, 4, 14, seed=5, nonnegative_indices=[1, 2], second_order_indices=[list(range(3, 15))]),
    "soc64_largest_dimension_70_5_66": lambda: pr.random_qp(70, 5, 66, seed=12, nonnegative_indices=[1, 2], second_order_indices=[list(range(3, 67))]),
    "mixed_120_50_64": lambda: pr.random_qp(120, 50, 64, seed=8, nonnegative_indices=list(range(1, 17)),
                                             second_order_indices=[list(range(17 + 3 * k, 20 + 3 * k)) for k in range(16)]),
}


@pytest.mark.parametrize("case", list(CASES))
def test_blocks_of_H_K_R_b_against_the_reference_closed_forms(case):
    prob = CASES[case]()
    g, pt, lam = device_solver(prob, seed=1)
    cf = closed_forms(prob, g, pt, lam)
    nx, ne, nc, N, n = prob.nx, prob.ne, prob.nc, g.N, g.n
    # problem.jl:162-178  residual
    assert rel(g.data("residual").all, cf["R"]) <= 1e-13
    # problem.jl:112-142  H (the device never stores it: every block is checked through H e_j for all unit vectors)
    Hdev = np.column_stack([g.jacobian_variables_mul(e) for e in np.eye(N)])
    assert rel(Hdev, cf["H"]) <= 1e-13
    assert np.linalg.matrix_rank(Hdev) == N
    # problem.jl:145-159  K, both triangles as the reference writes them
    Kdev = g.jacobian_variables_symmetric()
    assert rel(Kdev, cf["K"]) <= 1e-13
    if not any(prob.second_order_indices):
        s, t = pt["s"], pt["t"]
        sc = slice(nx + ne, n)
        assert rel(np.diag(Kdev[sc, sc]), -1.0 * (s - ED) / (t + (s - ED) * EP) - ED) <= 1e-14       # :158-159 verbatim
        assert rel(np.diag(Kdev[nx:nx + ne, nx:nx + ne]), np.full(ne, -1.0 / (RHO + EP) - ED)) <= 1e-15   # :152-153
    # problem.jl:184-189  condensed right-hand side
    g.residual_symmetric(0)
    assert rel(g.data("residual_symmetric"), cf["b"]) <= 1e-13


@pytest.mark.parametrize("case", list(CASES))
def test_condensed_step_equals_unreduced_solve_and_recovery_identities(case):
    """problem.jl:192-204 (condensed step == H \\ R) and the recovery formulas of search_direction.jl:59-101 incl. the closed-form arrow
    inverse (second_order.jl:50-65): given the device's own (dx, dy, dz), its (dr, ds, dt) satisfy the eliminated rows of H exactly"""
    prob = CASES[case]()
    g, pt, lam = device_solver(prob, seed=2)
    cf = closed_forms(prob, g, pt, lam)
    nx, ne, nc, N = prob.nx, prob.ne, prob.nc, g.N
    H, R = cf["H"], cf["R"]
    inertia, rc = g.factorize()
    assert inertia == (nx, ne + nc, 0)
    g.search_direction_symmetric(0)                                  # first solve: only triu(K) was factored (quirk B-3), so it is inexact with SOCs
    st = g.data("step")
    dr, ds, dy, dz, dt = st.equality_slack, st.cone_slack, st.equality_dual, st.cone_dual, st.cone_slack_dual
    Rr, Rs, Rt = R[nx:nx + ne], R[nx + ne:nx + ne + nc], R[nx + 2 * ne + 2 * nc:]
    # rows r, s, t of H d = R hold for the recovered blocks whatever (dx, dy, dz) are:
    assert rel((RHO + EP) * dr - dy, Rr) <= 1e-13                                    # search_direction.jl:69-71
    assert rel(EP * ds - dz - dt, Rs) <= 1e-12
    assert rel(cf["T"] @ ds + cf["Sbar"] @ dt, Rt) <= 1e-12                          # arrow(t) ds + (arrow(s) - ed I) dt = R_t  (closed-form arrow inverse)
    ok, rounds, norm = g.iterative_refinement()
    assert ok and norm <= 1e-10
    full = np.linalg.solve(H, R)
    assert rel(g.data("step").all, full) <= 1e-9                                     # problem.jl:204 asserts 1e-6
    assert np.abs(R - H @ g.data("step").all).max() <= 1e-10                          # iterative_refinement.jl: the tolerance it stops at
    # problem.jl:207-211: refinement recovers from a noisy step
    noisy = g.data("step").all + np.random.default_rng(7).standard_normal(N)
    g.set("step", noisy)
    ok, rounds, norm = g.iterative_refinement()
    assert ok and np.abs(R - H @ g.data("step").all).max() <= 1e-10


def test_second_order_product_barrier_and_target_closed_forms():
    """cones/second_order.jl:13-22,42 and nonnegative.jl:11-26 on the device: barrier 1/2 log(s1^2 - |s2:|^2), its gradient, s o t, e"""
    prob = CASES["soc3_soc4_6_3_9"]()
    g, pt, lam = device_solver(prob, seed=3)
    g.cone(barrier=True, barrier_gradient=True, product=True, jacobian=True, target=True)
    s, t = pt["s"], pt["t"]
    phi, grad, prod = 0.0, np.zeros(prob.nc), np.zeros(prob.nc)
    for i in prob.nonnegative_indices:
        phi += np.log(s[i - 1]); grad[i - 1] = 1.0 / s[i - 1]; prod[i - 1] = s[i - 1] * t[i - 1]
    for c in prob.second_order_indices:
        i = np.array(c) - 1
        det = s[i[0]] ** 2 - s[i[1:]] @ s[i[1:]]
        phi += 0.5 * np.log(det)
        grad[i] = np.concatenate([[s[i[0]]], -s[i[1:]]]) / det
        prod[i[0]] = s[i] @ t[i]; prod[i[1:]] = s[i[0]] * t[i[1:]] + t[i[0]] * s[i[1:]]
    assert abs(g.scalar("barrier") - phi) <= 1e-13 * max(1.0, abs(phi))
    assert rel(g.get("barrier_gradient", prob.nc), grad) <= 1e-14
    assert rel(g.get("cone_product", prob.nc), prod) <= 1e-15
    assert g.get("cone_target", prob.nc).tolist() == [1, 1, 1, 0, 0, 1, 0, 0, 0]
