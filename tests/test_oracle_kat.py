"""Pins the oracle's assemble / condense / residual / solve arithmetic to the closed-form identities of the
reference's own unit test, test/solver/problem.jl:112-211 (the only reference test that pins this arithmetic)."""
import numpy as np
import pytest

import problems as pr


def setup_solver(oracle_mod, prob, seed=1, kappa=0.17, rho=52.0, ep=0.12, ed=0.21):
    o = oracle_mod.OracleSolver(prob.nx, prob.np, prob.ne, prob.nc, prob.nonnegative_indices, prob.second_order_indices)
    rng = np.random.default_rng(seed)
    pt = o.point()
    pt["x"][:] = rng.standard_normal(prob.nx)
    pt["r"][:] = rng.random(prob.ne)
    pt["s"][:] = rng.random(prob.nc)
    pt["y"][:] = rng.standard_normal(prob.ne)
    pt["z"][:] = rng.standard_normal(prob.nc)
    pt["t"][:] = rng.random(prob.nc)
    # second-order blocks: make s, t interior
    for c in prob.second_order_indices:
        if c:
            i = np.array(c) - 1
            pt["s"][i[0]] = 1.0 + np.linalg.norm(pt["s"][i[1:]])
            pt["t"][i[0]] = 1.0 + np.linalg.norm(pt["t"][i[1:]])
    o.buf("central_path")[0] = kappa
    o.buf("penalty")[0] = rho
    o.buf("dual")[:] = rng.standard_normal(prob.ne)
    o.buf("primal_regularization")[0] = ep
    o.buf("dual_regularization")[0] = ed
    prob.evaluate(pr.ALL_VARIABLE_FLAGS, pt["x"], pt["y"], pt["z"], prob.parameters, o.buf)
    o.cone(product=True, jacobian=True, target=True)
    o.residual_jacobian_variables()
    o.residual_jacobian_variables_symmetric()
    o.residual()
    o.residual_symmetric(0)
    return o


def test_problem_jl_identities(oracle_mod):
    """test/solver/problem.jl:25-189 with nonnegative cones (10 vars, 5 eq, 5 cone; kappa=.17 rho=52 ep=.12 ed=.21)"""
    prob = pr.random_qp(10, 5, 5, seed=3)
    o = setup_solver(oracle_mod, prob)
    nx, ne, nc, N, n = o.nx, o.ne, o.nc, o.N, o.n
    ep, ed, rho, kappa = 0.12, 0.21, 52.0, 0.17
    H = o.H_dense()
    ix = {k: o.index(k) - 1 for k in ("variables", "equality_slack", "cone_slack", "equality_dual", "cone_dual",
                                      "cone_slack_dual", "symmetric_equality", "symmetric_cone")}
    pt = o.point()
    s, t = pt["s"], pt["t"]
    fxx = o.mat("objective_jacobian_variables_variables", nx, nx)
    gyxx = o.mat("equality_dual_jacobian_variables_variables", nx, nx)
    hzxx = o.mat("cone_dual_jacobian_variables_variables", nx, nx)
    gx = o.mat("equality_jacobian_variables", ne, nx)
    hx = o.mat("cone_jacobian_variables", nc, nx)
    I = np.eye
    blk = lambda a, b: H[np.ix_(ix[a], ix[b])]
    assert np.linalg.matrix_rank(H) == N                                              # :112
    assert np.linalg.norm(blk("variables", "variables") - (fxx + gyxx + hzxx + ep * I(nx))) < 1e-6   # :113-114
    assert np.linalg.norm(blk("equality_dual", "variables") - gx) < 1e-6               # :115-118
    assert np.linalg.norm(blk("variables", "equality_dual") - gx.T) < 1e-6
    assert np.linalg.norm(blk("equality_dual", "equality_dual") + ed * I(ne)) < 1e-6   # :119-120
    assert np.linalg.norm(blk("cone_dual", "variables") - hx) < 1e-6                   # :121-124
    assert np.linalg.norm(blk("variables", "cone_dual") - hx.T) < 1e-6
    assert np.linalg.norm(blk("cone_slack", "cone_dual") + I(nc)) < 1e-6               # :125-130
    assert np.linalg.norm(blk("cone_dual", "cone_slack") + I(nc)) < 1e-6
    assert np.linalg.norm(blk("cone_slack", "cone_slack_dual") + I(nc)) < 1e-6
    assert np.linalg.norm(blk("cone_slack_dual", "cone_slack") - np.diag(t)) < 1e-6    # :131-134
    assert np.linalg.norm(blk("cone_slack_dual", "cone_slack_dual") - (np.diag(s) - ed * I(nc))) < 1e-6
    assert np.linalg.norm(blk("equality_slack", "equality_dual") + I(ne)) < 1e-6       # :135-142
    assert np.linalg.norm(blk("equality_dual", "equality_slack") + I(ne)) < 1e-6
    assert np.linalg.norm(blk("equality_slack", "equality_slack") - (rho + ep) * I(ne)) < 1e-6
    assert np.linalg.norm(blk("cone_slack", "cone_slack") - ep * I(nc)) < 1e-6
    # entries not named by the test are zero
    assert np.count_nonzero(blk("cone_dual", "cone_slack_dual")) == 0 and np.count_nonzero(blk("equality_slack", "variables")) == 0

    K = o.K_dense()
    sx, se, sc = ix["variables"], ix["symmetric_equality"], ix["symmetric_cone"]
    assert np.linalg.matrix_rank(K) == n                                               # :145
    assert np.linalg.norm(K[np.ix_(sx, sx)] - (fxx + gyxx + hzxx + ep * I(nx))) < 1e-6  # :146-147
    assert np.linalg.norm(K[np.ix_(se, sx)] - gx) < 1e-6 and np.linalg.norm(K[np.ix_(sx, se)] - gx.T) < 1e-6
    assert np.linalg.norm(K[np.ix_(se, se)] - (-1.0 / (rho + ep) * I(ne) - ed * I(ne))) < 1e-6   # :152-153
    assert np.linalg.norm(K[np.ix_(sc, sx)] - hx) < 1e-6 and np.linalg.norm(K[np.ix_(sx, sc)] - hx.T) < 1e-6
    assert np.linalg.norm(K[np.ix_(sc, sc)] - np.diag(-1.0 * (s - ed) / (t + (s - ed) * ep) - ed)) < 1e-6   # :158-159
    assert np.count_nonzero(K[np.ix_(se, sc)]) == 0

    res = o.buf("residual")
    y, z, r = pt["y"], pt["z"], pt["r"]
    fx = o.buf("objective_gradient_variables")
    lam = o.buf("dual")
    assert np.linalg.norm(res[ix["variables"]] - (fx + gx.T @ y + hx.T @ z)) < 1e-6     # :162-163
    assert np.linalg.norm(res[ix["equality_slack"]] - (lam + rho * r - y)) < 1e-6       # :165-166
    assert np.linalg.norm(res[ix["cone_slack"]] - (-z - t)) < 1e-6                      # :168-169
    assert np.linalg.norm(res[ix["equality_dual"]] - (o.buf("equality_constraint") - r)) < 1e-6   # :171-172
    assert np.linalg.norm(res[ix["cone_dual"]] - (o.buf("cone_constraint") - s)) < 1e-6            # :174-175
    assert np.linalg.norm(res[ix["cone_slack_dual"]] - (s * t - kappa)) < 1e-6          # :177-178
    rs, rt = res[ix["cone_slack"]], res[ix["cone_slack_dual"]]
    rsym = o.buf("residual_symmetric")
    assert np.linalg.norm(rsym[sx] - res[ix["variables"]]) < 1e-6                        # :184-185
    assert np.linalg.norm(rsym[se] - (o.buf("equality_constraint") - r + res[ix["equality_slack"]] / (rho + ep))) < 1e-6   # :186-187
    assert np.linalg.norm(rsym[sc] - (o.buf("cone_constraint") - s + (rt + (s - ed) * rs) / (t + (s - ed) * ep))) < 1e-6  # :188-189

    # step: symmetric == non-symmetric (:192-204)
    D_full = np.linalg.solve(H, res)
    o.factorize(update=False)
    o.search_direction_symmetric(0, fact=True)
    assert np.linalg.norm(D_full - o.buf("step")) < 1e-6
    # structured H*v equals the dense product
    v = np.random.default_rng(5).standard_normal(N)
    assert np.allclose(o.H_mul(v), H @ v, rtol=0, atol=1e-12)
    # iterative refinement from a noisy step (:207-211)
    o.buf("step")[:] = o.buf("step") + np.random.default_rng(7).standard_normal(N)
    assert o.iterative_refinement()
    assert np.linalg.norm(res - H @ o.buf("step")) < 1e-10


@pytest.mark.parametrize("seed", [0, 1])
def test_second_order_blocks(oracle_mod, seed):
    """6 vars / 3 eq / (2 R+ + SOC3 + SOC4): cone blocks of H are arrow matrices, the condensed system with the FULL
    (non-symmetric) K reproduces the unreduced solve, and the upper-triangle-only factorisation + refinement
    (what the reference actually does) converges to it."""
    soc = [[3, 4, 5], [6, 7, 8, 9]]
    prob = pr.random_qp(6, 3, 9, seed=10 + seed, nonnegative_indices=[1, 2], second_order_indices=soc)
    o = setup_solver(oracle_mod, prob, seed=seed)
    nx, ne, nc, N, n = o.nx, o.ne, o.nc, o.N, o.n
    ep, ed = 0.12, 0.21
    H = o.H_dense()
    pt = o.point()
    s, t = pt["s"], pt["t"]

    def arrow(v):
        d = len(v)
        M = v[0] * np.eye(d)
        M[0, 1:] = v[1:]
        M[1:, 0] = v[1:]
        return M

    it = o.index("cone_slack_dual") - 1
    isl = o.index("cone_slack") - 1
    for c in soc:
        i = np.array(c) - 1
        assert np.allclose(H[np.ix_(it[i], isl[i])], arrow(t[i]))                 # cones/second_order.jl:19-22 via cone.jl:91-95
        assert np.allclose(H[np.ix_(it[i], it[i])], arrow(s[i]) - ed * np.eye(len(i)))
    # cone product / target  (second_order.jl:17,42)
    prod = o.buf("cone_product")
    i = np.array(soc[1]) - 1
    assert np.isclose(prod[i[0]], s[i] @ t[i]) and np.allclose(prod[i[1:]], s[i[0]] * t[i[1:]] + t[i[0]] * s[i[1:]])
    assert np.allclose(o.buf("cone_target"), [1, 1, 1, 0, 0, 1, 0, 0, 0])
    # arrow inverse closed form (second_order.jl:50-65) is an exact inverse
    K = o.K_dense().copy()
    res = o.buf("residual").copy()
    rsym = o.buf("residual_symmetric").copy()
    D_full = np.linalg.solve(H, res)
    dsym = np.linalg.solve(K, rsym)                      # full non-symmetric K: exact condensation
    assert np.allclose(dsym, np.concatenate([D_full[:nx], D_full[o.index("equality_dual") - 1], D_full[o.index("cone_dual") - 1]]), atol=1e-9)
    assert np.linalg.norm(K - K.T) > 1e-4                # not symmetric off the central path
    # inertia of the upper-triangle-symmetrised K
    Ku = np.triu(K) + np.triu(K, 1).T
    w = np.linalg.eigvalsh(Ku)
    o.factorize(update=False)
    assert o.compute_inertia() == (int((w > 0).sum()), int((w <= 0).sum()), 0)
    # reference path: triu-only solve, then refinement against the unreduced H
    o.search_direction_symmetric(0, fact=True)
    first = o.buf("step").copy()
    assert np.linalg.norm(first - D_full, np.inf) > 1e-6
    assert o.iterative_refinement()
    assert np.linalg.norm(o.buf("step") - D_full, np.inf) < 1e-8
    assert np.linalg.norm(res - H @ o.buf("step"), np.inf) <= 1e-10


def test_indices_layout(oracle_mod):
    """indices.jl:25-43, dimensions.jl:17-40: contiguous 1-based ranges (bit-exact integer work)."""
    o = oracle_mod.OracleSolver(4, 2, 3, 5, [1, 2], [[3, 4, 5]])
    assert o.index("variables").tolist() == [1, 2, 3, 4]
    assert o.index("equality_slack").tolist() == [5, 6, 7]
    assert o.index("cone_slack").tolist() == [8, 9, 10, 11, 12]
    assert o.index("equality_dual").tolist() == [13, 14, 15]
    assert o.index("cone_dual").tolist() == [16, 17, 18, 19, 20]
    assert o.index("cone_slack_dual").tolist() == [21, 22, 23, 24, 25]
    assert o.index("symmetric_equality").tolist() == [5, 6, 7]
    assert o.index("symmetric_cone").tolist() == [8, 9, 10, 11, 12]
    assert o.index("primals").tolist() == list(range(1, 13))
    assert o.index("duals").tolist() == list(range(13, 26))
    assert o.index("violation_equality").tolist() == [1, 2, 3] and o.index("violation_cone").tolist() == [4, 5, 6, 7, 8]
    assert o.index("parameters").tolist() == [1, 2]
    assert o.N == 25 and o.n == 12
