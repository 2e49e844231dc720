"""GPU (-m gpu): the LinearSolver seam of the reference (src/solver/linear_solver.jl:1-60) — factorize!(s, A), compute_inertia!(s),
linear_solve!(s, x, A, b) — driven exactly as the reference's own search_direction! / iterative_refinement! / differentiate!
drive `solver.linear_solver` (search_direction.jl:34, iterative_refinement.jl:25, differentiate.jl:19-46, inertia.jl:23-26): the
CALLER assembles the condensed K (here: the oracle, standing in for the Julia reference) and hands its SparseMatrixCSC arrays
and right-hand sides across the C ABI.  Also the handle-based sequence of julia/CalipsoHIP.jl's linear_solve!
(set_field residual_symmetric -> calipso_hip_linear_solve -> get_field step_symmetric).  Tolerances: SURVEY.md 8(c)."""
import numpy as np
import pytest
import scipy.sparse as sp

import problems as pr
from helpers import interior_point, load_pkg, make_pair

pytestmark = pytest.mark.gpu

CASES = {
    "qp_nonneg_10_5_5": lambda: pr.random_qp(10, 5, 5, seed=3),
    "qp_soc_6_3_9": lambda: pr.random_qp(6, 3, 9, seed=10, nonnegative_indices=[1, 2], second_order_indices=[[3, 4, 5], [6, 7, 8, 9]]),
    "qp_mixed_300_120_130": lambda: pr.random_qp(300, 120, 130, seed=8, nonnegative_indices=list(range(1, 41)),
                                                  second_order_indices=[list(range(41 + 3 * k, 44 + 3 * k)) for k in range(30)]),
    "qp_mixed_700_200_90": lambda: pr.random_qp(700, 200, 90, seed=9, nonnegative_indices=list(range(1, 31)),
                                                 second_order_indices=[list(range(31 + 4 * k, 35 + 4 * k)) for k in range(15)]),
}


def rel(a, b):
    return np.abs(np.asarray(a) - np.asarray(b)).max() / max(1.0, np.abs(np.asarray(b)).max())


@pytest.mark.parametrize("case", list(CASES))
def test_reference_search_direction_on_the_device_solver(oracle_mod, case):
    """the reference's search_direction_symmetric! with solver.linear_solver = the device LDL^T: K and b come from the caller"""
    pkg = load_pkg()
    prob = CASES[case]()
    pt, lam = interior_point(prob, seed=1)
    o, g = make_pair(oracle_mod, prob, pt, lam)
    n = o.n
    o.cone(barrier=True, barrier_gradient=True, product=True, jacobian=True, target=True)
    o.residual()
    o.residual_jacobian_variables(); o.residual_jacobian_variables_symmetric()
    K = o.K_dense().copy()                      # both triangles, as the reference writes them (residual_jacobian_variables.jl:110-167)
    o.residual_symmetric(0)
    b = o.buf("residual_symmetric").copy()
    ls = pkg.LDLSolver(n)
    assert ls.factorize(sp.csc_matrix(K)) == 0                      # factorize!(s, A) — only triu(A) is read
    o.factorize(update=False)
    assert ls.compute_inertia() == o.compute_inertia() == (o.nx, o.ne + o.nc, 0)
    x = ls.linear_solve(b)                                          # linear_solve!(s, x, A, b; fact=false)
    x_o = o.linear_solve(b, fact=False)
    assert rel(x, x_o) <= 1e-8
    Ku = np.triu(K); Ksym = Ku + np.triu(K, 1).T                    # what a triu-only factorisation solves with
    assert np.abs(Ksym @ x - b).max() <= 1e-9 * max(1.0, np.abs(b).max())
    # all right-hand sides of differentiate! at once (differentiate.jl:29-58 loops over the columns)
    B = np.random.default_rng(2).standard_normal((n, 7))
    X = ls.linear_solve(B)
    for j in range(7):
        assert rel(X[:, j], o.linear_solve(B[:, j], fact=False)) <= 1e-8
    # triu-only: garbage below the diagonal must not change anything
    K2 = K.copy(); K2[np.tril_indices(n, -1)] = 123.0
    ls.factorize(sp.csc_matrix(K2))
    assert np.array_equal(ls.linear_solve(b), x)
    ls.close()


def test_inertia_of_an_indefinite_and_of_a_singular_matrix(oracle_mod):
    """compute_inertia!: negative = #(d <= 0), zero = #(d == 0); a zero pivot gives positive = -1 and the warning status (qdldl.jl:456,579)"""
    pkg = load_pkg()
    rng = np.random.default_rng(0)
    n = 90
    Q = rng.standard_normal((n, n))
    A = Q @ Q.T + n * np.eye(n)
    A[60:, 60:] = -(A[60:, 60:])                                   # quasi-definite: 60 positive, 30 negative pivots
    A[:60, 60:] *= 0.1; A[60:, :60] = A[:60, 60:].T
    ls = pkg.LDLSolver(n)
    assert ls.factorize(sp.csc_matrix(A)) == 0 and ls.inertia == (60, 30, 0)
    d = np.linalg.eigvalsh(A)
    assert (d > 0).sum() == 60
    b = rng.standard_normal(n)
    assert np.abs(A @ ls.linear_solve(b) - b).max() <= 1e-9
    Z = np.zeros((8, 8)); Z[0, 0] = 1.0; Z[1, 1] = 0.0; Z[2:, 2:] = np.eye(6)
    ls8 = pkg.LDLSolver(8)
    assert ls8.factorize(sp.csc_matrix(Z)) == 1 and ls8.inertia == (-1, 7, 7)      # stale-zero tail of D


def test_handle_linear_solve_sequence_of_the_julia_wrapper(oracle_mod):
    """julia/CalipsoHIP.jl linear_solve!(s::HIPKKTSolver, x, A, b): set_field("residual_symmetric", b) -> calipso_hip_linear_solve ->
    get_field("step_symmetric") on a handle whose blocks / scalars the caller uploaded; result = K \\ b of the reference"""
    prob = CASES["qp_mixed_300_120_130"]()
    pt, lam = interior_point(prob, seed=4)
    o, g = make_pair(oracle_mod, prob, pt, lam)
    o.cone(product=True, jacobian=True, target=True); g.cone(product=True, target=True)
    o.residual_jacobian_variables(); o.residual_jacobian_variables_symmetric()
    o.factorize(update=False)
    inertia, warn = g.factorize()
    assert inertia == o.compute_inertia() and warn == 0
    rng = np.random.default_rng(5)
    for _ in range(3):
        b = rng.standard_normal(o.n)
        g.set("residual_symmetric", b)
        g.linear_solve()
        x = g.get("step_symmetric", o.n)
        assert rel(x, o.linear_solve(b, fact=False)) <= 1e-8


@pytest.mark.parametrize("method", ["rcm", "minimum_degree", "natural"])
def test_seam_with_an_elimination_order(oracle_mod, method):
    """ldl_solver(A) with perm = ordering(A) (the reference: perm = amd(A), qdldl.jl:135): the K of a stage-structured problem, which in the
    natural [x | y | z] order couples everything with everything, becomes banded under reverse Cuthill-McKee and the device factorisation only
    visits the band; the solution does not depend on the order (to rounding)."""
    pkg = load_pkg()
    prob, pt, lam = pr.staged_conic_qp(pkg.splitmix_uniform, 3, 16, 24, 16, 3, 2, 3)      # nx = 384, ne = 240, nc = 144: n = 768
    o = oracle_mod.OracleSolver(prob.nx, 0, prob.ne, prob.nc, prob.nonnegative_indices, prob.second_order_indices)
    o.point()["all"][:] = np.concatenate([pt[k] for k in "xrsyzt"])
    o.buf("dual")[:] = lam
    o.buf("central_path")[0] = 0.17; o.buf("penalty")[0] = 52.0
    o.buf("primal_regularization")[0] = 1e-7; o.buf("dual_regularization")[0] = 1e-7
    op = o.point()
    prob.evaluate(pr.ALL_VARIABLE_FLAGS, op["x"], op["y"], op["z"], np.zeros(0), o.buf)
    o.cone(product=True, jacobian=True, target=True)
    o.residual_jacobian_variables(); o.residual_jacobian_variables_symmetric()
    K = np.array(o.K_dense())
    o.factorize(update=False)
    n = o.n
    A = sp.csc_matrix(K)
    ls = pkg.LDLSolver(n)
    perm, info = ls.analyze(A, method=method)
    assert sorted(perm) == list(range(1, n + 1)) and info["nnz_upper"] == sp.triu(A).nnz
    nblk = 1024 // 64
    if method == "rcm":
        assert 0 < info["band_blocks"] < nblk - 1 and info["half_bandwidth"] < n // 3      # the band is found and used
    if method == "natural":
        assert info["half_bandwidth"] > n // 2                                              # [x | y | z]: the y block reaches back to the first stages
    assert ls.factorize(A) == 0
    assert ls.compute_inertia() == o.compute_inertia() == (prob.nx, prob.ne + prob.nc, 0)     # Sylvester: the same for every order
    rng = np.random.default_rng(7)
    B = rng.standard_normal((n, 5))
    X = ls.linear_solve(B)
    for j in range(5):
        assert rel(X[:, j], o.linear_solve(B[:, j], fact=False)) <= 1e-8
    # a caller-supplied order (method 3) reproduces the same solver state
    ls2 = pkg.LDLSolver(n)
    perm2, info2 = ls2.analyze(A, perm=perm)
    assert np.array_equal(perm2, perm) and info2 == info
    ls2.factorize(A)
    assert np.array_equal(ls2.linear_solve(B), X)
    # the symbolic factor the reference would build for this order (nnz(L)) is reported
    assert info["nnzL"] == pkg.symbolic(sp.triu(A), perm)["nnzL"] > 0
    ls.close(); ls2.close()


@pytest.mark.parametrize("case", ["qp_soc_6_3_9", "qp_mixed_300_120_130", "pendulum"])
@pytest.mark.parametrize("method", ["nested_dissection", "minimum_degree"])
def test_reference_search_direction_on_the_sparse_device_solver(oracle_mod, case, method):
    """the same seam with the SPARSE device solver (calipso_hip_sparse_*: what `HIPSparseLDLSolver <: LinearSolver` binds): the caller's condensed K
    — here also the genuinely sparse K of the pendulum trajectory problem (BASELINE config C2) — analysed once, factored and solved on the device;
    a second factorisation with new values on the same pattern (update = true in linear_solver.jl:24-27)"""
    pkg = load_pkg()
    prob = pr.pendulum(action_guess=np.zeros(10)) if case == "pendulum" else CASES[case]()
    pt, lam = interior_point(prob, seed=1)
    o, g = make_pair(oracle_mod, prob, pt, lam)
    n = o.n
    o.cone(barrier=True, barrier_gradient=True, product=True, jacobian=True, target=True)
    o.residual()
    o.residual_jacobian_variables(); o.residual_jacobian_variables_symmetric()
    K = o.K_dense().copy()
    o.residual_symmetric(0)
    b = o.buf("residual_symmetric").copy()
    A = sp.csc_matrix(np.triu(K)); A.sort_indices()
    ls = pkg.SparseLDL(A, method=method)
    assert ls.factorize(A) == 0
    o.factorize(update=False)
    assert ls.inertia == tuple(o.compute_inertia()) == (o.nx, o.ne + o.nc, 0)
    x = ls.solve(b)
    x_o = o.linear_solve(b, fact=False)
    assert rel(x, x_o) <= 1e-8
    B = np.random.default_rng(2).standard_normal((n, 5))
    X = ls.solve(B)
    for j in range(5):
        assert rel(X[:, j], o.linear_solve(B[:, j], fact=False)) <= 1e-8
    # new values, same pattern: the regularised K of the next inertia-correction try (inertia.jl:23-26)
    K2 = K.copy(); K2[np.arange(o.nx), np.arange(o.nx)] += 1e-3
    rows, cols = A.nonzero()
    A2 = sp.csc_matrix((np.asarray(K2[rows, cols]).ravel(), (rows, cols)), shape=A.shape)      # the values of K2 on the analysed pattern
    A2.sort_indices()
    assert np.array_equal(A2.indices, A.indices) and np.array_equal(A2.indptr, A.indptr)
    assert ls.factorize(A2) == 0
    Ksym2 = np.triu(K2) + np.triu(K2, 1).T
    assert np.abs(Ksym2 @ ls.solve(b) - b).max() <= 1e-8 * max(1.0, np.abs(b).max())
    ls.close()
