"""Stand-alone restatement of the vendored QDLDL (src/solver/qdldl.jl:358-742): structure is integer work and is
checked exactly against an independent construction; values against dense LDL^T / numpy solves."""
import ctypes as C

import numpy as np
import pytest


def _pi(a):
    return a.ctypes.data_as(C.POINTER(C.c_int64))


def _pd(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def triu_csc(A):
    n = A.shape[0]
    Ap = [1]
    Ai, Ax = [], []
    for j in range(n):
        for i in range(j + 1):
            if A[i, j] != 0.0 or i == j:
                Ai.append(i + 1)
                Ax.append(A[i, j])
        Ap.append(len(Ai) + 1)
    return np.array(Ap, dtype=np.int64), np.array(Ai, dtype=np.int64), np.array(Ax, dtype=np.float64)


def quasidefinite(n1, n2, rng, density=0.4):
    A = rng.standard_normal((n1, n1))
    A = A @ A.T + n1 * np.eye(n1)
    B = rng.standard_normal((n2, n1)) * (rng.random((n2, n1)) < density)
    C_ = np.diag(1.0 + rng.random(n2))
    return np.block([[A, B.T], [B, -C_]])


def reference_etree(n, Ap, Ai):
    """elimination tree by the textbook definition: parent(i) = min{ j > i : L[j,i] != 0 } via symbolic elimination"""
    pattern = [set() for _ in range(n)]          # column patterns of L (rows below diagonal)
    for j in range(n):
        for p in range(Ap[j] - 1, Ap[j + 1] - 1):
            i = Ai[p] - 1
            if i < j:
                pattern[i].add(j)
    parent = [-1] * n
    for i in range(n):
        if pattern[i]:
            pj = min(pattern[i])
            parent[i] = pj + 1
            pattern[pj] |= {r for r in pattern[i] if r != pj}
    return parent, [len(p) for p in pattern]


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_qdldl_structure_and_values(oracle_mod, seed):
    L = oracle_mod.lib()
    rng = np.random.default_rng(seed)
    n1, n2 = 9, 6
    K = quasidefinite(n1, n2, rng)
    n = n1 + n2
    Ap, Ai, Ax = triu_csc(K)
    nnz = len(Ai)
    perm = rng.permutation(n).astype(np.int64) + 1
    iperm = np.empty(n, dtype=np.int64)
    iperm[perm - 1] = np.arange(1, n + 1)                         # invperm (qdldl.jl:143)
    Pp = np.zeros(n + 1, dtype=np.int64); Pi = np.zeros(nnz, dtype=np.int64); Px = np.zeros(nnz); A2P = np.zeros(nnz, dtype=np.int64)
    L.oracle_qdldl_permute_symmetric(n, _pi(Ap), _pi(Ai), _pd(Ax), _pi(iperm), _pi(Pp), _pi(Pi), _pd(Px), _pi(A2P))
    # permute_symmetric (qdldl.jl:642-742): PAPt[i,j] = A[perm[i], perm[j]], upper triangular, with the nz map
    PK = K[np.ix_(perm - 1, perm - 1)]
    dense = np.zeros((n, n))
    for j in range(n):
        rows = Pi[Pp[j] - 1:Pp[j + 1] - 1]
        assert np.all(rows <= j + 1)
        dense[rows - 1, j] = Px[Pp[j] - 1:Pp[j + 1] - 1]
    assert np.array_equal(dense, np.triu(PK))                      # bit-exact values, exact structure
    assert sorted(A2P.tolist()) == list(range(1, nnz + 1))       # a bijection
    assert np.array_equal(Px[A2P - 1], Ax)                         # AtoPAPt maps entry k of triu(A) to its slot in PAPt
    # etree / Lnz (qdldl.jl:358-395) vs the textbook definition
    work = np.zeros(n, dtype=np.int64); Lnz = np.zeros(n, dtype=np.int64); etree = np.zeros(n, dtype=np.int64)
    sumLnz = L.oracle_qdldl_etree(n, _pi(Pp), _pi(Pi), _pi(work), _pi(Lnz), _pi(etree))
    parent, counts = reference_etree(n, Pp, Pi)
    assert etree.tolist() == parent and Lnz.tolist() == counts and sumLnz == sum(counts)
    # numeric factor (qdldl.jl:400-589)
    Lp = np.zeros(n + 1, dtype=np.int64); Li = np.zeros(sumLnz, dtype=np.int64); Lx = np.zeros(sumLnz)
    D = np.zeros(n); Dinv = np.zeros(n)
    pos = L.oracle_qdldl_factor(n, _pi(Pp), _pi(Pi), _pd(Px), _pi(Lp), _pi(Li), _pd(Lx), _pd(D), _pd(Dinv), _pi(Lnz), _pi(etree))
    Ld = np.eye(n)
    for j in range(n):
        rows = Li[Lp[j] - 1:Lp[j + 1] - 1]
        assert np.all(np.diff(rows) > 0) and np.all(rows > j + 1)   # up-looking fill order: ascending rows
        Ld[rows - 1, j] = Lx[Lp[j] - 1:Lp[j + 1] - 1]
    assert np.allclose(Ld @ np.diag(D) @ Ld.T, PK, atol=1e-10)
    w = np.linalg.eigvalsh(K)
    assert pos == int((w > 0).sum()) == n1 and int((D <= 0).sum()) == n2
    assert np.array_equal(Dinv, 1.0 / D)
    # solve (qdldl.jl:330-351, 592-640): permute, L, D, L', inverse permute
    b = rng.standard_normal(n)
    tmp = b[perm - 1].copy()
    L.oracle_qdldl_solve(n, _pi(Lp), _pi(Li), _pd(Lx), _pd(Dinv), _pd(tmp))
    x = np.empty(n); x[perm - 1] = tmp
    assert np.allclose(K @ x, b, atol=1e-9)


def test_qdldl_zero_pivot_and_bad_input(oracle_mod):
    L = oracle_mod.lib()
    K = np.array([[1.0, 1.0], [1.0, 1.0]])                         # second pivot is exactly zero
    Ap, Ai, Ax = triu_csc(K)
    n = 2
    work = np.zeros(n, dtype=np.int64); Lnz = np.zeros(n, dtype=np.int64); etree = np.zeros(n, dtype=np.int64)
    s = L.oracle_qdldl_etree(n, _pi(Ap), _pi(Ai), _pi(work), _pi(Lnz), _pi(etree))
    Lp = np.zeros(n + 1, dtype=np.int64); Li = np.zeros(max(s, 1), dtype=np.int64); Lx = np.zeros(max(s, 1)); D = np.zeros(n); Dinv = np.zeros(n)
    assert L.oracle_qdldl_factor(n, _pi(Ap), _pi(Ai), _pd(Ax), _pi(Lp), _pi(Li), _pd(Lx), _pd(D), _pd(Dinv), _pi(Lnz), _pi(etree)) == -1   # qdldl.jl:579
    # a lower-triangular entry / an empty column are rejected by etree (qdldl.jl:366-377)
    Ap2 = np.array([1, 3, 4], dtype=np.int64); Ai2 = np.array([1, 2, 2], dtype=np.int64)
    assert L.oracle_qdldl_etree(2, _pi(Ap2), _pi(Ai2), _pi(work), _pi(Lnz), _pi(etree)) == -1
    Ap3 = np.array([1, 1, 2], dtype=np.int64); Ai3 = np.array([2], dtype=np.int64)
    assert L.oracle_qdldl_etree(2, _pi(Ap3), _pi(Ai3), _pi(work), _pi(Lnz), _pi(etree)) == -1


def test_inertia_correction_quirks(oracle_mod):
    """inertia.jl:30-80 incl. quirk B-1: a non-convex Hessian forces IC-3.. with eps_p restarting at 1e-20 and growing x100"""
    import problems as pr
    prob = pr.random_qp(6, 2, 3, seed=4)
    prob.P = -prob.P                                                # concave objective: wrong inertia at eps_p = 1e-7
    prob.Psym = prob.c * (prob.P + prob.P.T)
    o = oracle_mod.OracleSolver(prob.nx, 0, prob.ne, prob.nc)
    pt = o.point()
    rng = np.random.default_rng(0)
    pt["x"][:] = rng.standard_normal(6); pt["s"][:] = 1.0; pt["t"][:] = 1.0
    prob.evaluate(pr.ALL_VARIABLE_FLAGS, pt["x"], pt["y"], pt["z"], prob.parameters, o.buf)
    o.cone(product=True, jacobian=True, target=True)
    o.buf("central_path")[0] = 1.0; o.buf("penalty")[0] = 1.0
    assert o.inertia_correction() == 0
    ep = o.buf("primal_regularization")[0]
    assert ep == o.buf("primal_regularization_last")[0]
    # the sequence is 1e-20 * 100^k (IC-5 first branch, because primal_regularization_last[1] == 0.0 on the first call)
    k = round(np.log(ep / 1e-20) / np.log(100.0))
    assert np.isclose(ep, 1e-20 * 100.0 ** k, rtol=1e-9) and k > 5
    assert o.compute_inertia() == (6, 5, 0)
    K = o.K_dense(); Ku = np.triu(K) + np.triu(K, 1).T
    assert int((np.linalg.eigvalsh(Ku) > 0).sum()) == 6
    # second call: eps_p restarts from max(1e-20, eps_last/3) and now grows x8 (IC-5 second branch)
    assert o.inertia_correction() == 0
    ep2 = o.buf("primal_regularization")[0]
    m = round(np.log(ep2 / (ep / 3.0)) / np.log(8.0))
    assert np.isclose(ep2, ep / 3.0 * 8.0 ** m, rtol=1e-9)
