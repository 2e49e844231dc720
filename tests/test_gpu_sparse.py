"""Sparse LDL^T on the device (csrc/sparse.hip; SURVEY.md 8(f4) sparse factorisation, 8(f1) stage-parallel elimination) against the oracle's
restatement of the vendored QDLDL (oracle_qdldl_factor / _solve, src/solver/qdldl.jl:400-640) for the SAME permutation: the factor of a
quasi-definite matrix is unique, so L and D must agree to rounding (the summation order differs: left-looking by levels vs up-looking)."""
import ctypes as C

import numpy as np
import pytest
import scipy.sparse as sp

try:                       # before libcalipso_hip.so is loaded: torch brings its own HIP runtime, and one process should hold only one
    import torch
except ImportError:        # (the device-pointer test is skipped without it)
    torch = None

from helpers import load_pkg
from test_oracle_qdldl import quasidefinite
from test_ordering_symbolic import csc1, kkt_matrices, staged_kkt, tree_height

pytestmark = pytest.mark.gpu


def _pi(a):
    return a.ctypes.data_as(C.POINTER(C.c_int64))


def _pd(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def oracle_factor(oracle_mod, K, perm):
    """QDLDL of triu(K) under perm by the oracle: dense unit-lower L, D, and a solve closure"""
    L = oracle_mod.lib()
    n = K.shape[0]
    Ap, Ai, Ax = csc1(sp.triu(sp.csc_matrix(K)))
    nnz = len(Ai)
    iperm = np.zeros(n, dtype=np.int64); iperm[perm - 1] = np.arange(1, n + 1)
    Pp, Pi, Px, mp = np.zeros(n + 1, dtype=np.int64), np.zeros(nnz, dtype=np.int64), np.zeros(nnz), np.zeros(nnz, dtype=np.int64)
    L.oracle_qdldl_permute_symmetric(n, _pi(Ap), _pi(Ai), _pd(Ax), _pi(iperm), _pi(Pp), _pi(Pi), _pd(Px), _pi(mp))
    work, Lnz, et = np.zeros(n, dtype=np.int64), np.zeros(n, dtype=np.int64), np.zeros(n, dtype=np.int64)
    tot = L.oracle_qdldl_etree(n, _pi(Pp), _pi(Pi), _pi(work), _pi(Lnz), _pi(et))
    assert tot >= 0
    Lp, Li, Lx = np.zeros(n + 1, dtype=np.int64), np.zeros(max(tot, 1), dtype=np.int64), np.zeros(max(tot, 1))
    D, Dinv = np.zeros(n), np.zeros(n)
    pos = L.oracle_qdldl_factor(n, _pi(Pp), _pi(Pi), _pd(Px), _pi(Lp), _pi(Li), _pd(Lx), _pd(D), _pd(Dinv), _pi(Lnz), _pi(et))
    Lm = sp.csc_matrix((Lx[:tot], Li[:tot] - 1, Lp - 1), shape=(n, n)) + sp.identity(n, format="csc")

    def solve(b):
        tmp = b[perm - 1].copy()
        L.oracle_qdldl_solve(n, _pi(Lp), _pi(Li), _pd(Lx), _pd(Dinv), _pd(tmp))
        x = np.empty(n); x[perm - 1] = tmp
        return x
    return dict(L=Lm, D=D, positive=int(pos), nnzL=int(tot), solve=solve)


def cases():
    rng = np.random.default_rng(11)
    out = dict(kkt_matrices())
    out["quasidefinite_60"] = quasidefinite(40, 20, rng, density=0.15)
    out["staged_12x(4+2)"] = staged_kkt(12, 4, 2, rng).toarray()
    return out


@pytest.mark.parametrize("name", ["quasidefinite_15", "pendulum", "conic_qp", "quasidefinite_60", "staged_12x(4+2)"])
@pytest.mark.parametrize("method", ["natural", "rcm", "minimum_degree", "nested_dissection", "nested_dissection_columns", "random"])
def test_factor_and_solve_match_the_oracle(oracle_mod, name, method):
    pkg = load_pkg()
    K = cases()[name]
    n = K.shape[0]
    A = sp.csc_matrix(sp.triu(sp.csc_matrix(K)))
    rng = np.random.default_rng(3)
    if method == "random":
        S = pkg.SparseLDL(A, perm=rng.permutation(n) + 1)
    else:
        S = pkg.SparseLDL(A, method=method)
    rc = S.factorize(A)
    perm, Lm, D = S.factor()
    assert sorted(perm.tolist()) == list(range(1, n + 1))
    ref = oracle_factor(oracle_mod, K, perm)
    assert rc == 0 and S.info["nnzL"] == ref["nnzL"]
    assert S.inertia == (ref["positive"], int((ref["D"] <= 0).sum()), 0)
    scale = max(1.0, np.abs(ref["L"].toarray()).max())
    assert np.array_equal((Lm != 0).toarray() | (ref["L"].toarray() != 0), (ref["L"] != 0).toarray() | (Lm.toarray() != 0))
    assert np.abs(Lm.toarray() - ref["L"].toarray()).max() <= 1e-11 * scale
    assert np.abs(D - ref["D"]).max() <= 1e-11 * np.abs(ref["D"]).max()
    PK = K[np.ix_(perm - 1, perm - 1)]
    assert np.allclose(Lm.toarray() @ np.diag(D) @ Lm.toarray().T, PK, atol=1e-10 * np.abs(K).max())
    B = rng.standard_normal((n, 3))
    X = S.solve(B)
    for c in range(3):
        xo = ref["solve"](B[:, c].copy())
        assert np.abs(X[:, c] - xo).max() <= 1e-9 * max(1.0, np.abs(xo).max())
    assert np.abs(S.solve(B[:, 0]) - X[:, 0]).max() == 0.0           # same launches, same order: bit-reproducible
    S.close()


def test_refactorisation_with_new_values_and_reproducibility(oracle_mod):
    pkg = load_pkg()
    rng = np.random.default_rng(5)
    K0 = staged_kkt(10, 3, 2, rng)
    S = pkg.SparseLDL(sp.triu(K0).tocsc(), method="nested_dissection")
    for trial in range(3):
        K = K0.copy()
        K.data = K.data * (1.0 + 0.1 * rng.random(K.data.size))
        K = ((K + K.T) * 0.5).tocsc()
        A = sp.triu(K).tocsc()
        assert S.factorize(A) == 0
        perm, Lm, D = S.factor()
        ref = oracle_factor(oracle_mod, K.toarray(), perm)
        assert np.abs(D - ref["D"]).max() <= 1e-11 * np.abs(ref["D"]).max()
        S.factorize(A)
        _, Lm2, D2 = S.factor()
        assert np.array_equal(D, D2) and np.array_equal(Lm.toarray(), Lm2.toarray())      # no atomics, fixed summation order
    S.close()


def test_zero_pivot_is_reported_like_qdldl(oracle_mod):
    pkg = load_pkg()
    K = np.array([[1.0, 1.0, 0.0], [1.0, 1.0, 0.0], [0.0, 0.0, 2.0]])      # second pivot exactly zero (qdldl.jl:456,579)
    S = pkg.SparseLDL(sp.triu(sp.csc_matrix(K)).tocsc(), method="natural")
    assert S.factorize(sp.triu(sp.csc_matrix(K)).tocsc()) == 1
    assert S.inertia == (-1, 2, 2)
    S.close()
    with pytest.raises(pkg.CalipsoHipError):
        pkg.SparseLDL(sp.csc_matrix(np.array([[0.0, 0.0], [0.0, 1.0]])), method="natural")   # empty column: QDLDL_etree! fails (qdldl.jl:366-371)


def test_nested_dissection_eliminates_the_stages_in_parallel(oracle_mod):
    """a T-stage trajectory KKT system.  In the natural (stage-by-stage) order the elimination tree is a chain through all the
    stages; under nested dissection the two halves of the horizon are independent sub-trees, recursively: the sequential depth (levels) grows
    like log T, and the factorisation / solves still match the oracle and a dense solve."""
    pkg = load_pkg()
    rng = np.random.default_rng(9)
    T, ns, nu = 64, 6, 2
    K = staged_kkt(T, ns, nu, rng)
    n = K.shape[0]
    A = sp.triu(K).tocsc()
    nat = pkg.SparseLDL(A, method="natural")
    nd = pkg.SparseLDL(A, method="nested_dissection_columns")
    mf = pkg.SparseLDL(A, method="nested_dissection")
    assert nat.info["levels"] > 0.3 * n                       # a chain through the horizon
    assert nd.info["levels"] < 0.2 * nat.info["levels"]       # parallel sub-trees
    assert nd.info["widest_level"] >= T // 2
    assert mf.info["numeric"] == "multifrontal" and mf.info["levels"] <= 12 and mf.info["launches"] == mf.info["levels"]   # ~log2 T levels of fronts
    b = rng.standard_normal(n)
    xd = np.linalg.solve(K.toarray(), b)
    for S in (nat, nd, mf):
        assert S.factorize(A) == 0
        assert S.inertia == (T * (ns + nu) + ns, T * ns, 0)
        x = S.solve(b)
        assert np.abs(x - xd).max() <= 1e-9 * np.abs(xd).max()
    for S in (nd, mf):
        perm, Lm, D = S.factor()
        ref = oracle_factor(oracle_mod, K.toarray(), perm)
        assert np.abs(D - ref["D"]).max() <= 1e-11 * np.abs(ref["D"]).max()
        assert abs(Lm - ref["L"]).max() <= 1e-11 * max(1.0, abs(ref["L"]).max())
    perm, _, _ = nd.factor()
    assert tree_height(pkg.symbolic(A, perm)["etree"]) == nd.info["levels"]
    nat.close(); nd.close(); mf.close()


def test_rate_report_on_trajectory_kkt_systems(oracle_mod):
    """Timings (device events) of the level-scheduled factorisation / solve on stage-structured KKT systems, natural order vs nested dissection,
    next to the oracle's QDLDL on one host core for the same permutation; written to gpurun_out/sparse_ldl_rate.json (copied to profiles/).
    Asserts only correctness and that nested dissection shortens the device factorisation."""
    import json, os, time
    pkg = load_pkg()
    rng = np.random.default_rng(2)
    rows = []
    for T, ns, nu in ((256, 6, 2), (512, 12, 4), (2048, 12, 4), (41, 28, 28), (41, 50, 50)):   # (41, 28, 28): BASELINE C4's size, 41 stages of 56 variables; (41, 50, 50): stages of 100 variables, fronts of 300 rows in global memory
        K = staged_kkt(T, ns, nu, rng)
        n = K.shape[0]
        A = sp.triu(K).tocsc()
        b = rng.standard_normal(n)
        row = dict(T=T, state=ns, control=nu, n=n, nnz_upper=int(A.nnz))
        for method in (("natural", "nested_dissection_columns", "nested_dissection") if 256 <= T <= 512 else ("nested_dissection_columns", "nested_dissection")):
            S = pkg.SparseLDL(A, method=method)
            S.factorize(A); S.solve(b)                       # warm-up (graph capture, allocations)
            f, s = [], []
            for _ in range(5):
                assert S.factorize(A) == 0
                x = S.solve(b)
                tf, ts = S.timing(); f.append(tf); s.append(ts)
            assert np.abs(K @ x - b).max() <= 1e-8 * max(1.0, np.abs(b).max()) * max(1.0, np.abs(x).max())
            perm, _, D = S.factor()
            t0 = time.perf_counter(); ref = oracle_factor(oracle_mod, K, perm); t_cpu = time.perf_counter() - t0
            assert np.abs(D - ref["D"]).max() <= 1e-10 * np.abs(ref["D"]).max()
            row[method] = dict(S.info, factor_ms=min(f), solve_ms=min(s), oracle_qdldl_analyse_plus_factor_ms_1core=1e3 * t_cpu)
            S.close()
        assert "natural" not in row or row["nested_dissection_columns"]["factor_ms"] < row["natural"]["factor_ms"]
        assert row["nested_dissection"]["numeric"] == "multifrontal" and row["nested_dissection"]["factor_ms"] < row["nested_dissection_columns"]["factor_ms"]
        rows.append(row)
    # a batch of independent systems of one structure (BASELINE config C4's shape of work), multifrontal: all matrices in the same launches
    for T, ns, nu, Bn in ((64, 6, 2, 256), (256, 6, 2, 64), (41, 14, 14, 256), (41, 28, 28, 64)):
        K = staged_kkt(T, ns, nu, rng)
        n = K.shape[0]
        A = sp.triu(K).tocsc(); A.sort_indices()
        S = pkg.SparseLDL(A, method="nested_dissection")
        vals = A.data[None, :] * (1.0 + 0.05 * rng.random((Bn, A.nnz)))
        S.set_batch(Bn)
        bb = rng.standard_normal((Bn, n))
        S.factorize(vals); S.solve(bb)
        f, sv = [], []
        for _ in range(5):
            assert S.factorize(vals) == 0
            x = S.solve(bb)
            tf, ts = S.timing(); f.append(tf); sv.append(ts)
        z = Bn - 1
        Kz = sp.csc_matrix((vals[z], A.indices, A.indptr), shape=A.shape); Kz = Kz + sp.triu(Kz, 1).T
        assert np.abs(Kz @ x[z] - bb[z]).max() <= 1e-8 * max(1.0, np.abs(x[z]).max())
        rows.append(dict(T=T, state=ns, control=nu, n=n, batch=Bn, numeric=S.info["numeric"], levels=S.info["levels"], factor_ms_whole_batch=min(f),
                         solve_ms_whole_batch=min(sv), factorisations_per_s=Bn / (1e-3 * min(f)), solves_per_s=Bn / (1e-3 * min(sv))))
        S.close()
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/sparse_ldl_rate.json", "w") as fh:
        json.dump(rows, fh, indent=1)
    print(json.dumps(rows))


def test_global_accumulator_path_beyond_the_lds_limit(oracle_mod):
    """n above what one CU's LDS can hold as a column accumulator (19 400 doubles): the accumulators live in global scratch, one per resident
    workgroup; same parity bar"""
    pkg = load_pkg()
    rng = np.random.default_rng(21)
    K = staged_kkt(700, 12, 4, rng)
    n = K.shape[0]
    assert n > 19400
    A = sp.triu(K).tocsc()
    S = pkg.SparseLDL(A, method="nested_dissection_columns")
    assert S.info["numeric"] == "columns_global_accumulator" and S.info["levels"] < 200
    assert S.factorize(A) == 0
    assert S.inertia == (700 * 16 + 12, 700 * 12, 0)
    perm, Lm, D = S.factor()
    ref = oracle_factor(oracle_mod, K, perm)
    assert np.abs(D - ref["D"]).max() <= 1e-10 * np.abs(ref["D"]).max()
    assert abs(Lm - ref["L"]).max() <= 1e-10 * max(1.0, abs(ref["L"]).max())
    b = rng.standard_normal((n, 2))
    x = S.solve(b)
    assert np.abs(K @ x - b).max() <= 1e-8 * max(1.0, np.abs(x).max())
    S.close()


def test_fronts_larger_than_the_lds_live_in_global_memory(oracle_mod):
    """a 2-D grid has separators of ~sqrt(n) vertices: the top fronts (hundreds of rows) exceed one CU's LDS and are factored in global memory by
    many workgroups per front (round 5, csrc/sparse_wide.hpp) — up to 8192 rows (a dense block of 1500 and a mesh with a 1500-vertex separator below); beyond that
    the column method takes the whole matrix — same answers either way"""
    pkg = load_pkg()
    g = 180
    I = sp.identity(g, format="csc")
    T1 = sp.diags([-1.0, 2.5, -1.0], [-1, 0, 1], shape=(g, g), format="csc")
    K = (sp.kron(I, T1) + sp.kron(T1, I)).tocsc()                # SPD (hence quasi-definite), n = 32 400
    K.sort_indices()
    A = sp.triu(K).tocsc()
    S = pkg.SparseLDL(A, method="nested_dissection")
    assert S.info["numeric"] == "multifrontal"
    assert S.factorize(A) == 0 and S.inertia == (g * g, 0, 0)
    rng = np.random.default_rng(1)
    b = rng.standard_normal((g * g, 2))
    x = S.solve(b)
    assert np.abs(K @ x - b).max() <= 1e-9 * max(1.0, np.abs(x).max())
    perm, Lm, D = S.factor()
    ref = oracle_factor(oracle_mod, K, perm)
    assert np.abs(D - ref["D"]).max() <= 1e-11 * np.abs(ref["D"]).max()
    assert abs(Lm - ref["L"]).max() <= 1e-10 * max(1.0, abs(ref["L"]).max())
    tf, ts = S.timing()
    C = pkg.SparseLDL(A, method="nested_dissection_columns")
    C.factorize(A); C.factorize(A)
    assert tf < C.timing()[0]                                    # the tree-parallel fronts beat the column method on this matrix
    S.close(); C.close()
    # one dense block of 1500: its first front has 1500 rows — a global-memory front (round 3: the column method took over at 1024)
    M = rng.standard_normal((1500, 1500)); Kd = M @ M.T + 1500 * np.eye(1500)
    Ad = sp.csc_matrix(np.triu(Kd))
    Sd = pkg.SparseLDL(Ad, method="nested_dissection")
    assert Sd.info["numeric"] == "multifrontal"
    assert Sd.factorize(Ad) == 0
    bd = rng.standard_normal(1500)
    assert np.abs(Kd @ Sd.solve(bd) - bd).max() <= 1e-8 * np.abs(bd).max() * 10
    Sd.factorize(Ad)
    Cd = pkg.SparseLDL(Ad, method="nested_dissection_columns")
    Cd.factorize(Ad); Cd.factorize(Ad)
    print("dense block of 1500: fronts %.2f ms, column method %.2f ms" % (Sd.timing()[0], Cd.timing()[0]))
    Sd.close(); Cd.close()
    # two 60 x 60 meshes joined through a separator of 1500 vertices (every separator vertex touches its neighbours in the separator and one vertex of each mesh)
    gm, ns = 60, 1500
    Tm = sp.diags([-1.0, 2.5, -1.0], [-1, 0, 1], shape=(gm, gm), format="csc")
    Km = (sp.kron(sp.identity(gm), Tm) + sp.kron(Tm, sp.identity(gm))).tocsc()
    nm = gm * gm
    Ssep = sp.diags([-0.5, 6.0, -0.5], [-1, 0, 1], shape=(ns, ns), format="lil")
    band = 40
    for off in range(2, band):
        Ssep.setdiag(-0.01, off); Ssep.setdiag(-0.01, -off)
    C1 = sp.lil_matrix((ns, nm)); C2 = sp.lil_matrix((ns, nm))
    for i in range(ns):
        C1[i, (i * 7) % nm] = -0.3; C2[i, (i * 11) % nm] = -0.3
    Kw = sp.bmat([[Km, None, C1.T], [None, Km, C2.T], [C1, C2, Ssep]], format="csc")
    Kw.sort_indices()
    Aw = sp.triu(Kw).tocsc()
    Sw = pkg.SparseLDL(Aw, method="nested_dissection")
    assert Sw.factorize(Aw) == 0 and Sw.inertia[2] == 0
    bw = rng.standard_normal(Kw.shape[0])
    xw = Sw.solve(bw)
    assert np.abs(Kw @ xw - bw).max() <= 1e-8 * max(1.0, np.abs(xw).max())
    Sw.factorize(Aw)
    Cw = pkg.SparseLDL(Aw, method="nested_dissection_columns")
    Cw.factorize(Aw); Cw.factorize(Aw)
    print("two meshes + separator of 1500 (n = %d): %s, largest front %s rows, fronts %.2f ms, column method %.2f ms" % (
        Kw.shape[0], Sw.info["numeric"], Sw.info.get("largest_front", "?"), Sw.timing()[0], Cw.timing()[0]))
    Sw.close(); Cw.close()


@pytest.mark.parametrize("method", ["nested_dissection", "nested_dissection_columns"])
def test_a_batch_of_matrices_with_one_pattern(oracle_mod, method):
    """BASELINE config C4's shape of work for the LinearSolver seam: many independent KKT systems of ONE structure.  The multifrontal path factors
    the whole batch in the same launches; every matrix gets the bits it gets alone."""
    pkg = load_pkg()
    rng = np.random.default_rng(13)
    K0 = staged_kkt(24, 5, 2, rng)
    n, nnz = K0.shape[0], sp.triu(K0).nnz
    A0 = sp.triu(K0).tocsc(); A0.sort_indices()
    Bn = 7
    mats, vals = [], np.zeros((Bn, A0.nnz))
    for z in range(Bn):
        K = K0.copy(); K.data = K.data * (1.0 + 0.2 * rng.random(K.data.size)); K = ((K + K.T) * 0.5).tocsc(); K.sort_indices()
        mats.append(K)
        Az = sp.triu(K).tocsc(); Az.sort_indices()
        assert np.array_equal(Az.indices, A0.indices)
        vals[z] = Az.data
    vals[3] = vals[3] * 1.0
    S = pkg.SparseLDL(A0, method=method)
    alone = []
    for z in range(Bn):
        assert S.factorize(vals[z]) == 0
        perm, Lm, D = S.factor()
        alone.append((Lm.toarray(), D, S.solve(np.arange(1.0, n + 1.0))))
    S.set_batch(Bn)
    assert S.factorize(vals) == 0
    assert np.array_equal(S.inertia_all, np.tile(np.array(S.inertia_all[0]), (Bn, 1))) and S.inertia_all[0][0] == 24 * 7 + 5
    rhs = np.tile(np.arange(1.0, n + 1.0), (Bn, 1))
    X = S.solve(rhs)
    X3 = S.solve(np.stack([rhs, 2.0 * rhs], axis=2))
    for z in range(Bn):
        S.select(z)
        perm, Lm, D = S.factor()
        assert np.array_equal(Lm.toarray(), alone[z][0]) and np.array_equal(D, alone[z][1])      # same bits as alone
        assert np.array_equal(X[z], alone[z][2]) and np.array_equal(X3[z, :, 0], alone[z][2])
        assert np.abs(mats[z] @ X3[z, :, 1] - 2.0 * rhs[z]).max() <= 1e-8 * np.abs(X3[z]).max()
        ref = oracle_factor(oracle_mod, mats[z].toarray(), perm)
        assert np.abs(D - ref["D"]).max() <= 1e-11 * np.abs(ref["D"]).max()
    S.close()


def test_zero_pivot_in_the_multifrontal_path():
    """60 decoupled 2 x 2 blocks [[1, 1], [1, 1]]: every second pivot is exactly zero.  QDLDL stops at the first one in elimination order
    (positive = -1, the rest of D counted as zeros, qdldl.jl:444,456,579); the fronts do not stop, the reported inertia is the same"""
    pkg = load_pkg()
    K = sp.block_diag([np.ones((2, 2))] * 60, format="csc")
    S = pkg.SparseLDL(sp.triu(K).tocsc(), method="nested_dissection")
    assert S.info["numeric"] == "multifrontal"
    assert S.factorize(sp.triu(K).tocsc()) == 1
    perm, _, D = S.factor()
    first_zero = int(np.argmax(D == 0.0))
    assert D[first_zero] == 0.0 and S.inertia == (-1, 120 - first_zero, 120 - first_zero)
    S.close()


@pytest.mark.parametrize("seed", [0, 1])
def test_wide_fronts_on_random_quasidefinite_structures(seed):
    """a banded SPD block with a dense arrow (40 variables coupled to everything), sparse random coupling to a negative-definite block: nested dissection puts the arrow and
    the band's separators into fronts of several hundred rows with MANY children each (the per-row child lists of k_wf_assemble / k_wfs_head / k_wfs_tail) — inertia and
    residual against the matrix itself, three right-hand sides, a batch of two"""
    pkg = load_pkg()
    rng = np.random.default_rng(100 + seed)
    n1, nq, na = 2500, 300, 40
    band = sp.diags([rng.uniform(-0.3, 0.3, n1 - k) for k in range(1, 25)], list(range(1, 25)), shape=(n1, n1))
    A = (band + band.T + sp.diags(np.full(n1, 30.0))).tolil()
    arrow = rng.standard_normal((na, n1)) * 0.2
    A[:na, :] = arrow; A[:, :na] = arrow.T
    A[:na, :na] = arrow[:, :na] @ arrow[:, :na].T + 60.0 * np.eye(na)
    A = sp.csc_matrix(A); A = ((A + A.T) * 0.5).tocsc()
    B = sp.random(nq, n1, density=0.004, random_state=np.random.RandomState(seed), data_rvs=lambda k: rng.standard_normal(k)).tocsc()
    C = sp.diags(rng.uniform(0.5, 2.0, nq))
    K = sp.bmat([[A, B.T], [B, -C]], format="csc"); K.sort_indices()
    U = sp.triu(K).tocsc()
    S = pkg.SparseLDL(U, method="nested_dissection")
    assert S.info["numeric"] == "multifrontal"
    assert S.factorize(U) == 0 and S.inertia == (n1, nq, 0)
    b = rng.standard_normal((n1 + nq, 3))
    x = S.solve(b)
    assert np.abs(K @ x - b).max() <= 1e-9 * max(1.0, np.abs(x).max())
    _, L1, D1 = S.factor()
    S.set_batch(2)
    U2 = U.copy(); U2.data = U2.data * (1.0 + 0.001 * rng.standard_normal(U2.nnz))
    U2 = (U2 + sp.diags(np.r_[np.full(n1, 5.0), np.full(nq, -5.0)])).tocsc(); U2.sort_indices()
    assert np.array_equal(U2.indices, U.indices)
    S.factorize(np.stack([U2.data, U.data]))
    S.select(1)
    _, L2, D2 = S.factor()
    assert np.array_equal(D1, D2) and (L1 != L2).nnz == 0            # in a batch: the bits it gets alone
    K2 = (U2 + sp.triu(U2, 1).T).tocsc()
    xb = S.solve(np.stack([b[:, 0], b[:, 0]]))
    assert np.abs(K2 @ xb[0] - b[:, 0]).max() <= 1e-9 * max(1.0, np.abs(xb[0]).max())
    assert np.array_equal(xb[1], x[:, 0])
    S.close()


def test_zero_pivot_in_a_front_factored_by_many_workgroups():
    """a 300 x 300 matrix of ones (rank one: the second pivot is exactly 1 - 1 = 0 whatever the order) is one clique = fronts of up to 300 rows in global memory
    (csrc/sparse_wide.hpp): the zero pivot is reported as QDLDL reports it (positive = -1, everything from it on counted as zero)"""
    pkg = load_pkg()
    K = sp.csc_matrix(np.ones((300, 300)))
    S = pkg.SparseLDL(sp.triu(K).tocsc(), method="nested_dissection")
    assert S.info["numeric"] == "multifrontal"
    assert S.factorize(sp.triu(K).tocsc()) == 1
    _, _, D = S.factor()
    assert D[0] == 1.0 and D[1] == 0.0 and S.inertia == (-1, 299, 299)
    S.close()


def test_device_resident_values_and_right_hand_sides():
    """calipso_hip_sparse_factorize_device / _solve_device: nothing but pointers crosses the boundary; same bits as the host-array calls"""
    if torch is None:
        pytest.skip("torch not installed")
    pkg = load_pkg()
    rng = np.random.default_rng(17)
    K = staged_kkt(32, 5, 2, rng)
    A = sp.triu(K).tocsc(); A.sort_indices()
    n = K.shape[0]
    S = pkg.SparseLDL(A, method="nested_dissection")
    S.set_batch(3)
    vals = A.data[None, :] * (1.0 + 0.1 * rng.random((3, A.nnz)))
    b = rng.standard_normal((3, n))
    S.factorize(vals)
    x_host = S.solve(b)
    tv = torch.from_numpy(vals).to("cuda:0")
    tb = torch.from_numpy(b.reshape(3, 1, n).copy()).to("cuda:0")
    assert S.factorize_device(tv) == 0
    tx = S.solve_device(tb)
    torch.cuda.synchronize()
    assert np.array_equal(tx.cpu().numpy().reshape(3, n), x_host)
    S.close()


def test_batch_with_fronts_in_global_memory():
    """a batch of matrices whose fronts (stages of 100 variables: 300-row fronts) live in the global-memory pool: every matrix has its own pool slice"""
    pkg = load_pkg()
    rng = np.random.default_rng(23)
    K0 = staged_kkt(9, 50, 50, rng)
    A0 = sp.triu(K0).tocsc(); A0.sort_indices()
    n = K0.shape[0]
    S = pkg.SparseLDL(A0, method="nested_dissection")
    assert S.info["numeric"] == "multifrontal"
    Bn = 3
    vals = A0.data[None, :] * (1.0 + 0.05 * rng.random((Bn, A0.nnz)))
    alone = []
    b = rng.standard_normal(n)
    for z in range(Bn):
        assert S.factorize(vals[z]) == 0
        alone.append(S.solve(b))
    S.set_batch(Bn)
    assert S.factorize(vals) == 0
    X = S.solve(np.tile(b, (Bn, 1)))
    for z in range(Bn):
        assert np.array_equal(X[z], alone[z])
        Kz = sp.csc_matrix((vals[z], A0.indices, A0.indptr), shape=A0.shape); Kz = Kz + sp.triu(Kz, 1).T
        assert np.abs(Kz @ X[z] - b).max() <= 1e-8 * max(1.0, np.abs(X[z]).max())
    S.close()


def test_error_paths_of_the_sparse_handle():
    """the reference raises on these (DimensionMismatch / BoundsError / "factorize first" has no counterpart); here: a negative status with a message"""
    pkg = load_pkg()
    rng = np.random.default_rng(3)
    K = staged_kkt(6, 3, 2, rng)
    A = sp.triu(K).tocsc(); A.sort_indices()
    n = K.shape[0]
    with pytest.raises(pkg.CalipsoHipError):
        pkg.SparseLDL(A, perm=np.arange(n))                       # 0-based: not a permutation of 1:n
    with pytest.raises(pkg.CalipsoHipError):
        pkg.SparseLDL(A, perm=np.r_[np.arange(1, n), n - 1])      # a repeated vertex
    S = pkg.SparseLDL(A, method="nested_dissection")
    with pytest.raises(pkg.CalipsoHipError):
        S.solve(np.ones(n))                                       # no factorisation yet
    with pytest.raises(pkg.CalipsoHipError):
        S.factorize(np.ones(A.nnz - 1))                           # wrong number of values
    with pytest.raises(pkg.CalipsoHipError):
        S.factorize(sp.triu(staged_kkt(6, 3, 3, rng)).tocsc())    # another pattern
    assert S.factorize(A) == 0
    with pytest.raises(pkg.CalipsoHipError):
        S.select(1)                                               # only one matrix in the batch
    with pytest.raises(pkg.CalipsoHipError):
        S.set_batch(0)
    S.set_batch(2)
    with pytest.raises(pkg.CalipsoHipError):
        S.solve(np.ones((2, n)))                                  # set_batch invalidated the factorisation
    assert S.factorize(np.stack([A.data, 2.0 * A.data])) == 0
    X = S.solve(np.ones((2, n)))
    assert np.abs(2.0 * X[1] - X[0]).max() <= 1e-12 * np.abs(X[0]).max()
    S.close()
    S.close()                                                     # idempotent


def _wide_fronts(pkg, on):
    """calipso_hip_debug_wide_fronts: plans made afterwards give the global-memory fronts to many workgroups (1, the default) or to one (0); returns the old value"""
    fn = pkg._lib.lib().calipso_hip_debug_wide_fronts
    fn.restype = C.c_int32
    return fn(C.c_int32(on))


def test_wide_fronts_by_many_workgroups_match_the_one_workgroup_kernel_and_the_oracle(oracle_mod):
    """round 5: a front beyond the LDS is assembled, factored and pushed up by a handful of multi-workgroup launches (sparse_wide.hpp) instead of one
    workgroup: same factor as the oracle's QDLDL (1e-10), same as the one-workgroup kernel to rounding, same bits from run to run, a batch gets the
    bits a matrix gets alone — on a quasi-definite matrix (pivots of both signs in the wide fronts) with fronts of ~300 to 1100 rows and nodes of fewer
    than 64 columns, and faster on the 1500-row cases of the test above."""
    pkg = load_pkg()
    rng = np.random.default_rng(5)
    # two meshes joined through a dense-ish separator of 700 + a quasi-definite tail: the separator's chunks are wide fronts with children
    gm, ns, nq = 24, 700, 150
    Tm = sp.diags([-1.0, 2.5, -1.0], [-1, 0, 1], shape=(gm, gm), format="csc")
    Km = (sp.kron(sp.identity(gm), Tm) + sp.kron(Tm, sp.identity(gm))).tocsc()
    nm = gm * gm
    B = rng.standard_normal((ns, ns)) * (rng.random((ns, ns)) < 0.3)
    Ssep = sp.csc_matrix(B @ B.T / ns + 4.0 * np.eye(ns))
    C1 = sp.lil_matrix((ns, nm)); C2 = sp.lil_matrix((ns, nm))
    for i in range(ns):
        C1[i, (i * 7) % nm] = -0.3; C2[i, (i * 11) % nm] = -0.3
    G = sp.csc_matrix(rng.standard_normal((nq, ns)) * (rng.random((nq, ns)) < 0.2))
    Kw = sp.bmat([[Km, None, C1.T, None], [None, Km, C2.T, None], [C1, C2, Ssep, G.T], [None, None, G, -0.5 * sp.identity(nq)]], format="csc")
    Kw.sort_indices()
    Aw = sp.triu(Kw).tocsc()
    n = Kw.shape[0]
    b = rng.standard_normal((n, 3))
    was = _wide_fronts(pkg, 1)
    try:
        W = pkg.SparseLDL(Aw, method="nested_dissection")
        assert W.info["numeric"] == "multifrontal"
        assert W.factorize(Aw) == 0
        assert W.inertia == (2 * nm + ns, nq, 0)
        perm, Lw, Dw = W.factor()
        xw = W.solve(b)
        assert np.abs(Kw @ xw - b).max() <= 1e-9 * max(1.0, np.abs(xw).max())
        ref = oracle_factor(oracle_mod, Kw, perm)
        assert np.abs(Dw - ref["D"]).max() <= 1e-11 * np.abs(ref["D"]).max()
        assert abs(Lw - ref["L"]).max() <= 1e-10 * max(1.0, abs(ref["L"]).max())
        W.factorize(Aw)
        _, Lw2, Dw2 = W.factor()
        assert np.array_equal(Dw, Dw2) and (Lw != Lw2).nnz == 0           # run to run: the same bits
        tw = W.timing()[0]
        # a batch of three (the second and third with other values): every matrix gets the bits it gets alone
        A2 = Aw.copy(); A2.data = A2.data * (1.0 + 0.01 * rng.standard_normal(A2.nnz)); A2 = A2 + sp.diags(np.r_[np.full(2 * nm + ns, 3.0), np.full(nq, -3.0)])
        A2 = sp.triu(A2).tocsc(); A2.sort_indices()
        assert np.array_equal(A2.indices, Aw.indices)
        W.factorize(A2)
        _, L2, D2 = W.factor()
        W.set_batch(3)
        W.factorize(np.stack([Aw.data, A2.data, Aw.data]))
        for z, (Lr, Dr) in enumerate(((Lw, Dw), (L2, D2), (Lw, Dw))):
            W.select(z)
            _, Lz, Dz = W.factor()
            assert np.array_equal(Dz, Dr) and (Lz != Lr).nnz == 0
        W.close()
        _wide_fronts(pkg, 0)
        O = pkg.SparseLDL(Aw, method="nested_dissection")
        assert O.factorize(Aw) == 0
        _, Lo, Do = O.factor()
        O.factorize(Aw)
        to = O.timing()[0]
        O.close()
        assert np.abs(Dw - Do).max() <= 1e-12 * np.abs(Do).max() and abs(Lw - Lo).max() <= 1e-11 * max(1.0, abs(Lo).max())
        print("separator of %d + %d (n = %d): many workgroups per front %.2f ms, one workgroup %.2f ms" % (ns, nq, n, tw, to))
        assert tw < to
        # a dense block of 1500 (24 chained fronts of up to 1500 rows)
        M = rng.standard_normal((1500, 1500)); Kd = M @ M.T + 1500 * np.eye(1500)
        Ad = sp.csc_matrix(np.triu(Kd))
        bd = rng.standard_normal(1500)
        times = {}
        for on in (1, 0):
            _wide_fronts(pkg, on)
            Sd = pkg.SparseLDL(Ad, method="nested_dissection")
            assert Sd.factorize(Ad) == 0
            assert np.abs(Kd @ Sd.solve(bd) - bd).max() <= 1e-8 * np.abs(bd).max() * 10
            Sd.factorize(Ad)
            times[on] = Sd.timing()[0]
            Sd.close()
        print("dense block of 1500: many workgroups per front %.2f ms, one workgroup %.2f ms" % (times[1], times[0]))
        assert times[1] < 0.25 * times[0]
        # beyond 4095 rows (where the one-workgroup kernels and their LDS-resident sweeps end): fronts of up to 4300 rows, the sweeps of the solve by many workgroups
        # too; a quasi-definite matrix (the last 300 pivots negative), three right-hand sides
        _wide_fronts(pkg, 1)
        nb, nq2 = 4000, 300
        M = rng.standard_normal((nb, nb)); H = M @ M.T / nb + 2.0 * np.eye(nb)
        G2 = rng.standard_normal((nq2, nb)) / np.sqrt(nb)
        C2 = rng.standard_normal((nq2, nq2)); C2 = C2 @ C2.T / nq2 + np.eye(nq2)             # (dense: one clique of 4300 vertices — a diagonal block would make 300 leaves of 4001 rows)
        Kq = np.block([[H, G2.T], [G2, -C2]])
        Aq = sp.csc_matrix(np.triu(Kq))
        Sq = pkg.SparseLDL(Aq, method="nested_dissection")
        assert Sq.info["numeric"] == "multifrontal"
        assert Sq.factorize(Aq) == 0 and Sq.inertia == (nb, nq2, 0)
        bq = rng.standard_normal((nb + nq2, 3))
        xq = Sq.solve(bq)
        assert np.abs(Kq @ xq - bq).max() <= 1e-9 * max(1.0, np.abs(xq).max())
        xr = np.linalg.solve(Kq, bq)
        assert np.abs(xq - xr).max() <= 1e-9 * max(1.0, np.abs(xr).max())
        print("quasi-definite block of 4300: factor %.2f ms, solve (3 right-hand sides) %.2f ms" % Sq.timing())
        Sq.close()
    finally:
        _wide_fronts(pkg, was)


def _mf_items(pkg, on):
    """calipso_hip_debug_mf_items: plans made afterwards carry the extend-add items of the LDS fronts (1, the default) or assemble child by child (0); returns the old value"""
    fn = pkg._lib.lib().calipso_hip_debug_mf_items
    fn.restype = C.c_int32
    return fn(C.c_int32(on))


@pytest.mark.parametrize("shape", [(41, 20, 8), (24, 5, 2), (64, 6, 2), (12, 40, 8)])
def test_assembly_through_extend_add_items_has_the_bits_of_the_row_wise_assembly(oracle_mod, shape):
    """round 6: an LDS front is assembled from a list of (pool entry, place in the front) items built with the pattern — every global load of the assembly in
    flight at once — instead of child by child and row by row; the operations on every entry and their order are the same, so L, D and a solve are the SAME
    BITS with and without the items (and equal the oracle's QDLDL to 1e-11), alone and in a batch.  Shapes: C4T-like stages (fronts of 56 / 112 / 168 rows: the
    512-thread kernel, half panels), small stages (256-thread kernel), a deep tree, wide stages."""
    pkg = load_pkg()
    T, ns, nu = shape
    rng = np.random.default_rng(100 + T)
    K = staged_kkt(T, ns, nu, rng)
    A = sp.triu(K).tocsc(); A.sort_indices()
    n = K.shape[0]
    b = rng.standard_normal(n)
    out = {}
    was = _mf_items(pkg, 1)
    try:
        for on in (1, 0):
            _mf_items(pkg, on)
            S = pkg.SparseLDL(A, method="nested_dissection")
            assert S.info["numeric"] == "multifrontal"
            assert S.factorize(A) == 0
            perm, Lm, D = S.factor()
            x = S.solve(b)
            S.set_batch(3)
            vals = np.stack([A.data, A.data * 1.0, A.data])
            assert S.factorize(vals) == 0
            S.select(2)
            _, Lb, Db = S.factor()
            out[on] = (perm, Lm.toarray(), D, x, Lb.toarray(), Db)
            S.close()
    finally:
        _mf_items(pkg, was)
    for k in range(6):
        assert np.array_equal(out[1][k], out[0][k]), k
    assert np.array_equal(out[1][1], out[1][4]) and np.array_equal(out[1][2], out[1][5])          # a batch member has the bits of the matrix alone
    ref = oracle_factor(oracle_mod, K.toarray(), out[1][0])
    assert np.abs(out[1][2] - ref["D"]).max() <= 1e-11 * np.abs(ref["D"]).max()
    assert abs(out[1][1] - ref["L"]).max() <= 1e-11 * max(1.0, abs(ref["L"]).max())
    assert np.abs(K @ out[1][3] - b).max() <= 1e-8 * max(1.0, np.abs(out[1][3]).max())
