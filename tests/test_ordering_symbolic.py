"""The ordering / symbolic service of the product (csrc/ordering.hip) — host-side integer work, so these run WITHOUT a GPU:
  * calipso_hip_symbolic (permute_symmetric + QDLDL_etree!, qdldl.jl:358-395,642-742) is BIT-EXACT against the oracle's restatement
    for the same permutation: Pp, Pi (incl. the unsorted placement order inside columns), AtoPAPt, etree, Lnz, nnz(L);
  * calipso_hip_ordering returns permutations; reverse Cuthill-McKee recovers the band of a shuffled banded matrix, minimum degree does
    not produce more fill than the natural order on arrow / grid patterns (the property a fill-reducing order is for)."""
import ctypes as C

import numpy as np
import pytest
import scipy.sparse as sp

import problems as pr
from helpers import load_pkg
from test_oracle_qdldl import quasidefinite, reference_etree


def _pi(a):
    return a.ctypes.data_as(C.POINTER(C.c_int64))


def _pd(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def csc1(A):
    A = sp.csc_matrix(A); A.sort_indices()
    return A.indptr.astype(np.int64) + 1, A.indices.astype(np.int64) + 1, A.data.astype(np.float64)


def oracle_symbolic(oracle_mod, K, perm):
    """the oracle's permute_symmetric + etree on triu(K) (as qdldl(A) does: triu first, qdldl.jl:145-147)"""
    L = oracle_mod.lib()
    n = K.shape[0]
    Ap, Ai, Ax = csc1(sp.triu(sp.csc_matrix(K)))
    nnz = len(Ai)
    iperm = np.zeros(n, dtype=np.int64); iperm[perm - 1] = np.arange(1, n + 1)
    Pp, Pi, Px, mp = np.zeros(n + 1, dtype=np.int64), np.zeros(nnz, dtype=np.int64), np.zeros(nnz), np.zeros(nnz, dtype=np.int64)
    L.oracle_qdldl_permute_symmetric(n, _pi(Ap), _pi(Ai), _pd(Ax), _pi(iperm), _pi(Pp), _pi(Pi), _pd(Px), _pi(mp))
    work, Lnz, et = np.zeros(n, dtype=np.int64), np.zeros(n, dtype=np.int64), np.zeros(n, dtype=np.int64)
    tot = L.oracle_qdldl_etree(n, _pi(Pp), _pi(Pi), _pi(work), _pi(Lnz), _pi(et))
    return dict(Pp=Pp, Pi=Pi, AtoPAPt=mp, etree=et, Lnz=Lnz, nnzL=int(tot))


def kkt_matrices():
    rng = np.random.default_rng(0)
    out = {"quasidefinite_15": quasidefinite(9, 6, rng)}
    # the condensed K pattern of a trajectory problem (pendulum, BASELINE config C2) and of a mixed conic QP
    for name, prob in (("pendulum", pr.pendulum(action_guess=np.zeros(10))), ("conic_qp", pr.random_qp(12, 5, 7, seed=3, nonnegative_indices=[1, 2, 3],
                                                                                                    second_order_indices=[[4, 5, 6, 7]]))):
        nx, ne, nc = prob.nx, prob.ne, prob.nc
        bufs = {}
        size = dict(objective=1, objective_gradient_variables=nx, equality_constraint=ne, cone_constraint=nc, equality_dual_jacobian_variables=nx,
                    cone_dual_jacobian_variables=nx, objective_jacobian_variables_variables=nx * nx, equality_dual_jacobian_variables_variables=nx * nx,
                    cone_dual_jacobian_variables_variables=nx * nx, equality_jacobian_variables=ne * nx, cone_jacobian_variables=nc * nx)
        out_f = lambda nm: bufs.setdefault(nm, np.zeros(size[nm]))
        x = rng.standard_normal(nx); y = rng.standard_normal(ne); z = rng.standard_normal(nc)
        prob.evaluate(pr.ALL_VARIABLE_FLAGS, x, y, z, np.zeros(0), out_f)
        H = sum(bufs[k].reshape(nx, nx).T for k in ("objective_jacobian_variables_variables", "equality_dual_jacobian_variables_variables",
                                                    "cone_dual_jacobian_variables_variables"))
        gx = bufs["equality_jacobian_variables"].reshape(nx, ne).T
        hx = bufs["cone_jacobian_variables"].reshape(nx, nc).T if nc else np.zeros((0, nx))
        K = np.block([[H + 0.1 * np.eye(nx), gx.T, hx.T], [gx, -0.5 * np.eye(ne), np.zeros((ne, nc))], [hx, np.zeros((nc, ne)), -0.7 * np.eye(nc)]])
        if nc >= 7:
            K[nx + ne + 3:nx + ne + 7, nx + ne + 3:nx + ne + 7] -= 0.1          # the dense second-order block of K_zz
        out[name] = K
    return out


@pytest.mark.parametrize("name", ["quasidefinite_15", "pendulum", "conic_qp"])
def test_symbolic_matches_the_oracle_bit_exact(oracle_mod, name):
    pkg = load_pkg()
    K = kkt_matrices()[name]
    n = K.shape[0]
    rng = np.random.default_rng(5)
    perms = [np.arange(1, n + 1), np.arange(n, 0, -1), rng.permutation(n) + 1, pkg.ordering(K, "rcm"), pkg.ordering(K, "minimum_degree")]
    for perm in perms:
        perm = np.asarray(perm, dtype=np.int64)
        ref = oracle_symbolic(oracle_mod, K, perm)
        got = pkg.symbolic(sp.triu(sp.csc_matrix(K)), perm)
        for key in ("Pp", "Pi", "AtoPAPt", "etree", "Lnz"):
            assert np.array_equal(got[key], ref[key]), (name, key)
        assert got["nnzL"] == ref["nnzL"]
        # both triangles supplied: entries below the diagonal are ignored (AtoPAPt = 0 there), the rest is unchanged
        full = pkg.symbolic(sp.csc_matrix(K), perm)
        assert np.array_equal(full["etree"], ref["etree"]) and np.array_equal(full["Lnz"], ref["Lnz"]) and full["nnzL"] == ref["nnzL"]
    # independent textbook construction of the elimination tree for the natural order
    Ap, Ai, _ = csc1(sp.triu(sp.csc_matrix(K)))
    parent, counts = reference_etree(n, Ap, Ai)
    nat = pkg.symbolic(sp.triu(sp.csc_matrix(K)))
    assert [int(v) for v in nat["etree"]] == parent and [int(v) for v in nat["Lnz"]] == counts


def test_orderings_are_permutations_and_do_their_job():
    pkg = load_pkg()
    rng = np.random.default_rng(1)
    # a banded matrix, symmetrically shuffled: RCM finds an order with (about) the original bandwidth
    n, hb = 120, 4
    B = sp.diags([np.ones(n - abs(k)) for k in range(-hb, hb + 1)], list(range(-hb, hb + 1))).toarray()
    q = rng.permutation(n)
    A = B[np.ix_(q, q)]
    assert pkg.symbolic(A)["half_bandwidth"] > 60
    p = pkg.ordering(A, "rcm")
    assert sorted(p) == list(range(1, n + 1))
    assert pkg.symbolic(A, p)["half_bandwidth"] <= 2 * hb
    # an arrow matrix (dense first row / column): eliminating the hub last avoids all fill; a 2-D grid: minimum degree beats the natural order
    m = 40
    arrow = np.eye(m); arrow[0, :] = 1.0; arrow[:, 0] = 1.0
    pm = pkg.ordering(arrow, "minimum_degree")
    assert sorted(pm) == list(range(1, m + 1))
    assert pkg.symbolic(arrow)["nnzL"] == m * (m - 1) // 2 and pkg.symbolic(arrow, pm)["nnzL"] == m - 1
    g = 12
    G = sp.kron(sp.identity(g), sp.diags([1, 1, 1], [-1, 0, 1], shape=(g, g))) + sp.kron(sp.diags([1, 1], [-1, 1], shape=(g, g)), sp.identity(g))
    pg = pkg.ordering(G, "minimum_degree")
    assert sorted(pg) == list(range(1, g * g + 1))
    assert pkg.symbolic(G, pg)["nnzL"] < 0.8 * pkg.symbolic(G)["nnzL"]
    assert list(pkg.ordering(G, "natural")) == list(range(1, g * g + 1))
    # the trajectory-structured K of the pendulum: natural order [x | y] couples the first and last block, RCM interleaves them into a band
    K = kkt_matrices()["pendulum"]
    assert pkg.symbolic(K, pkg.ordering(K, "rcm"))["half_bandwidth"] < 0.5 * pkg.symbolic(K)["half_bandwidth"]


def staged_kkt(T, ns, nu, rng, delta=1e-2):
    """KKT matrix [H G'; G -delta I] of a T-stage trajectory problem in the reference's variable order (trajectory_optimization/indices.jl:41-180):
    variables (x_1,u_1,...,x_T,u_T,x_{T+1}), H block-diagonal per stage, dynamics rows x_{t+1} = A_t x_t + B_t u_t coupling neighbouring stages"""
    nv = T * (ns + nu) + ns
    H = sp.lil_matrix((nv, nv))
    for t in range(T + 1):
        w = ns + nu if t < T else ns
        M = rng.standard_normal((w, w))
        o = t * (ns + nu)
        H[o:o + w, o:o + w] = M @ M.T + w * np.eye(w)
    G = sp.lil_matrix((T * ns, nv))
    for t in range(T):
        o = t * (ns + nu)
        G[t * ns:(t + 1) * ns, o:o + ns + nu] = rng.standard_normal((ns, ns + nu))
        G[t * ns:(t + 1) * ns, o + ns + nu:o + ns + nu + ns] = -np.eye(ns)
    K = sp.bmat([[H, G.T], [G, -delta * sp.identity(T * ns)]], format="csc")
    K.sort_indices()
    return K


def tree_height(etree):
    """levels of the elimination tree (1-based parents, -1 = root)"""
    n = len(etree)
    lev = np.zeros(n, dtype=np.int64)
    for j in range(n):
        p = int(etree[j])
        if p > 0:
            lev[p - 1] = max(lev[p - 1], lev[j] + 1)
    return int(lev.max()) + 1


def test_nested_dissection_is_a_permutation_and_flattens_the_tree():
    pkg = load_pkg()
    rng = np.random.default_rng(4)
    for T, ns, nu, ratio in ((16, 3, 1, 0.5), (64, 6, 2, 0.15), (256, 6, 2, 0.05)):
        K = staged_kkt(T, ns, nu, rng)
        n = K.shape[0]
        perm = pkg.ordering(K, "nested_dissection")
        assert sorted(perm.tolist()) == list(range(1, n + 1))
        h_nd = tree_height(pkg.symbolic(sp.triu(K).tocsc(), perm)["etree"])
        h_nat = tree_height(pkg.symbolic(sp.triu(K).tocsc(), None)["etree"])
        h_md = tree_height(pkg.symbolic(sp.triu(K).tocsc(), pkg.ordering(K, "minimum_degree"))["etree"])
        assert h_nd < ratio * h_nat and h_nd < ratio * h_md, (h_nd, h_nat, h_md)       # O(log T) separators instead of a chain through the horizon
    # pieces the level structure cannot split (a clique, disconnected vertices) still give a valid order
    for M in (np.ones((70, 70)), np.eye(100), sp.block_diag([np.ones((60, 60)), np.eye(30)]).toarray()):
        perm = pkg.ordering(sp.csc_matrix(M), "nested_dissection")
        assert sorted(perm.tolist()) == list(range(1, M.shape[0] + 1))


def test_malformed_csc_patterns_are_rejected_before_any_indexing():
    """calipso_hip_ordering / calipso_hip_symbolic walk the caller's colptr / rowval: a row index outside 1..n, a colptr that does not start at 1
    or decreases must come back as CALIPSO_ERR_ARGUMENT (-4), not index past the arrays (host functions: no device needed)"""
    import ctypes as C
    pkg = load_pkg()
    from calipso_jl_amd._lib import lib
    L = lib()
    pi = lambda a: a.ctypes.data_as(C.POINTER(C.c_int64))
    n = 4
    good_ptr = np.array([1, 2, 4, 6, 8], dtype=np.int64)
    good_row = np.array([1, 1, 2, 2, 3, 3, 4], dtype=np.int64)
    perm = np.zeros(n, dtype=np.int64)
    info = np.zeros(2, dtype=np.int64)
    assert L.calipso_hip_ordering(n, pi(good_ptr), pi(good_row), 1, pi(perm)) == 0 and sorted(perm.tolist()) == [1, 2, 3, 4]
    assert L.calipso_hip_symbolic(n, pi(good_ptr), pi(good_row), None, None, None, None, None, None, pi(info)) >= 0
    for bad_ptr, bad_row in (
            (good_ptr, np.array([1, 0, 2, 2, 3, 3, 4], dtype=np.int64)),        # row 0
            (good_ptr, np.array([1, 1, 2, 2, 3, 9, 4], dtype=np.int64)),        # row > n
            (good_ptr, np.array([1, -3, 2, 2, 3, 3, 4], dtype=np.int64)),       # negative row
            (np.array([0, 1, 3, 5, 7], dtype=np.int64), good_row),              # 0-based colptr
            (np.array([1, 4, 2, 6, 8], dtype=np.int64), good_row)):             # decreasing colptr
        assert L.calipso_hip_ordering(n, pi(bad_ptr), pi(bad_row), 2, pi(perm)) == -4
        assert L.calipso_hip_symbolic(n, pi(bad_ptr), pi(bad_row), None, None, None, None, None, None, pi(info)) == -4
