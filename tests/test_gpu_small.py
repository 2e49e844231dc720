"""GPU (-m gpu): the batched LDS-resident path for small KKT systems (csrc/small.hip; SURVEY.md 8(f2), BASELINE config C5): one launch
factors `batch` systems and solves `nrhs` right-hand sides each.  Checked against the oracle's factorize!/linear_solve! on the
cart-pole MPC system (n = 89, the 102 condensed right-hand sides of differentiate!, src/solver/differentiate.jl:29-58) and on the
pendulum system (n = 56), tolerances of SURVEY.md 8(c)."""
import numpy as np
import pytest

import problems as pr
from helpers import load_pkg
from test_oracle_solve import run as run_oracle

pytestmark = pytest.mark.gpu

OPTS = dict(residual_tolerance=1e-3, optimality_tolerance=1e-3, equality_tolerance=1e-3, complementarity_tolerance=1e-3,
            slack_tolerance=1e-3, differentiate=1)


def rel(a, b):
    return np.abs(np.asarray(a) - np.asarray(b)).max() / max(1.0, np.abs(np.asarray(b)).max())


def condensed_system(o):
    """K and the inertia of the oracle's last factorisation of it"""
    o.residual_jacobian_variables(); o.residual_jacobian_variables_symmetric()
    K = np.array(o.K_dense())
    o.factorize(update=False)
    return K, o.compute_inertia()


def test_c5_sensitivity_solves_batched(oracle_mod):
    """many MPC steps at once: every instance is the cart-pole system at the oracle's solution with its own right-hand sides"""
    pkg = load_pkg()
    prob = pr.cartpole_mpc()
    o, st = run_oracle(oracle_mod, prob, **OPTS)
    assert st == 1
    K, inertia = condensed_system(o)
    n, p, batch = o.n, prob.np, 24
    assert (n, p) == (89, 102)
    rng = np.random.default_rng(1)
    Ks = np.repeat(K[None], batch, axis=0)
    for b in range(1, batch):                              # instances differ: perturb the (x, x) block, keep it quasi-definite
        Q = rng.standard_normal((prob.nx, prob.nx)) * 0.05
        Ks[b, :prob.nx, :prob.nx] += Q @ Q.T
    Ks_garbage_below = Ks.copy()
    il = np.tril_indices(n, -1)
    Ks_garbage_below[:, il[0], il[1]] = 777.0              # only triu(K) may be read
    Bs = rng.standard_normal((batch, n, p))
    sb = pkg.SmallBatch(n, p, batch)
    sb.set(Ks_garbage_below, Bs)
    ms = sb.solve()
    X, inr, bad = sb.get()
    assert bad == 0 and ms > 0
    assert tuple(inr[0]) == tuple(inertia) == (prob.nx, prob.ne + prob.nc, 0)
    for j in (0, 17, 101):                                 # instance 0 against the oracle's linear_solve! (qdldl.jl:330-351)
        assert rel(X[0][:, j], o.linear_solve(Bs[0][:, j], fact=False)) <= 1e-8
    for b in range(batch):                                 # every instance against its own symmetrised matrix
        Ksym = np.triu(Ks[b]) + np.triu(Ks[b], 1).T
        assert np.abs(Ksym @ X[b] - Bs[b]).max() <= 1e-9 * max(1.0, np.abs(Bs[b]).max())
        assert tuple(inr[b]) == (prob.nx, prob.ne, 0)
    # the same through the general device seam (calipso_hip_ldl_*): two independent device paths agree to rounding
    import scipy.sparse as sp
    ls = pkg.LDLSolver(n)
    ls.factorize(sp.csc_matrix(Ks[3]))
    assert rel(X[3], ls.linear_solve(Bs[3])) <= 1e-9
    # new right-hand sides for the resident matrices
    B2 = rng.standard_normal((batch, n, p))
    sb.set(None, B2)
    sb.solve()
    X2, _, _ = sb.get()
    assert rel(X2[5], np.linalg.solve(np.triu(Ks[5]) + np.triu(Ks[5], 1).T, B2[5])) <= 1e-9
    sb.close()


def test_c2_pendulum_system_and_rhs_chunking(oracle_mod):
    """n = 56 (pendulum, BASELINE config C2) with more right-hand sides than fit beside the matrix in one pass, and n = 128 (the limit)"""
    pkg = load_pkg()
    prob = pr.pendulum(action_guess=np.zeros(10))
    o, st = run_oracle(oracle_mod, prob)
    assert st == 1
    K, inertia = condensed_system(o)
    n = o.n
    assert n == 56
    rng = np.random.default_rng(2)
    for nrhs in (1, 7, 400):
        B = rng.standard_normal((2, n, nrhs))
        sb = pkg.SmallBatch(n, nrhs, 2)
        sb.set(np.stack([K, K]), B)
        sb.solve()
        X, inr, bad = sb.get()
        assert bad == 0 and tuple(inr[1]) == tuple(inertia)
        assert rel(X[1][:, nrhs - 1], o.linear_solve(B[1][:, nrhs - 1], fact=False)) <= 1e-8
        assert np.array_equal(X[0] * 0 + X[1], X[1])      # finite
        sb.close()
    n = 128
    Q = rng.standard_normal((n, n))
    A = Q @ Q.T + n * np.eye(n)
    A[90:, 90:] = -A[90:, 90:]; A[:90, 90:] *= 0.1; A[90:, :90] = A[:90, 90:].T
    B = rng.standard_normal((3, n, 40))
    sb = pkg.SmallBatch(n, 40, 3)
    sb.set(np.stack([A, A, A]), B)
    sb.solve()
    X, inr, bad = sb.get()
    assert bad == 0 and tuple(inr[2]) == (90, 38, 0)
    assert np.abs(A @ X[2] - B[2]).max() <= 1e-9
    with pytest.raises(pkg.CalipsoHipError):
        pkg.SmallBatch(129, 1, 1)


def test_zero_pivot_is_reported_per_instance():
    pkg = load_pkg()
    n = 8
    good = np.diag([2.0, 1.0, 3.0, -1.0, -2.0, 1.0, 1.0, 1.0])
    sing = good.copy(); sing[1, 1] = 0.0
    sb = pkg.SmallBatch(n, 1, 2)
    sb.set(np.stack([good, sing]), np.ones((2, n, 1)))
    sb.solve()
    X, inr, bad = sb.get()
    assert bad == 1 and tuple(inr[0]) == (6, 2, 0) and tuple(inr[1]) == (-1, 7, 7)      # stale-zero tail of D
    assert np.allclose(X[0][:, 0], 1.0 / np.diag(good))
