"""GPU (-m gpu): the scatter of evaluate! on the device (csrc/scatter.hip; src/solver/evaluate.jl:37-121).  The
caller registers methods.<field>_sparsity once and then hands over value caches only; the device reproduces the reference's assignment order
(the last writer of a repeated entry wins) and sums the three Hessian matrices.  Pendulum (BASELINE config C2): consecutive dynamics stages
write the same (X_t+1, X_t+1) Hessian entries."""
import numpy as np
import pytest

import problems as pr
from helpers import load_pkg

pytestmark = pytest.mark.gpu


def trajectory_caches(prob, x, y):
    """what the reference's trajectory layer hands to evaluate!: per-stage value caches + (row, col) lists, concatenated without de-duplication
    (src/trajectory_optimization/methods.jl:24-27, dynamics.jl:181-192,245-260)"""
    fn, pattern = prob._stage_hessian
    rows, cols, vals = [], [], []
    for idx, dsl in zip(prob._stage_slices, prob._stage_dual_slices):
        loc = np.asarray(fn(*x[idx], *y[dsl]), dtype=np.float64).reshape(len(idx), len(idx))
        for (i, j) in pattern:
            rows.append(idx[i] + 1); cols.append(idx[j] + 1); vals.append(loc[i, j])
    return np.array(rows), np.array(cols), np.array(vals)


def dense(prob, x, y, names):
    bufs = {}
    size = dict(objective=1, objective_gradient_variables=prob.nx, equality_constraint=prob.ne, cone_constraint=0, equality_dual_jacobian_variables=prob.nx,
                cone_dual_jacobian_variables=prob.nx, objective_jacobian_variables_variables=prob.nx ** 2,
                equality_dual_jacobian_variables_variables=prob.nx ** 2, cone_dual_jacobian_variables_variables=prob.nx ** 2,
                equality_jacobian_variables=prob.ne * prob.nx, cone_jacobian_variables=0)
    prob.evaluate(pr.ALL_VARIABLE_FLAGS, x, y, np.zeros(0), np.zeros(0), lambda nm: bufs.setdefault(nm, np.zeros(size[nm])))
    return [bufs[n] for n in names]


def test_scatter_reproduces_the_last_writer_wins_hessian_bit_for_bit():
    pkg = load_pkg()
    prob = pr.pendulum(action_guess=np.zeros(10))                 # hessian_mode = "last_writer": the reference's semantics
    exact = pr.pendulum(action_guess=np.zeros(10), hessian_mode="sum")
    rng = np.random.default_rng(0)
    x, y = rng.standard_normal(prob.nx), rng.standard_normal(prob.ne)
    fxx, gyxx, gx = dense(prob, x, y, ["objective_jacobian_variables_variables", "equality_dual_jacobian_variables_variables", "equality_jacobian_variables"])
    gyxx_exact, = dense(exact, x, y, ["equality_dual_jacobian_variables_variables"])
    assert np.abs(gyxx - gyxx_exact).max() > 1e-3                 # the two semantics really differ on this problem
    s = pkg.Solver(prob, prob.nx, 0, prob.ne, 0)
    # objective Hessian 0.2 I (test/examples/pendulum.jl:36-39); dynamics Hessians with repeated entries; the Jacobian by its non-zeros
    dg = np.arange(1, prob.nx + 1)
    s.set_sparsity("objective_jacobian_variables_variables", dg, dg)
    hr, hc, hv = trajectory_caches(prob, x, y)
    assert len(set(zip(hr, hc))) < len(hr)                        # duplicates are present
    s.set_sparsity("equality_dual_jacobian_variables_variables", hr, hc)
    G = gx.reshape(prob.nx, prob.ne).T
    jr, jc = np.nonzero(G)
    s.set_sparsity("equality_jacobian_variables", jr + 1, jc + 1)
    s.scatter_hessian(objective=np.diag(fxx.reshape(prob.nx, prob.nx)).copy(), equality_dual=hv)
    s.scatter_field("equality_jacobian_variables", G[jr, jc])
    L = s.get("lagrangian_hessian", prob.nx ** 2)
    assert np.array_equal(L, fxx + gyxx)                          # = what the dense upload of the host-scattered matrices holds
    assert np.array_equal(s.get("equality_jacobian_variables", prob.ne * prob.nx), gx)
    # a second evaluation at another point re-uses the registered lists
    x2, y2 = rng.standard_normal(prob.nx), rng.standard_normal(prob.ne)
    fxx2, gyxx2 = dense(prob, x2, y2, ["objective_jacobian_variables_variables", "equality_dual_jacobian_variables_variables"])
    s.scatter_hessian(objective=np.diag(fxx2.reshape(prob.nx, prob.nx)).copy(), equality_dual=trajectory_caches(prob, x2, y2)[2])
    assert np.array_equal(s.get("lagrangian_hessian", prob.nx ** 2), fxx2 + gyxx2)
    # a PARTIAL re-evaluation (evaluate.jl:37-42: only the flagged matrices are rewritten, residual_jacobian_variables.jl:10-16 sums all three):
    # the part that is not passed keeps the values of its last scatter
    hv3 = trajectory_caches(prob, x, y2)[2]
    gyxx3, = dense(prob, x, y2, ["equality_dual_jacobian_variables_variables"])
    s.scatter_hessian(equality_dual=hv3)                           # e.g. after a dual update: the objective part is NOT re-evaluated
    assert np.array_equal(s.get("lagrangian_hessian", prob.nx ** 2), fxx2 + gyxx3)
    s.scatter_hessian(objective=np.diag(fxx.reshape(prob.nx, prob.nx)).copy())
    assert np.array_equal(s.get("lagrangian_hessian", prob.nx ** 2), fxx + gyxx3)
    # constraint_tensor = false: the tensor parts are never evaluated, so they contribute nothing (a part never scattered is zero; an
    # earlier one is dropped by un-registering its list)
    s.set_sparsity("equality_dual_jacobian_variables_variables", [], [])
    s.scatter_hessian(objective=np.diag(fxx2.reshape(prob.nx, prob.nx)).copy())
    assert np.array_equal(s.get("lagrangian_hessian", prob.nx ** 2), fxx2)
    s.set_sparsity("equality_dual_jacobian_variables_variables", hr, hc)
    with pytest.raises(pkg.CalipsoHipError):
        s.scatter_hessian(equality_dual=hv[:-1])                  # cache length must match the registered list
    with pytest.raises(pkg.CalipsoHipError):
        s.set_sparsity("equality_jacobian_variables", [prob.ne + 1], [1])


def test_solve_through_the_sparse_scatter_equals_the_dense_upload():
    """solve! with the evaluation callback handing over value caches (scatter on the device) walks the same iterates as with dense uploads"""
    pkg = load_pkg()
    prob = pr.pendulum(action_guess=np.zeros(10))

    class SparseSolver(pkg.Solver):
        def upload(self, flags):
            F = pkg.FLAGS
            hess = F["objective_jacobian_variables_variables"] | F["equality_dual_jacobian_variables_variables"] | F["cone_dual_jacobian_variables_variables"]
            jac = F["equality_jacobian_variables"]
            pkg.Solver.upload(self, flags & ~(hess | jac))        # vectors and scalars as usual
            w = self.get("solution" if self._which == 0 else "candidate", self.N)
            x, y = w[:self.nx], w[self.nx + self.ne + self.nc: self.nx + 2 * self.ne + self.nc]
            if flags & hess:
                fxx = self.problem["objective_jacobian_variables_variables"].reshape(self.nx, self.nx)
                self.scatter_hessian(objective=np.diag(fxx).copy(), equality_dual=trajectory_caches(prob, x, y)[2] if flags & F["equality_dual_jacobian_variables_variables"] else None)
            if flags & jac:
                G = self.problem["equality_jacobian_variables"].reshape(self.nx, self.ne).T
                self.scatter_field("equality_jacobian_variables", G[self._jr, self._jc])

        def _evaluate_callback(self, user, flags, px, py, pz, pth):
            self._which = 0 if self._next_which is None else self._next_which
            return pkg.Solver._evaluate_callback(self, user, flags, px, py, pz, pth)

    dense_s = pkg.Solver(prob, prob.nx, 0, prob.ne, 0)
    pkg.initialize_b(dense_s, prob.x0)
    assert pkg.solve_b(dense_s)
    sp_s = SparseSolver(prob, prob.nx, 0, prob.ne, 0)
    sp_s._next_which = None
    # register the lists once (pattern of the equality Jacobian: structural non-zeros at a generic point)
    rng = np.random.default_rng(1)
    xg, yg = rng.standard_normal(prob.nx), rng.standard_normal(prob.ne)
    gx, = dense(prob, xg, yg, ["equality_jacobian_variables"])
    G = gx.reshape(prob.nx, prob.ne).T
    sp_s._jr, sp_s._jc = np.nonzero(G)
    dg = np.arange(1, prob.nx + 1)
    sp_s.set_sparsity("objective_jacobian_variables_variables", dg, dg)
    hr, hc, _ = trajectory_caches(prob, xg, yg)
    sp_s.set_sparsity("equality_dual_jacobian_variables_variables", hr, hc)
    sp_s.set_sparsity("equality_jacobian_variables", sp_s._jr + 1, sp_s._jc + 1)
    # the callback is always asked for Hessians / Jacobians at the SOLUTION point (solve.jl:175-181), line-search re-evaluations only need f, g, h
    pkg.initialize_b(sp_s, prob.x0)
    assert pkg.solve_b(sp_s)
    assert sp_s.stats()["total_iterations"] == dense_s.stats()["total_iterations"]
    assert np.array_equal(sp_s.solution.all, dense_s.solution.all)
    # the same solve on a STRUCTURED handle (calipso_hip_create_structured): the structure declared from the lists, the caches scattered straight into the
    # packed blocks (no dense Lxx / [gx; hx] / S on the device); the block kernels sum in another order than the dense ones, so: same iteration count, same
    # solution to rounding
    first = np.ones(prob.ne, dtype=np.int64); last = np.zeros(prob.ne, dtype=np.int64)
    for k in range(prob.ne):
        cols = sp_s._jc[sp_s._jr == k]
        if cols.size:
            first[k], last[k] = cols.min() + 1, cols.max() + 1
    reach = np.arange(prob.nx)
    for i, j in list(zip(dg - 1, dg - 1)) + list(zip(hr - 1, hc - 1)):
        lo, hi = min(i, j), max(i, j)
        reach[lo] = max(reach[lo], hi)
    starts, r = [1], -1
    for j in range(prob.nx):
        if j > starts[-1] - 1 and r < j:
            starts.append(j + 1)
        r = max(r, reach[j])
    assert len(starts) >= 2
    st_s = SparseSolver(prob, prob.nx, 0, prob.ne, 0, structure=dict(row_first=first, row_last=last, hessian_block_start=np.array(starts)))
    st_s._next_which = None
    st_s._jr, st_s._jc = sp_s._jr, sp_s._jc
    st_s.set_sparsity("objective_jacobian_variables_variables", dg, dg)
    st_s.set_sparsity("equality_dual_jacobian_variables_variables", hr, hc)
    st_s.set_sparsity("equality_jacobian_variables", st_s._jr + 1, st_s._jc + 1)
    assert st_s.device_bytes() < dense_s.device_bytes()
    pkg.initialize_b(st_s, prob.x0)
    assert pkg.solve_b(st_s)
    assert st_s.stats()["total_iterations"] == dense_s.stats()["total_iterations"]
    assert np.abs(st_s.solution.all - dense_s.solution.all).max() <= 1e-7 * max(1.0, np.abs(dense_s.solution.all).max())
    with pytest.raises(pkg.CalipsoHipError, match="outside the structure"):
        st_s.set_sparsity("equality_jacobian_variables", [1], [prob.nx])          # row 1 does not reach the last column
