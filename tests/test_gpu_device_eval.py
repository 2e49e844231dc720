"""GPU (-m gpu): user evaluation on the DEVICE (include/calipso_hip.h: calipso_device_eval_fn).  The fixture
tests/device_eval/libuser_device_eval.so plays the user: its evaluators enqueue kernels on the solver's stream and write f, g, h and
their derivatives straight into the solver's device buffers, so that solve! — including every backtracking re-evaluation of the residual
line search (solve.jl:254-302) — never downloads the point, never uploads a block and never calls back into the host language."""
import ctypes as C
import os

import numpy as np
import pytest

import problems as pr
from helpers import ROOT, load_pkg

pytestmark = pytest.mark.gpu


def user_lib():
    L = C.CDLL(os.path.join(ROOT, "tests", "device_eval", "libuser_device_eval.so"))
    L.qp_user_create.restype = C.c_void_p
    L.qp_user_create.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_double] + [C.POINTER(C.c_double)] * 6
    L.qp_user_destroy.argtypes = [C.c_void_p]
    return L


def fnptr(f):
    return C.cast(f, C.c_void_p)


class CountingProblem:
    """wraps a problem and counts host evaluations"""

    def __init__(self, prob):
        self.prob, self.calls = prob, 0

    def evaluate(self, *a):
        self.calls += 1
        return self.prob.evaluate(*a)


def test_wachter_solved_without_a_single_host_evaluation(oracle_mod):
    pkg, UL = load_pkg(), user_lib()
    prob = pr.wachter()
    host = pkg.Solver(prob, 3, 0, 2, 2)
    pkg.initialize_b(host, prob.x0)
    assert pkg.solve_b(host)
    counted = CountingProblem(prob)
    dev = pkg.Solver(counted, 3, 0, 2, 2)
    dev.set_device_evaluator(fnptr(UL.wachter_device_eval))
    pkg.initialize_b(dev, prob.x0)
    assert pkg.solve_b(dev)
    assert counted.calls == 0                                              # the loop never left the device
    assert dev.stats()["total_iterations"] == host.stats()["total_iterations"]
    assert np.abs(dev.solution.all - host.solution.all).max() <= 1e-9
    assert np.abs(dev.solution.variables - np.array([1.0, 0.0, 0.5])).max() <= 1e-3       # test/solver/wachter.jl:47
    # and the oracle agrees
    o = oracle_mod.OracleSolver(3, 0, 2, 2)
    o.point()["x"][:] = prob.x0
    assert o.solve(prob) == 1
    assert np.abs(dev.solution.all - o.point()["all"]).max() <= 1e-8


def make_qp_user(UL, prob):
    f = lambda M: np.ascontiguousarray(np.asarray(M, dtype=np.float64)).reshape(-1)          # row-major, as the fixture expects
    pd = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    arrs = [f(prob.P), f(prob.q), f(prob.A), f(prob.b), f(prob.G), f(prob.h)]
    return UL.qp_user_create(prob.nx, prob.ne, prob.nc, prob.c, *[pd(a) for a in arrs])


def test_user_qp_evaluator_fields_and_solve(oracle_mod):
    pkg, UL = load_pkg(), user_lib()
    prob = pr.random_qp(40, 10, 12, seed=4, nonnegative_indices=[1, 2, 3, 4], second_order_indices=[[5, 6, 7, 8], [9, 10, 11, 12]])
    user = make_qp_user(UL, prob)
    counted = CountingProblem(prob)
    dev = pkg.Solver(counted, prob.nx, 0, prob.ne, prob.nc, nonnegative_indices=prob.nonnegative_indices, second_order_indices=prob.second_order_indices)
    dev.set_device_evaluator(fnptr(UL.qp_device_eval), user)
    host = pkg.Solver(prob, prob.nx, 0, prob.ne, prob.nc, nonnegative_indices=prob.nonnegative_indices, second_order_indices=prob.second_order_indices)
    rng = np.random.default_rng(0)
    w = rng.standard_normal(dev.N)
    for s in (dev, host):
        s.set("solution", w)
    dev.device_evaluate(pr.ALL_VARIABLE_FLAGS, 0)
    host.evaluate(pr.ALL_VARIABLE_FLAGS, 0)
    for name, ln in (("objective", 1), ("objective_gradient_variables", prob.nx), ("equality_constraint", prob.ne), ("cone_constraint", prob.nc),
                     ("equality_dual_jacobian_variables", prob.nx), ("cone_dual_jacobian_variables", prob.nx), ("lagrangian_hessian", prob.nx ** 2),
                     ("equality_jacobian_variables", prob.ne * prob.nx), ("cone_jacobian_variables", prob.nc * prob.nx)):
        a, b = dev.get(name, ln), host.get(name, ln)
        assert np.abs(a - b).max() <= 1e-12 * max(1.0, np.abs(b).max()), name
    assert counted.calls == 0
    # full solves: device-evaluated vs host-callback vs oracle
    x0 = np.zeros(prob.nx)
    for s in (dev, host):
        pkg.initialize_b(s, x0)
    assert pkg.solve_b(dev) and pkg.solve_b(host)
    assert counted.calls == 0
    assert dev.stats()["total_iterations"] == host.stats()["total_iterations"]
    assert np.abs(dev.solution.all - host.solution.all).max() <= 1e-8 * max(1.0, np.abs(host.solution.all).max())
    o = oracle_mod.OracleSolver(prob.nx, 0, prob.ne, prob.nc, prob.nonnegative_indices, prob.second_order_indices)
    o.point()["x"][:] = x0
    assert o.solve(prob) == 1
    assert np.abs(dev.solution.variables - o.point()["x"]).max() <= 1e-6
    UL.qp_user_destroy(user)


def test_group_members_with_user_device_evaluators():
    """a group whose members are evaluated by user kernels on the group's stream: same results as solving the members one by one"""
    pkg, UL = load_pkg(), user_lib()
    probs = [pr.random_qp(30, 8, 6, seed=10 + k) for k in range(3)]
    users = [make_qp_user(UL, p) for p in probs]

    def make(k):
        s = pkg.Solver(CountingProblem(probs[k]), probs[k].nx, 0, probs[k].ne, probs[k].nc)
        s.set_device_evaluator(fnptr(UL.qp_device_eval), users[k])
        pkg.initialize_b(s, np.zeros(probs[k].nx))
        return s

    singles = [make(k) for k in range(3)]
    for s in singles:
        assert pkg.solve_b(s)
    members = [make(k) for k in range(3)]
    g = pkg.Group(members)
    assert g.solve() == [1, 1, 1]
    for s, m in zip(singles, members):
        assert m.methods.calls == 0
        assert m.stats()["total_iterations"] == s.stats()["total_iterations"]
        assert np.array_equal(m.solution.all, s.solution.all)
    g.close()
    for u in users:
        UL.qp_user_destroy(u)


def test_device_evaluator_on_a_structured_handle():
    """a STRUCTURED handle (calipso_hip_create_structured) holds no dense Lxx / [gx; hx]: the user's kernels write the dense ProblemData layout into scratch
    arrays of the handle and the entries go into the blocks behind them — same fields, same solve as the dense handle with the same evaluator; an evaluator
    that writes outside the declared structure is refused."""
    pkg, UL = load_pkg(), user_lib()
    prob, pt, lam = pr.staged_conic_qp(pkg.splitmix_uniform, 9, 6, 8, 4, 3, 1, 3)
    user = make_qp_user(UL, prob)
    kw = dict(nonnegative_indices=prob.nonnegative_indices, second_order_indices=prob.second_order_indices)
    counted = CountingProblem(prob)
    st = pkg.Solver(counted, prob.nx, 0, prob.ne, prob.nc, structure=pr.declared_structure(prob), **kw)
    st.set_device_evaluator(fnptr(UL.qp_device_eval), user)
    dense = pkg.Solver(prob, prob.nx, 0, prob.ne, prob.nc, **kw)
    dense.set_device_evaluator(fnptr(UL.qp_device_eval), user)
    w = np.random.default_rng(1).standard_normal(st.N)
    for s in (st, dense):
        s.set("solution", w)
        s.device_evaluate(pr.ALL_VARIABLE_FLAGS, 0)
    for name, ln in (("objective", 1), ("objective_gradient_variables", prob.nx), ("equality_constraint", prob.ne), ("cone_constraint", prob.nc),
                     ("equality_dual_jacobian_variables", prob.nx), ("cone_dual_jacobian_variables", prob.nx), ("lagrangian_hessian", prob.nx ** 2),
                     ("equality_jacobian_variables", prob.ne * prob.nx), ("cone_jacobian_variables", prob.nc * prob.nx)):
        assert np.array_equal(st.get(name, ln), dense.get(name, ln)), name
    x0 = np.zeros(prob.nx)
    for s in (st, dense):
        pkg.initialize_b(s, x0)
    assert pkg.solve_b(st) and pkg.solve_b(dense)
    assert counted.calls == 0
    assert st.stats()["total_iterations"] == dense.stats()["total_iterations"]
    assert np.abs(st.solution.all - dense.solution.all).max() <= 1e-8 * max(1.0, np.abs(dense.solution.all).max())
    UL.qp_user_destroy(user)
    # the same structure, but the user's A couples the first and the last stage
    import copy
    bad = copy.copy(prob)
    bad.A = prob.A.copy()
    bad.A[0, prob.nx - 1] = 0.7
    user2 = make_qp_user(UL, bad)
    st2 = pkg.Solver(bad, prob.nx, 0, prob.ne, prob.nc, structure=pr.declared_structure(prob), **kw)
    st2.set_device_evaluator(fnptr(UL.qp_device_eval), user2)
    st2.set("solution", w)
    with pytest.raises(pkg.CalipsoHipError, match="outside the declared structure"):
        st2.device_evaluate(pr.ALL_VARIABLE_FLAGS, 0)
    UL.qp_user_destroy(user2)


def test_block_evaluator_writes_the_blocks_of_a_structured_handle_without_dense_scratch():
    """calipso_device_block_eval_fn: the user's kernels write the packed Jacobian / Hessian blocks of a structured handle directly (what the reference's generated functions
    do through their sparsity lists, evaluate.jl:37-121): the same fields and the same solve as the dense-layout evaluator on the same handle type, and no nx^2 + m nx
    scratch on the device"""
    pkg, UL = load_pkg(), user_lib()
    prob, pt, lam = pr.staged_conic_qp(pkg.splitmix_uniform, 9, 6, 8, 4, 3, 1, 3)
    user = make_qp_user(UL, prob)
    kw = dict(nonnegative_indices=prob.nonnegative_indices, second_order_indices=prob.second_order_indices)
    counted = CountingProblem(prob)
    blk = pkg.Solver(counted, prob.nx, 0, prob.ne, prob.nc, structure=pr.declared_structure(prob), **kw)
    blk.set_device_block_evaluator(fnptr(UL.qp_block_device_eval), user)
    scr = pkg.Solver(prob, prob.nx, 0, prob.ne, prob.nc, structure=pr.declared_structure(prob), **kw)
    scr.set_device_evaluator(fnptr(UL.qp_device_eval), user)
    bytes_before = blk.device_bytes()
    w = np.random.default_rng(1).standard_normal(blk.N)
    for s in (blk, scr):
        s.set("solution", w)
        s.device_evaluate(pr.ALL_VARIABLE_FLAGS, 0)
    for name, ln in (("objective", 1), ("objective_gradient_variables", prob.nx), ("equality_constraint", prob.ne), ("cone_constraint", prob.nc),
                     ("equality_dual_jacobian_variables", prob.nx), ("cone_dual_jacobian_variables", prob.nx), ("lagrangian_hessian", prob.nx ** 2),
                     ("equality_jacobian_variables", prob.ne * prob.nx), ("cone_jacobian_variables", prob.nc * prob.nx)):
        assert np.array_equal(blk.get(name, ln), scr.get(name, ln)), name
    assert blk.device_bytes() == bytes_before                                             # nothing was allocated for the evaluation ...
    assert scr.device_bytes() >= bytes_before + 8 * (prob.nx ** 2 + (prob.ne + prob.nc) * prob.nx)   # ... where the dense-layout evaluator costs the handle its dense scratch
    # only one Jacobian asked for: the other one's blocks keep their values
    blk.device_evaluate(pr.EQUALITY_JACOBIAN, 0)
    assert np.array_equal(blk.get("cone_jacobian_variables", prob.nc * prob.nx), scr.get("cone_jacobian_variables", prob.nc * prob.nx))
    x0 = np.zeros(prob.nx)
    for s in (blk, scr):
        pkg.initialize_b(s, x0)
    assert pkg.solve_b(blk) and pkg.solve_b(scr)
    assert counted.calls == 0
    assert blk.stats()["total_iterations"] == scr.stats()["total_iterations"]
    assert np.array_equal(blk.solution.all, scr.solution.all)                             # the same blocks, the same launches
    # a dense handle has no blocks to write
    dense = pkg.Solver(prob, prob.nx, 0, prob.ne, prob.nc, **kw)
    with pytest.raises(Exception):
        dense.set_device_block_evaluator(fnptr(UL.qp_block_device_eval), user)
    UL.qp_user_destroy(user)
