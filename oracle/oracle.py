"""ctypes binding of the CPU oracle (oracle/liboracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg.  The product package never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

EVAL_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint32, C.POINTER(C.c_double), C.POINTER(C.c_double),
                      C.POINTER(C.c_double), C.POINTER(C.c_double))

# evaluate! flags (calipso_oracle.h)
OBJECTIVE = 1 << 0
OBJECTIVE_GRADIENT = 1 << 1
OBJECTIVE_HESSIAN = 1 << 2
EQUALITY = 1 << 3
EQUALITY_JACOBIAN = 1 << 4
EQUALITY_DUAL_GRADIENT = 1 << 5
EQUALITY_DUAL_HESSIAN = 1 << 6
CONE = 1 << 7
CONE_JACOBIAN = 1 << 8
CONE_DUAL_GRADIENT = 1 << 9
CONE_DUAL_HESSIAN = 1 << 10
OBJECTIVE_JACOBIAN_PARAMETERS = 1 << 11
EQUALITY_JACOBIAN_PARAMETERS = 1 << 12
EQUALITY_DUAL_JACOBIAN_PARAMETERS = 1 << 13
CONE_JACOBIAN_PARAMETERS = 1 << 14
CONE_DUAL_JACOBIAN_PARAMETERS = 1 << 15


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "calipso_oracle.cpp")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(so):
            build()
        L = C.CDLL(so)
        i64, dbl, vp = C.c_int64, C.c_double, C.c_void_p
        pi64, pd = C.POINTER(C.c_int64), C.POINTER(C.c_double)
        L.oracle_create.restype = vp
        L.oracle_create.argtypes = [i64, i64, i64, i64, i64, pi64, i64, pi64, pi64]
        L.oracle_destroy.argtypes = [vp]
        L.oracle_buffer.restype = pd
        L.oracle_buffer.argtypes = [vp, C.c_char_p, pi64]
        L.oracle_index.restype = pi64
        L.oracle_index.argtypes = [vp, C.c_char_p, pi64]
        L.oracle_int.restype = pi64
        L.oracle_int.argtypes = [vp, C.c_char_p]
        L.oracle_cone.argtypes = [vp] + [C.c_int] * 6
        L.oracle_cone_violation.argtypes = [vp, pd, pd, dbl]
        for f in ("oracle_residual", "oracle_residual_jacobian_variables", "oracle_residual_jacobian_variables_symmetric",
                  "oracle_merit_gradient", "oracle_filter_reset"):
            getattr(L, f).argtypes = [vp]
            getattr(L, f).restype = None
        L.oracle_H_dense.argtypes = [vp, pd]
        L.oracle_H_mul.argtypes = [vp, pd, pd]
        L.oracle_residual_symmetric.argtypes = [vp, C.c_int]
        L.oracle_factorize.restype = i64
        L.oracle_factorize.argtypes = [vp, C.c_int]
        L.oracle_compute_inertia.argtypes = [vp, pi64]
        L.oracle_linear_solve.argtypes = [vp, pd, pd, C.c_int, C.c_int]
        L.oracle_search_direction_symmetric.argtypes = [vp, C.c_int, C.c_int]
        L.oracle_iterative_refinement.argtypes = [vp]
        L.oracle_inertia_correction.argtypes = [vp]
        L.oracle_search_direction.argtypes = [vp]
        L.oracle_merit.restype = dbl
        L.oracle_merit.argtypes = [vp, dbl, pd, dbl]
        L.oracle_constraint_violation.restype = dbl
        L.oracle_constraint_violation.argtypes = [vp, pd, pd, pd, pd]
        L.oracle_optimality_error.restype = dbl
        L.oracle_optimality_error.argtypes = [vp]
        L.oracle_check_filter.argtypes = [vp, dbl, dbl]
        L.oracle_augment_filter.argtypes = [vp, dbl, dbl]
        L.oracle_augment_filter.restype = None
        L.oracle_filter_pairs.restype = i64
        L.oracle_filter_pairs.argtypes = [vp, pd]
        L.oracle_switching_condition.argtypes = [dbl, pd, pd, i64, dbl, dbl, dbl, dbl]
        L.oracle_sufficient_progress.argtypes = [dbl] * 7
        L.oracle_armijo.argtypes = [dbl, dbl, pd, pd, i64, dbl, dbl, dbl]
        L.oracle_solve.argtypes = [vp, EVAL_FN, vp]
        L.oracle_differentiate.argtypes = [vp, EVAL_FN, vp]
        L.oracle_stats.argtypes = [vp, pi64]
        L.oracle_trace.restype = i64
        L.oracle_trace.argtypes = [vp, pd, i64]
        L.oracle_set_perm.argtypes = [vp, pi64]
        L.oracle_qdldl_permute_symmetric.argtypes = [i64, pi64, pi64, pd, pi64, pi64, pi64, pd, pi64]
        L.oracle_qdldl_etree.restype = i64
        L.oracle_qdldl_etree.argtypes = [i64, pi64, pi64, pi64, pi64, pi64]
        L.oracle_qdldl_factor.restype = i64
        L.oracle_qdldl_factor.argtypes = [i64, pi64, pi64, pd, pi64, pi64, pd, pd, pd, pi64, pi64]
        L.oracle_qdldl_solve.argtypes = [i64, pi64, pi64, pd, pd, pd]
        L.oracle_splitmix_uniform.argtypes = [C.c_uint64, C.c_uint64, dbl, dbl, i64, pd]
        _LIB = L
    return _LIB


def _pd(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _pi(a):
    return a.ctypes.data_as(C.POINTER(C.c_int64))


def splitmix_uniform(problem_id, stream_id, lo, hi, count):
    out = np.empty(int(count), dtype=np.float64)
    lib().oracle_splitmix_uniform(problem_id, stream_id, lo, hi, int(count), _pd(out))
    return out


class OracleSolver:
    """Mirror of the reference `Solver` (src/solver/solver.jl:1-27) over the C oracle.

    nonnegative_indices / second_order_indices are 1-based cone-local indices as in Julia."""

    def __init__(self, nx, np_, ne, nc, nonnegative_indices=None, second_order_indices=None):
        L = lib()
        if nonnegative_indices is None:
            nonnegative_indices = list(range(1, nc + 1))
        if second_order_indices is None:
            second_order_indices = [[]]
        self.nx, self.np, self.ne, self.nc = nx, np_, ne, nc
        self.n = nx + ne + nc
        self.N = nx + 2 * ne + 3 * nc
        nn = np.asarray(nonnegative_indices, dtype=np.int64)
        ptr = np.zeros(len(second_order_indices) + 1, dtype=np.int64)
        flat = []
        for k, c in enumerate(second_order_indices):
            flat.extend(c)
            ptr[k + 1] = len(flat)
        flat = np.asarray(flat, dtype=np.int64)
        self._h = L.oracle_create(nx, np_, ne, nc, len(nn), _pi(nn), len(second_order_indices), _pi(ptr), _pi(flat))
        self._L = L
        self._eval_keep = None

    def __del__(self):
        try:
            self._L.oracle_destroy(self._h)
        except Exception:
            pass

    # ---- data access ---------------------------------------------------------------------
    def buf(self, name):
        ln = C.c_int64()
        p = self._L.oracle_buffer(self._h, name.encode(), C.byref(ln))
        if ln.value < 0:
            raise KeyError(name)
        if ln.value == 0:
            return np.zeros(0)
        return np.ctypeslib.as_array(p, shape=(ln.value,))

    def mat(self, name, rows, cols):
        return self.buf(name).reshape((cols, rows)).T  # column-major view

    def index(self, name):
        ln = C.c_int64()
        p = self._L.oracle_index(self._h, name.encode(), C.byref(ln))
        if ln.value < 0:
            raise KeyError(name)
        if ln.value == 0:
            return np.zeros(0, dtype=np.int64)
        return np.ctypeslib.as_array(p, shape=(ln.value,)).copy()

    def set_int(self, name, v):
        self._L.oracle_int(self._h, name.encode())[0] = int(v)

    def get_int(self, name):
        return int(self._L.oracle_int(self._h, name.encode())[0])

    def set_opt(self, name, v):
        self.buf("opt." + name)[0] = v

    def point(self, name="solution"):
        w = self.buf(name)
        nx, ne, nc = self.nx, self.ne, self.nc
        o = np.cumsum([0, nx, ne, nc, ne, nc, nc])
        return dict(all=w, x=w[o[0]:o[1]], r=w[o[1]:o[2]], s=w[o[2]:o[3]], y=w[o[3]:o[4]], z=w[o[4]:o[5]], t=w[o[5]:o[6]])

    # ---- hot-path functions ----------------------------------------------------------------
    def cone(self, which=0, barrier=False, barrier_gradient=False, product=False, jacobian=False, target=False):
        self._L.oracle_cone(self._h, which, int(barrier), int(barrier_gradient), int(product), int(jacobian), int(target))

    def cone_violation(self, xhat, x, tau):
        xhat = np.ascontiguousarray(xhat, dtype=np.float64)
        x = np.ascontiguousarray(x, dtype=np.float64)
        return bool(self._L.oracle_cone_violation(self._h, _pd(xhat), _pd(x), tau))

    def residual(self):
        self._L.oracle_residual(self._h)

    def residual_jacobian_variables(self):
        self._L.oracle_residual_jacobian_variables(self._h)

    def residual_jacobian_variables_symmetric(self):
        self._L.oracle_residual_jacobian_variables_symmetric(self._h)

    def H_dense(self):
        out = np.zeros(self.N * self.N)
        self._L.oracle_H_dense(self._h, _pd(out))
        return out.reshape((self.N, self.N)).T

    def H_mul(self, v):
        v = np.ascontiguousarray(v, dtype=np.float64)
        out = np.zeros(self.N)
        self._L.oracle_H_mul(self._h, _pd(v), _pd(out))
        return out

    def K_dense(self):
        return self.mat("jacobian_variables_symmetric", self.n, self.n)

    def residual_symmetric(self, which=0):
        self._L.oracle_residual_symmetric(self._h, which)

    def factorize(self, update=True):
        return int(self._L.oracle_factorize(self._h, int(update)))

    def compute_inertia(self):
        out = np.zeros(3, dtype=np.int64)
        self._L.oracle_compute_inertia(self._h, _pi(out))
        return tuple(int(v) for v in out)

    def linear_solve(self, b, fact=True, update=True):
        b = np.ascontiguousarray(b, dtype=np.float64)
        x = np.zeros(self.n)
        self._L.oracle_linear_solve(self._h, _pd(x), _pd(b), int(fact), int(update))
        return x

    def search_direction_symmetric(self, which=0, fact=True):
        self._L.oracle_search_direction_symmetric(self._h, which, int(fact))

    def iterative_refinement(self):
        return bool(self._L.oracle_iterative_refinement(self._h))

    def inertia_correction(self):
        return int(self._L.oracle_inertia_correction(self._h))

    def search_direction(self):
        return int(self._L.oracle_search_direction(self._h))

    def merit(self, f, r, Phi):
        r = np.ascontiguousarray(r, dtype=np.float64)
        return float(self._L.oracle_merit(self._h, f, _pd(r), Phi))

    def merit_gradient(self):
        self._L.oracle_merit_gradient(self._h)

    def constraint_violation(self, g, r, h, s):
        a = [np.ascontiguousarray(v, dtype=np.float64) for v in (g, r, h, s)]
        return float(self._L.oracle_constraint_violation(self._h, *[_pd(v) for v in a]))

    def optimality_error(self):
        return float(self._L.oracle_optimality_error(self._h))

    def set_perm(self, perm_1based):
        p = np.ascontiguousarray(perm_1based, dtype=np.int64)
        self._L.oracle_set_perm(self._h, _pi(p))

    def trace(self):
        n = int(self._L.oracle_trace(self._h, None, 0))
        out = np.zeros((max(n, 1), self.N))
        self._L.oracle_trace(self._h, _pd(out), n)
        return out[:n]

    def stats(self):
        out = np.zeros(8, dtype=np.int64)
        self._L.oracle_stats(self._h, _pi(out))
        return dict(total_iterations=int(out[0]), outer=int(out[1]), factorizations=int(out[2]), refinement_failures=int(out[3]),
                    max_refinement_rounds=int(out[4]), lu_fallbacks=int(out[5]), last_refinement_rounds=int(out[6]))

    # ---- evaluation callback plumbing --------------------------------------------------------
    def make_eval(self, problem):
        """problem: object with method evaluate(flags, x, y, z, theta, out) where out(name) returns the writable
        numpy buffer of that ProblemData field (column-major flat)."""
        nx, ne, nc, npar = self.nx, self.ne, self.nc, self.np

        def cb(user, flags, px, py, pz, pth):
            try:
                x = np.ctypeslib.as_array(px, shape=(nx,)) if nx else np.zeros(0)
                y = np.ctypeslib.as_array(py, shape=(ne,)) if ne else np.zeros(0)
                z = np.ctypeslib.as_array(pz, shape=(nc,)) if nc else np.zeros(0)
                th = np.ctypeslib.as_array(pth, shape=(npar,)) if npar else np.zeros(0)
                problem.evaluate(flags, x, y, z, th, self.buf)
                return 0
            except Exception:  # pragma: no cover
                import traceback
                traceback.print_exc()
                return 1

        fn = EVAL_FN(cb)
        self._eval_keep = fn
        return fn

    def solve(self, problem):
        fn = self.make_eval(problem)
        return int(self._L.oracle_solve(self._h, fn, None))

    def differentiate(self, problem):
        fn = self.make_eval(problem)
        return int(self._L.oracle_differentiate(self._h, fn, None))
