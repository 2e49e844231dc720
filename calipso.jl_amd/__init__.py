"""calipso.jl_amd — MI355X (gfx950) implementation of CALIPSO's Newton/KKT hot path behind the reference's
Solver / initialize! / solve! surface (thowell/CALIPSO.jl, src/CALIPSO.jl:48-50).

This Python module is the host-side mirror used where Julia is unavailable (the Julia `ccall` module with the same
surface is in julia/CalipsoHIP.jl).  It keeps the reference's names: `Solver`, `Options`, `initialize_b` (= initialize!),
`solve_b` (= solve!), and the field names of `solver.solution`, `solver.data`, `solver.problem`.  All arithmetic of the hot
path runs in libcalipso_hip.so (hand-written HIP); this module only moves user-evaluated data across the C ABI.
"""
import ctypes as C

import numpy as np

from ._lib import CALLBACK_FN, EVAL_FN, CalipsoHipError, lib

__all__ = ["Solver", "Group", "LDLSolver", "SmallBatch", "Comm", "ordering", "symbolic", "Options", "initialize_b", "solve_b", "CalipsoHipError", "FLAGS", "splitmix_uniform",
           "mfma_f64_peak"]

# evaluate! flags (include/calipso_hip.h)
FLAGS = dict(
    objective=1 << 0, objective_gradient_variables=1 << 1, objective_jacobian_variables_variables=1 << 2,
    equality_constraint=1 << 3, equality_jacobian_variables=1 << 4, equality_dual_jacobian_variables=1 << 5,
    equality_dual_jacobian_variables_variables=1 << 6, cone_constraint=1 << 7, cone_jacobian_variables=1 << 8,
    cone_dual_jacobian_variables=1 << 9, cone_dual_jacobian_variables_variables=1 << 10,
    objective_jacobian_variables_parameters=1 << 11, equality_jacobian_parameters=1 << 12,
    equality_dual_jacobian_variables_parameters=1 << 13, cone_jacobian_parameters=1 << 14,
    cone_dual_jacobian_variables_parameters=1 << 15)
CONE_BARRIER, CONE_BARRIER_GRADIENT, CONE_PRODUCT, CONE_JACOBIAN, CONE_TARGET = 1, 2, 4, 8, 16

STATUS_TEXT = {-1: "inertia correction failure", -2: "cone search failure", -3: "evaluation callback failed",
               -4: "bad argument", -5: "HIP error", -6: "cone layout not supported"}


def _pd(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _pi(a):
    return a.ctypes.data_as(C.POINTER(C.c_int64))


def splitmix_uniform(problem_id, stream_id, lo, hi, count):
    """SplitMix64 uniform stream of the synthetic benchmark problems (host function of the library)."""
    out = np.empty(int(count), dtype=np.float64)
    lib().calipso_hip_splitmix_uniform(problem_id, stream_id, lo, hi, int(count), _pd(out))
    return out


class Options(dict):
    """Options(...) of src/solver/options.jl:6-59 — keyword arguments override the reference defaults held by the library."""

    def __init__(self, **kw):
        super().__init__(**kw)


class _Point:
    """read-only snapshot of a Point (src/solver/point.jl:1-22)"""

    def __init__(self, w, nx, ne, nc):
        o = np.cumsum([0, nx, ne, nc, ne, nc, nc])
        self.all = w
        self.variables = w[o[0]:o[1]]
        self.equality_slack = w[o[1]:o[2]]
        self.cone_slack = w[o[2]:o[3]]
        self.equality_dual = w[o[3]:o[4]]
        self.cone_dual = w[o[4]:o[5]]
        self.cone_slack_dual = w[o[5]:o[6]]
        self.primals = w[:o[3]]


class _Dims:
    pass


class Solver:
    """Solver(methods, num_variables, num_parameters, num_equality, num_cone; parameters, nonnegative_indices,
    second_order_indices, options)  — src/solver/solver.jl:46-150.

    `methods` stands in for ProblemMethods (src/solver/methods.jl): an object with
    evaluate(flags, x, y, z, theta, out) writing the requested ProblemData fields through out(name)."""

    def __init__(self, methods, num_variables, num_parameters, num_equality, num_cone, parameters=None,
                 nonnegative_indices=None, second_order_indices=None, options=None, device=0, structure=None):
        """structure = dict(row_first, row_last, hessian_block_start) (1-based) makes a STRUCTURED handle (calipso_hip_create_structured): only the
        stage blocks live on the device"""
        L = lib()
        self._L = L
        nx, npar, ne, nc = int(num_variables), int(num_parameters), int(num_equality), int(num_cone)
        if nonnegative_indices is None:
            nonnegative_indices = list(range(1, nc + 1))
        if second_order_indices is None:
            second_order_indices = [[]]
        nn = np.asarray(list(nonnegative_indices), dtype=np.int64)
        ptr = np.zeros(len(second_order_indices) + 1, dtype=np.int64)
        flat = []
        for k, c in enumerate(second_order_indices):
            flat.extend(c)
            ptr[k + 1] = len(flat)
        flat = np.asarray(flat, dtype=np.int64)
        h = C.c_void_p()
        self.structure = structure
        if structure is not None:
            rf = np.ascontiguousarray(structure["row_first"], dtype=np.int64); rl = np.ascontiguousarray(structure["row_last"], dtype=np.int64)
            hb = np.ascontiguousarray(structure["hessian_block_start"], dtype=np.int64)
            assert rf.size == ne + nc and rl.size == ne + nc
            rc = L.calipso_hip_create_structured(nx, npar, ne, nc, len(nn), _pi(nn), len(second_order_indices), _pi(ptr), _pi(flat), device, _pi(rf), _pi(rl),
                                                 hb.size, _pi(hb), C.byref(h))
        else:
            rc = L.calipso_hip_create(nx, npar, ne, nc, len(nn), _pi(nn), len(second_order_indices), _pi(ptr), _pi(flat), device, C.byref(h))
        if rc != 0:
            msg = L.calipso_hip_last_error(h if h.value else None).decode()
            if h.value:
                L.calipso_hip_destroy(h)
            raise CalipsoHipError("calipso_hip_create failed (%d): %s" % (rc, msg))
        self._h = h
        self.methods = methods
        self.dimensions = _Dims()
        d = self.dimensions
        d.variables, d.parameters, d.equality_slack, d.cone_slack = nx, npar, ne, nc
        d.equality_dual, d.cone_dual, d.cone_slack_dual = ne, nc, nc
        d.symmetric = d.primal = nx + ne + nc
        d.total = nx + 2 * ne + 3 * nc
        self.nx, self.np, self.ne, self.nc, self.n, self.N = nx, npar, ne, nc, d.symmetric, d.total
        self.indices = {k: self.index(k) for k in ("variables", "equality_slack", "cone_slack", "equality_dual", "cone_dual",
                                                    "cone_slack_dual", "symmetric_equality", "symmetric_cone", "primals", "duals",
                                                    "violation_equality", "violation_cone", "parameters", "cone_nonnegative",
                                                    "cone_second_order", "cone_second_order_ptr")}
        # host ProblemData (problem_data.jl:33-100) that the user functions fill
        z = np.zeros
        self.problem = dict(
            objective=z(1), objective_gradient_variables=z(nx), objective_jacobian_variables_variables=z(nx * nx),
            objective_jacobian_variables_parameters=z(nx * npar), equality_constraint=z(ne), equality_jacobian_variables=z(ne * nx),
            equality_jacobian_parameters=z(ne * npar), equality_dual_jacobian_variables=z(nx),
            equality_dual_jacobian_variables_variables=z(nx * nx), equality_dual_jacobian_variables_parameters=z(nx * npar),
            cone_constraint=z(nc), cone_jacobian_variables=z(nc * nx), cone_jacobian_parameters=z(nc * npar),
            cone_dual_jacobian_variables=z(nx), cone_dual_jacobian_variables_variables=z(nx * nx),
            cone_dual_jacobian_variables_parameters=z(nx * npar))
        self.parameters = np.zeros(npar) if parameters is None else np.asarray(parameters, dtype=np.float64).copy()
        if npar:
            self.set("parameters", self.parameters)
        self.options = {}
        for k, v in (options or {}).items():
            self.set_option(k, v)
        self._cb = EVAL_FN(self._evaluate_callback)

    def __del__(self):
        try:
            if getattr(self, "_h", None) is not None and self._h.value:
                self._L.calipso_hip_destroy(self._h)
                self._h = C.c_void_p()
        except Exception:
            pass

    # ---- plumbing ---------------------------------------------------------------------------------------------
    def _check(self, rc, what):
        if rc < 0:
            raise CalipsoHipError("%s: %s (%d): %s" % (what, STATUS_TEXT.get(rc, "error"), rc, self._L.calipso_hip_last_error(self._h).decode()))
        return rc

    def set(self, name, arr):
        a = np.ascontiguousarray(np.asarray(arr, dtype=np.float64).reshape(-1))
        self._check(self._L.calipso_hip_set_field(self._h, name.encode(), _pd(a), a.size), "set_field(%s)" % name)

    def get(self, name, length):
        out = np.zeros(int(length), dtype=np.float64)
        self._check(self._L.calipso_hip_get_field(self._h, name.encode(), _pd(out), out.size), "get_field(%s)" % name)
        return out

    def scalar(self, name):
        return float(self.get(name, 1)[0])

    def set_option(self, name, value):
        self.options[name] = value
        self.set("opt." + name, [float(value)])

    def index(self, name):
        n = self._L.calipso_hip_get_index(self._h, name.encode(), None, 0)
        if n < 0:
            raise KeyError(name)
        out = np.zeros(max(int(n), 1), dtype=np.int64)
        self._L.calipso_hip_get_index(self._h, name.encode(), _pi(out), n)
        return out[:n]

    def _out(self, name):
        return self.problem[name]

    _UPLOAD = [("objective", "objective"), ("objective_gradient_variables", "objective_gradient_variables"),
               ("equality_constraint", "equality_constraint"), ("equality_jacobian_variables", "equality_jacobian_variables"),
               ("equality_dual_jacobian_variables", "equality_dual_jacobian_variables"), ("cone_constraint", "cone_constraint"),
               ("cone_jacobian_variables", "cone_jacobian_variables"), ("cone_dual_jacobian_variables", "cone_dual_jacobian_variables"),
               ("equality_jacobian_parameters", "equality_jacobian_parameters"), ("cone_jacobian_parameters", "cone_jacobian_parameters")]

    # ---- sparse scatter of evaluate! (evaluate.jl:37-121): value caches + registered index lists instead of dense blocks ---------------
    def set_sparsity(self, field, rows, cols):
        """methods.<field>_sparsity: 1-based (row, col) lists in cache order; duplicates allowed (the last writer wins)"""
        r = np.ascontiguousarray(rows, dtype=np.int64); c = np.ascontiguousarray(cols, dtype=np.int64)
        self._check(self._L.calipso_hip_set_sparsity(self._h, field.encode(), r.size, _pi(r), _pi(c)), "set_sparsity(%s)" % field)

    def scatter_field(self, field, values):
        v = np.ascontiguousarray(values, dtype=np.float64)
        self._check(self._L.calipso_hip_scatter_field(self._h, field.encode(), _pd(v), v.size), "scatter_field(%s)" % field)

    def scatter_hessian(self, objective=None, equality_dual=None, cone_dual=None):
        a = [None if v is None else np.ascontiguousarray(v, dtype=np.float64) for v in (objective, equality_dual, cone_dual)]
        args = []
        for v in a:
            args += [_pd(v) if v is not None else None, 0 if v is None else v.size]
        self._check(self._L.calipso_hip_scatter_hessian(self._h, *args), "scatter_hessian")

    def upload(self, flags):
        """hand the flagged ProblemData fields to the device (the tail of evaluate!, src/solver/evaluate.jl:37-121)"""
        p = self.problem
        for key, field in self._UPLOAD:
            if flags & FLAGS[key] and p[key].size:
                self.set(field, p[key])
        hess = FLAGS["objective_jacobian_variables_variables"] | FLAGS["equality_dual_jacobian_variables_variables"] | FLAGS["cone_dual_jacobian_variables_variables"]
        if flags & hess:
            # Lxx = fxx + (g'y)xx + (h'z)xx  (residual_jacobian_variables.jl:10-16; tensor terms iff options.constraint_tensor)
            L = p["objective_jacobian_variables_variables"].copy()
            if self.options.get("constraint_tensor", 1.0):
                L += p["equality_dual_jacobian_variables_variables"]
                L += p["cone_dual_jacobian_variables_variables"]
            self.set("lagrangian_hessian", L)
        par = (FLAGS["objective_jacobian_variables_parameters"] | FLAGS["equality_dual_jacobian_variables_parameters"] |
               FLAGS["cone_dual_jacobian_variables_parameters"])
        if flags & par and self.np:
            G = p["objective_jacobian_variables_parameters"].copy()     # residual_jacobian_parameters.jl:8-14
            G += p["equality_dual_jacobian_variables_parameters"]
            G += p["cone_dual_jacobian_variables_parameters"]
            self.set("lagrangian_gradient_parameters", G)

    def _evaluate_callback(self, user, flags, px, py, pz, pth):
        try:
            as_arr = np.ctypeslib.as_array
            x = as_arr(px, shape=(self.nx,))
            y = as_arr(py, shape=(self.ne,)) if self.ne else np.zeros(0)
            z = as_arr(pz, shape=(self.nc,)) if self.nc else np.zeros(0)
            th = as_arr(pth, shape=(self.np,)) if self.np else np.zeros(0)
            self.methods.evaluate(flags, x, y, z, th, self._out)
            self.upload(flags)
            return 0
        except Exception:   # pragma: no cover
            import traceback
            traceback.print_exc()
            return 1

    def evaluate(self, flags, which=0):
        """evaluate!(problem, methods, idx, point, parameters; <flags>) at the solution (0) or candidate (1) point"""
        w = self.get("solution" if which == 0 else "candidate", self.N)
        pt = _Point(w, self.nx, self.ne, self.nc)
        self.methods.evaluate(flags, pt.variables, pt.equality_dual, pt.cone_dual, self.parameters, self._out)
        self.upload(flags)

    # ---- reference-style views -----------------------------------------------------------------------------------
    @property
    def solution(self):
        return _Point(self.get("solution", self.N), self.nx, self.ne, self.nc)

    @property
    def candidate(self):
        return _Point(self.get("candidate", self.N), self.nx, self.ne, self.nc)

    def data(self, name):
        """solver.data.<name> (solver_data.jl): residual, step, residual_symmetric, step_symmetric, merit_gradient, solution_sensitivity"""
        sizes = dict(residual=self.N, residual_error=self.N, step=self.N, step_correction=self.N, residual_symmetric=self.n,
                     step_symmetric=self.n, merit_gradient=self.n, solution_sensitivity=self.N * self.np,
                     jacobian_parameters=self.N * self.np)
        v = self.get(name, sizes[name])
        if name in ("solution_sensitivity", "jacobian_parameters"):
            return v.reshape(self.np, self.N).T
        if name in ("residual", "residual_error", "step", "step_correction"):
            return _Point(v, self.nx, self.ne, self.nc)
        return v

    def jacobian_variables_symmetric(self):
        self._check(self._L.calipso_hip_residual_jacobian_variables_symmetric(self._h), "residual_jacobian_variables_symmetric")
        return self.get("jacobian_variables_symmetric", self.n * self.n).reshape(self.n, self.n).T

    # ---- hot-path entry points (one per reference function) ---------------------------------------------------------
    def cone(self, which=0, barrier=False, barrier_gradient=False, product=False, jacobian=False, target=False):
        fl = (CONE_BARRIER * barrier) | (CONE_BARRIER_GRADIENT * barrier_gradient) | (CONE_PRODUCT * product) | (CONE_JACOBIAN * jacobian) | (CONE_TARGET * target)
        self._check(self._L.calipso_hip_cone(self._h, which, int(fl)), "cone")

    def residual(self):
        self._check(self._L.calipso_hip_residual(self._h), "residual")

    def violations(self):
        out = np.zeros(5)
        self._check(self._L.calipso_hip_violations(self._h, _pd(out)), "violations")
        return dict(residual_violation=out[0], optimality_violation=out[1], slack_violation=out[2], equality_violation=out[3],
                    cone_product_violation=out[4])

    def jacobian_variables_mul(self, v):
        v = np.ascontiguousarray(v, dtype=np.float64)
        out = np.zeros(self.N)
        self._check(self._L.calipso_hip_jacobian_variables_mul(self._h, _pd(v), _pd(out)), "jacobian_variables_mul")
        return out

    def factorize(self):
        inertia = np.zeros(3, dtype=np.int64)
        rc = self._check(self._L.calipso_hip_factorize(self._h, _pi(inertia)), "factorize")
        return tuple(int(v) for v in inertia), rc

    def inertia_correction(self):
        nf = C.c_int64(0)
        self._check(self._L.calipso_hip_inertia_correction(self._h, C.byref(nf)), "inertia_correction")
        return int(nf.value)

    def residual_symmetric(self, which=0):
        self._check(self._L.calipso_hip_residual_symmetric(self._h, which), "residual_symmetric")

    def linear_solve(self):
        self._check(self._L.calipso_hip_linear_solve(self._h), "linear_solve")

    def search_direction_symmetric(self, which=0):
        self._check(self._L.calipso_hip_search_direction_symmetric(self._h, which), "search_direction_symmetric")

    def iterative_refinement(self):
        r = C.c_int32(0)
        nrm = C.c_double(0.0)
        rc = self._check(self._L.calipso_hip_iterative_refinement(self._h, C.byref(r), C.byref(nrm)), "iterative_refinement")
        return rc == 0, int(r.value), float(nrm.value)

    def search_direction(self):
        return self._check(self._L.calipso_hip_search_direction(self._h), "search_direction")

    def search_direction_nonsymmetric(self):
        """step = H \\ residual on the unreduced system (src/solver/search_direction.jl:106-119), the reference's fallback when refinement fails"""
        return self._check(self._L.calipso_hip_search_direction_nonsymmetric(self._h), "search_direction_nonsymmetric")

    def cone_search(self):
        a, b = C.c_double(0), C.c_double(0)
        self._check(self._L.calipso_hip_cone_search(self._h, C.byref(a), C.byref(b)), "cone_search")
        return float(a.value), float(b.value)

    def cone_violation(self, xhat, x, tau):
        xhat = np.ascontiguousarray(xhat, dtype=np.float64)
        x = np.ascontiguousarray(x, dtype=np.float64)
        v = C.c_int32(0)
        self._check(self._L.calipso_hip_cone_violation(self._h, _pd(xhat), _pd(x), tau, C.byref(v)), "cone_violation")
        return bool(v.value)

    def make_candidate(self, step_size, with_cone_slack=False):
        self._check(self._L.calipso_hip_candidate(self._h, step_size, int(with_cone_slack)), "candidate")

    def merit(self, which=0):
        m = C.c_double(0)
        self._check(self._L.calipso_hip_merit(self._h, which, C.byref(m)), "merit")
        return float(m.value)

    def merit_gradient(self):
        self._check(self._L.calipso_hip_merit_gradient(self._h), "merit_gradient")

    def constraint_violation(self, which=0):
        t = C.c_double(0)
        self._check(self._L.calipso_hip_constraint_violation(self._h, which, C.byref(t)), "constraint_violation")
        return float(t.value)

    def differentiate(self):
        self._check(self._L.calipso_hip_differentiate(self._h, self._cb, None), "differentiate")

    def set_device_evaluator(self, fn_ptr, user=None):
        """install a device-side evaluator (calipso_device_eval_fn, include/calipso_hip.h): `fn_ptr` is the C function's address (e.g.
        ctypes.cast(lib.sym, c_void_p)), `user` its opaque pointer; solve_b / differentiate then never call back into Python"""
        self._check(self._L.calipso_hip_set_device_evaluator(self._h, fn_ptr, user), "set_device_evaluator")
        self._device_eval = fn_ptr is not None

    def set_device_block_evaluator(self, fn_ptr, user=None):
        """structured handles: install an evaluator that writes the packed blocks themselves (calipso_device_block_eval_fn): no dense scratch on the device"""
        self._check(self._L.calipso_hip_set_device_block_evaluator(self._h, fn_ptr, user), "set_device_block_evaluator")
        self._device_eval = fn_ptr is not None

    def device_evaluate(self, flags, which=0):
        self._check(self._L.calipso_hip_device_evaluate(self._h, which, int(flags)), "device_evaluate")

    def set_callbacks(self, inner=None, outer=None):
        """callback_inner(custom, solver) / callback_outer(custom, solver) (src/solver/solver.jl:183,193): python callables taking the Solver"""
        self._cbi = CALLBACK_FN(lambda u, h: inner(self)) if inner else None
        self._cbo = CALLBACK_FN(lambda u, h: outer(self)) if outer else None
        ci = C.cast(self._cbi, C.c_void_p) if self._cbi else None
        co = C.cast(self._cbo, C.c_void_p) if self._cbo else None
        self._check(self._L.calipso_hip_set_callbacks(self._h, ci, co, None), "set_callbacks")

    def stats(self):
        out = np.zeros(8, dtype=np.int64)
        self._L.calipso_hip_stats(self._h, _pi(out))
        return dict(total_iterations=int(out[0]), outer=int(out[1]), factorizations=int(out[2]), refinement_failures=int(out[3]),
                    max_refinement_rounds=int(out[4]), fallbacks=int(out[5]), last_refinement_rounds=int(out[6]), newton_steps=int(out[7]))

    # ---- device-resident QP (synthetic benchmark) -----------------------------------------------------------------------
    def qp_attach(self, P, q, A, b, G, h, objective_scale=0.5):
        f = lambda M: np.ascontiguousarray(np.asarray(M, dtype=np.float64).T).reshape(-1)   # column-major
        v = lambda a: np.ascontiguousarray(np.asarray(a, dtype=np.float64))
        Pc, Ac, Gc = f(P), f(A) if self.ne else np.zeros(1), f(G) if self.nc else np.zeros(1)
        qv, bv, hv = v(q), v(b) if self.ne else np.zeros(1), v(h) if self.nc else np.zeros(1)
        self._check(self._L.calipso_hip_qp_attach(self._h, _pd(Pc), _pd(qv), _pd(Ac), _pd(bv), _pd(Gc), _pd(hv), objective_scale), "qp_attach")

    def qp_evaluate(self, flags, which=0):
        self._check(self._L.calipso_hip_qp_evaluate(self._h, which, int(flags)), "qp_evaluate")

    def newton_step(self, advance=False):
        info = np.zeros(6)
        rc = self._check(self._L.calipso_hip_newton_step(self._h, int(advance), _pd(info)), "newton_step")
        return dict(status=rc, step_size=info[0], step_size_cone_slack_dual=info[1], refinement_rounds=int(info[2]),
                    factorizations=int(info[3]), merit_candidate=info[4], violation_candidate=info[5])

    def newton_steps(self, count, advance=False):
        """`count` Newton steps in one call (calipso_hip_newton_steps): the list of the per-step dicts newton_step returns"""
        info = np.zeros(6 * max(int(count), 1))
        status = np.zeros(max(int(count), 1), dtype=np.int32)
        self._check(self._L.calipso_hip_newton_steps(self._h, int(count), int(advance), _pd(info), status.ctypes.data_as(C.POINTER(C.c_int32))), "newton_steps")
        I = info.reshape(-1, 6)
        return [dict(status=int(status[k]), step_size=I[k, 0], step_size_cone_slack_dual=I[k, 1], refinement_rounds=int(I[k, 2]), factorizations=int(I[k, 3]),
                     merit_candidate=I[k, 4], violation_candidate=I[k, 5]) for k in range(int(count))]

    def phase_times(self):
        out = np.zeros(9)
        self._L.calipso_hip_phase_times(self._h, _pd(out))
        return out

    def kernel_times(self):
        """[0] ms of the panel-step launches of the last LDL^T (the pivot chain), [1] their number, [2] NP, [3] bytes of the device slab"""
        out = np.zeros(8)
        self._L.calipso_hip_kernel_times(self._h, _pd(out))
        return out

    def structure_work(self):
        """work of one Newton step on a handle that exploits its stage structure: dict(schur_flops, packed_doubles, segment_pairs, factor_flops, factor_nnz, order, structured)"""
        out = np.zeros(8)
        self._check(self._L.calipso_hip_structure_work(self._h, _pd(out)), "structure_work")
        return dict(schur_flops=out[0], packed_doubles=out[1], segment_pairs=int(out[2]), factor_flops=out[3], factor_nnz=out[4], order=int(out[5]), structured=bool(out[6]))

    def padded_nx(self):
        return int(self.kernel_times()[2])

    def device_bytes(self):
        return int(self.kernel_times()[3])

    def analyze_structure(self):
        """stage-banded structure from the non-zero pattern of the blocks currently on the device (include/calipso_hip.h); returns
        dict(half_bandwidth, band_blocks (0 = dense treatment), equality_rows_per_group, cone_rows_per_group)"""
        out = np.zeros(4, dtype=np.int64)
        self._check(self._L.calipso_hip_analyze_structure(self._h, _pi(out)), "analyze_structure")
        return dict(half_bandwidth=int(out[0]), band_blocks=int(out[1]), equality_rows_per_group=int(out[2]), cone_rows_per_group=int(out[3]))

    def clear_structure(self):
        self._check(self._L.calipso_hip_clear_structure(self._h), "clear_structure")

    def set_stage_parallel(self, on=True, batch=1):
        """after analyze_structure: factor the Schur complement by the multifrontal sparse LDL^T over a nested dissection of its pattern (for a
        trajectory problem log2(stages) launches instead of the chain of nx pivots); batch >= the largest group this handle leads.
        Returns dict(levels, largest_front, nnz_upper)."""
        out = np.zeros(4, dtype=np.int64)
        self._check(self._L.calipso_hip_set_stage_parallel(self._h, 1 if on else 0, int(batch), _pi(out)), "set_stage_parallel")
        return dict(levels=int(out[0]), largest_front=int(out[1]), nnz_upper=int(out[2]))

    def set_stage_blocks(self, on=True):
        """after analyze_structure: pack the blocks of [gx; hx] and of the Lagrangian Hessian and run the mat-vecs / the Schur complement on them
        (csrc/blocks.hip).  Returns dict(z_blocks, hessian_blocks, segments, packed_doubles)."""
        out = np.zeros(4, dtype=np.int64)
        self._check(self._L.calipso_hip_set_stage_blocks(self._h, 1 if on else 0, _pi(out)), "set_stage_blocks")
        return dict(z_blocks=int(out[0]), hessian_blocks=int(out[1]), segments=int(out[2]), packed_doubles=int(out[3]))

    def synchronize(self):
        self._check(self._L.calipso_hip_synchronize(self._h), "synchronize")

    def streams_concurrent(self, other):
        """calipso_hip_streams_concurrent: (side by side?, short kernel behind self's long one [us], behind other's [us], the long kernel [us], one chain of short kernels alone [us], both chains at once [us])"""
        out = np.zeros(6)
        self._check(self._L.calipso_hip_streams_concurrent(self._h, other._h, _pd(out)), "streams_concurrent")
        return bool(out[0]), float(out[1]), float(out[2]), float(out[3]), float(out[4]), float(out[5])

    def rebind_stream(self, priority_class=-1):
        """calipso_hip_rebind_stream: a new HIP stream for this handle (another hardware queue)"""
        self._check(self._L.calipso_hip_rebind_stream(self._h, int(priority_class)), "rebind_stream")


class Group:
    """Up to 128 Solver handles of one shape (same dimensions and cone layout, same device) stepped in lockstep: every kernel
    launch of a group step covers all members (include/calipso_hip.h, "groups").  The reference has no counterpart — its
    `Solver`s are independent objects; this is how BASELINE config C4 keeps many of them in flight on one GPU."""

    def __init__(self, solvers):
        import ctypes as C
        self.solvers = list(solvers)
        self._L = self.solvers[0]._L
        arr = (C.c_void_p * len(self.solvers))(*[s._h for s in self.solvers])
        h = C.c_void_p()
        rc = self._L.calipso_hip_group_create(arr, len(self.solvers), C.byref(h))
        if rc != 0:
            raise CalipsoHipError("group_create failed (%d): %s" % (rc, self._L.calipso_hip_last_error(self.solvers[0]._h).decode()))
        self._g = h

    def newton_step(self, advance=False):
        """calipso_hip_newton_step for every member; returns one info dict per member (same keys as Solver.newton_step)"""
        import ctypes as C
        B = len(self.solvers)
        info = np.zeros(6 * B)
        status = (C.c_int32 * B)()
        rc = self._L.calipso_hip_group_newton_step(self._g, int(advance), _pd(info), status)
        if rc < 0:
            raise CalipsoHipError("group_newton_step: %s (%d): %s" % (STATUS_TEXT.get(rc, "error"), rc,
                                                                      self._L.calipso_hip_last_error(self.solvers[0]._h).decode()))
        out = []
        for k in range(B):
            r = info[6 * k: 6 * k + 6]
            out.append(dict(status=int(status[k]), step_size=r[0], step_size_cone_slack_dual=r[1], refinement_rounds=int(r[2]),
                            factorizations=int(r[3]), merit_candidate=r[4], violation_candidate=r[5]))
        return out

    def solve(self):
        """solve!(solver) for every member in lockstep (device evaluators attached); returns the per-member results
        (1 converged, 0 iteration caps reached, negative = error code of that member)"""
        import ctypes as C
        B = len(self.solvers)
        res = (C.c_int32 * B)()
        self._evals = (EVAL_FN * B)(*[s._cb for s in self.solvers])       # host callbacks (used by members without a device evaluator)
        self._L.calipso_hip_group_set_evaluators(self._g, self._evals, None)
        rc = self._L.calipso_hip_group_solve(self._g, res)
        if rc < 0:
            raise CalipsoHipError("group_solve: %s (%d): %s" % (STATUS_TEXT.get(rc, "error"), rc, self._L.calipso_hip_last_error(self.solvers[0]._h).decode()))
        return [int(v) for v in res]

    def phase_times(self):
        return self.solvers[0].phase_times()

    def synchronize(self):
        self.solvers[0].synchronize()

    def close(self):
        if self._g is not None:
            self._L.calipso_hip_group_destroy(self._g)
            self._g = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _csc_1based(A):
    import scipy.sparse as sp
    A = sp.csc_matrix(A)
    A.sort_indices()
    return (A, np.ascontiguousarray(A.indptr, dtype=np.int64) + 1, np.ascontiguousarray(A.indices, dtype=np.int64) + 1,
            np.ascontiguousarray(A.data, dtype=np.float64))


ORDERINGS = dict(natural=0, rcm=1, minimum_degree=2, nested_dissection=4)
SPARSE_METHODS = dict(ORDERINGS, nested_dissection_columns=5)     # 5: nested-dissection order with the column-level numeric phase (no fronts)


def ordering(A, method="rcm"):
    """elimination order (1-based, perm[k] = vertex eliminated k-th) of the symmetric pattern of A: "natural", "rcm", "minimum_degree"
    (calipso_hip_ordering; the reference takes perm = amd(A), qdldl.jl:135).  Host function: needs no device."""
    A, colptr, rowval, _ = _csc_1based(A)
    n = A.shape[0]
    perm = np.zeros(n, dtype=np.int64)
    rc = lib().calipso_hip_ordering(n, _pi(colptr), _pi(rowval), ORDERINGS[method], _pi(perm))
    if rc != 0:
        raise CalipsoHipError("calipso_hip_ordering failed (%d)" % rc)
    return perm


def symbolic(A, perm=None):
    """permute_symmetric + QDLDL_etree! of triu(A) under perm (qdldl.jl:358-395,642-742): dict(Pp, Pi, AtoPAPt, etree, Lnz, nnzL, half_bandwidth)"""
    A, colptr, rowval, _ = _csc_1based(A)
    n, nnz = A.shape[0], A.nnz
    Pp, Pi, mp = np.zeros(n + 1, dtype=np.int64), np.zeros(max(nnz, 1), dtype=np.int64), np.zeros(max(nnz, 1), dtype=np.int64)
    et, lnz, info = np.zeros(n, dtype=np.int64), np.zeros(n, dtype=np.int64), np.zeros(2, dtype=np.int64)
    pp = None if perm is None else np.ascontiguousarray(perm, dtype=np.int64)
    tot = lib().calipso_hip_symbolic(n, _pi(colptr), _pi(rowval), _pi(pp) if pp is not None else None, _pi(Pp), _pi(Pi), _pi(mp), _pi(et), _pi(lnz), _pi(info))
    if tot < -1:
        raise CalipsoHipError("calipso_hip_symbolic failed (%d)" % tot)
    return dict(Pp=Pp, Pi=Pi[:int(info[1])], AtoPAPt=mp[:nnz], etree=et, Lnz=lnz, nnzL=int(tot), half_bandwidth=int(info[0]))


class LDLSolver:
    """LDLSolver / ldl_solver(A) of src/solver/linear_solver.jl:1-60 on the device: factorize!(s, A), compute_inertia!(s),
    linear_solve!(s, x, A, b).  A is a scipy.sparse CSC matrix (or anything scipy can convert); only triu(A) is read."""

    def __init__(self, n, device=0):
        self._L = lib()
        self.n = int(n)
        h = C.c_void_p()
        rc = self._L.calipso_hip_ldl_create(self.n, device, C.byref(h))
        if rc != 0:
            raise CalipsoHipError("calipso_hip_ldl_create failed (%d): %s" % (rc, self._L.calipso_hip_last_error(h if h.value else None).decode()))
        self._h = h
        self.inertia = (0, 0, 0)          # Inertia(positive, negative, zero)  inertia.jl:1-5

    def _check(self, rc, what):
        if rc < 0:
            raise CalipsoHipError("%s: %s (%d): %s" % (what, STATUS_TEXT.get(rc, "error"), rc, self._L.calipso_hip_last_error(self._h).decode()))
        return rc

    def analyze(self, A, method="rcm", perm=None):
        """install the elimination order of the following factorisations ("natural", "rcm", "minimum_degree", or perm=<1-based order>);
        returns (perm, dict(half_bandwidth, band_blocks (0 = dense treatment), nnzL, nnz_upper))"""
        A, colptr, rowval, _ = _csc_1based(A)
        p = np.zeros(self.n, dtype=np.int64) if perm is None else np.ascontiguousarray(perm, dtype=np.int64).copy()
        info = np.zeros(4, dtype=np.int64)
        m = 3 if perm is not None else ORDERINGS[method]
        self._check(self._L.calipso_hip_ldl_analyze_csc(self._h, self.n, _pi(colptr), _pi(rowval), m, _pi(p), _pi(info)), "analyze")
        return p, dict(half_bandwidth=int(info[0]), band_blocks=int(info[1]), nnzL=int(info[2]), nnz_upper=int(info[3]))

    def factorize(self, A):
        """factorize!(s, A; update) + compute_inertia!(s); returns the warning status (1 = zero pivot met)"""
        A, colptr, rowval, nzval = _csc_1based(A)
        out = np.zeros(3, dtype=np.int64)
        rc = self._check(self._L.calipso_hip_ldl_factorize_csc(self._h, self.n, _pi(colptr), _pi(rowval), _pd(nzval), _pi(out)), "factorize!")
        self.inertia = tuple(int(v) for v in out)
        return rc

    def compute_inertia(self):
        out = np.zeros(3, dtype=np.int64)
        self._check(self._L.calipso_hip_ldl_inertia(self._h, _pi(out)), "compute_inertia!")
        self.inertia = tuple(int(v) for v in out)
        return self.inertia

    def linear_solve(self, b, A=None, fact=False):
        """linear_solve!(s, x, A, b; fact): b is a vector (n) or a matrix (n x nrhs); returns x"""
        if fact:
            self.factorize(A)
        b = np.asarray(b, dtype=np.float64)
        nrhs = 1 if b.ndim == 1 else b.shape[1]
        bf = np.ascontiguousarray(b.reshape(self.n, nrhs).T).reshape(-1)          # column-major
        x = np.zeros_like(bf)
        self._check(self._L.calipso_hip_ldl_solve(self._h, self.n, nrhs, _pd(bf), _pd(x)), "linear_solve!")
        return x if b.ndim == 1 else x.reshape(nrhs, self.n).T

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._L.calipso_hip_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class SparseLDL:
    """Sparse LDL^T on the device (include/calipso_hip.h, "sparse LDL^T on the device"): qdldl(A; perm) / QDLDL_factor! / solve! of
    src/solver/qdldl.jl for a scipy.sparse matrix A (only triu(A) is read), memory O(nnz(L)), level-scheduled over the elimination tree.
    method: "natural", "rcm", "minimum_degree", "nested_dissection" (multifrontal over the dissection tree when every front fits one CU's LDS,
    else as "nested_dissection_columns": the same order with the column-level numeric phase), or perm=<1-based order>."""

    def __init__(self, A, method="nested_dissection", perm=None, device=0):
        self._L = lib()
        A, colptr, rowval, _ = _csc_1based(A)
        self.n, self.nnz = A.shape[0], A.nnz
        pp = None if perm is None else np.ascontiguousarray(perm, dtype=np.int64)
        h = C.c_void_p()
        rc = self._L.calipso_hip_sparse_create(self.n, _pi(colptr), _pi(rowval), 3 if pp is not None else SPARSE_METHODS[method], _pi(pp) if pp is not None else None,
                                               device, C.byref(h))
        if rc != 0:
            msg = self._L.calipso_hip_sparse_last_error(h if h.value else None).decode()
            if h.value:
                self._L.calipso_hip_sparse_destroy(h)
            raise CalipsoHipError("calipso_hip_sparse_create failed (%d): %s" % (rc, msg))
        self._h = h
        info = np.zeros(8, dtype=np.int64)
        self._check(self._L.calipso_hip_sparse_info(self._h, _pi(info)), "sparse_info")
        self.info = dict(n=int(info[0]), nnz_upper=int(info[1]), nnzL=int(info[2]), levels=int(info[3]), launches=int(info[4]), widest_level=int(info[5]),
                         multiply_adds=int(info[6]), lds_accumulator=int(info[7]) == 1,
                         numeric={0: "columns_global_accumulator", 1: "columns_lds_accumulator", 2: "multifrontal"}[int(info[7])])
        self.inertia = (0, 0, 0)

    def _check(self, rc, what):
        if rc < 0:
            raise CalipsoHipError("%s failed (%d): %s" % (what, rc, self._L.calipso_hip_sparse_last_error(self._h).decode()))
        return rc

    def set_batch(self, batch):
        """treat `batch` matrices of the analysed pattern per factorize / solve call (values (batch, nnz), right-hand sides (batch, n[, nrhs]))"""
        self._check(self._L.calipso_hip_sparse_set_batch(self._h, int(batch)), "sparse_set_batch")
        self.batch = int(batch)

    def factorize(self, A):
        """numeric factorisation of A (same pattern as analysed; a scipy.sparse matrix, or the nnz values in CSC order — (batch, nnz) after
        set_batch); returns the warning status (1 = an exact zero pivot was met, inertia[0] = -1 for that matrix)"""
        B = getattr(self, "batch", 1)
        if hasattr(A, "shape") and len(getattr(A, "shape")) == 2 and not isinstance(A, np.ndarray):
            A, _, _, vals = _csc_1based(A)
            if A.nnz != self.nnz:
                raise CalipsoHipError("SparseLDL.factorize: the pattern differs from the analysed one")
        else:
            vals = np.ascontiguousarray(A, dtype=np.float64).reshape(-1)
        if vals.size != B * self.nnz:
            raise CalipsoHipError("SparseLDL.factorize: expected %d x %d values" % (B, self.nnz))
        inr = np.zeros(3 * B, dtype=np.int64)
        rc = self._check(self._L.calipso_hip_sparse_factorize(self._h, _pd(vals), _pi(inr)), "sparse_factorize")
        self.inertia = tuple(int(v) for v in inr[:3])
        self.inertia_all = inr.reshape(B, 3)
        return rc

    def solve(self, b):
        """x = A^-1 b for a vector or an (n, nrhs) matrix; after set_batch: b of shape (batch, n) or (batch, n, nrhs)"""
        b = np.asarray(b, dtype=np.float64)
        B = getattr(self, "batch", 1)
        if B > 1:
            b3 = b.reshape(B, self.n, -1)
            nrhs = b3.shape[2]
            flat = np.ascontiguousarray(np.transpose(b3, (0, 2, 1))).reshape(-1)     # per matrix column-major n x nrhs
            X = np.zeros_like(flat)
            self._check(self._L.calipso_hip_sparse_solve(self._h, nrhs, _pd(flat), _pd(X)), "sparse_solve")
            X = np.transpose(X.reshape(B, nrhs, self.n), (0, 2, 1))
            return X[:, :, 0].copy() if b.ndim == 2 else X.copy()
        one = b.ndim == 1
        Bm = np.ascontiguousarray(b.reshape(self.n, -1).T).reshape(-1)          # column-major n x nrhs
        nrhs = Bm.size // self.n
        X = np.zeros_like(Bm)
        self._check(self._L.calipso_hip_sparse_solve(self._h, nrhs, _pd(Bm), _pd(X)), "sparse_solve")
        X = X.reshape(nrhs, self.n).T
        return X[:, 0].copy() if one else X.copy()

    def factorize_device(self, values):
        """factorize with the values already on the device: a float64 CUDA/HIP torch tensor of batch x nnz entries (CSC order)"""
        B = getattr(self, "batch", 1)
        assert values.is_cuda and values.is_contiguous() and values.numel() == B * self.nnz and values.element_size() == 8
        inr = np.zeros(3 * B, dtype=np.int64)
        rc = self._check(self._L.calipso_hip_sparse_factorize_device(self._h, C.c_void_p(values.data_ptr()), _pi(inr)), "sparse_factorize_device")
        self.inertia = tuple(int(v) for v in inr[:3])
        self.inertia_all = inr.reshape(B, 3)
        return rc

    def solve_device(self, b, out=None):
        """solve with device-resident right-hand sides: float64 torch tensor laid out batch x nrhs x n (each right-hand side contiguous);
        returns a tensor of the same layout (or writes `out`)"""
        B = getattr(self, "batch", 1)
        assert b.is_cuda and b.is_contiguous() and b.element_size() == 8 and b.numel() % (B * self.n) == 0
        nrhs = b.numel() // (B * self.n)
        if out is None:
            out = b.new_empty(b.shape)
        self._check(self._L.calipso_hip_sparse_solve_device(self._h, nrhs, C.c_void_p(b.data_ptr()), C.c_void_p(out.data_ptr())), "sparse_solve_device")
        return out

    def select(self, instance):
        """which matrix of the batch factor() reads"""
        self._check(self._L.calipso_hip_sparse_select(self._h, int(instance)), "sparse_select")

    def factor(self):
        """(perm (1-based), L as a scipy.sparse CSC unit-lower matrix, D)"""
        import scipy.sparse as sp
        nl = self.info["nnzL"]
        perm, Lp, Li = np.zeros(self.n, dtype=np.int64), np.zeros(self.n + 1, dtype=np.int64), np.zeros(max(nl, 1), dtype=np.int64)
        Lx, D = np.zeros(max(nl, 1)), np.zeros(self.n)
        self._check(self._L.calipso_hip_sparse_get_factor(self._h, _pi(perm), _pi(Lp), _pi(Li), _pd(Lx), _pd(D)), "sparse_get_factor")
        Lm = sp.csc_matrix((Lx[:nl], Li[:nl] - 1, Lp - 1), shape=(self.n, self.n)) + sp.identity(self.n, format="csc")
        return perm, Lm, D

    def timing(self):
        """(device milliseconds of the last factorisation, of the last solve)"""
        ms = np.zeros(2)
        self._check(self._L.calipso_hip_sparse_timing(self._h, _pd(ms)), "sparse_timing")
        return float(ms[0]), float(ms[1])

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._L.calipso_hip_sparse_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class SmallBatch:
    """`batch` independent small systems (n <= 128) factored and solved for `nrhs` right-hand sides each in ONE launch, every system
    resident in the LDS of one workgroup (include/calipso_hip.h, "batched small systems"): the sensitivity solves of differentiate!
    (src/solver/differentiate.jl:29-58) for many MPC steps at once."""

    def __init__(self, n, nrhs, batch, device=0):
        self._L = lib()
        self.n, self.nrhs, self.batch = int(n), int(nrhs), int(batch)
        h = C.c_void_p()
        rc = self._L.calipso_hip_small_create(self.n, self.nrhs, self.batch, device, C.byref(h))
        if rc != 0:
            msg = self._L.calipso_hip_small_last_error(h if h.value else None).decode()
            if h.value:
                self._L.calipso_hip_small_destroy(h)
            raise CalipsoHipError("calipso_hip_small_create failed (%d): %s" % (rc, msg))
        self._h = h

    def _check(self, rc, what):
        if rc < 0:
            raise CalipsoHipError("%s failed (%d): %s" % (what, rc, self._L.calipso_hip_small_last_error(self._h).decode()))
        return rc

    def set(self, K=None, B=None):
        """K: (batch, n, n) matrices (only the upper triangles are read); B: (batch, n, nrhs) right-hand sides"""
        kk = bb = None
        if K is not None:
            kk = np.ascontiguousarray(np.transpose(np.asarray(K, dtype=np.float64).reshape(self.batch, self.n, self.n), (0, 2, 1))).reshape(-1)
        if B is not None:
            bb = np.ascontiguousarray(np.transpose(np.asarray(B, dtype=np.float64).reshape(self.batch, self.n, self.nrhs), (0, 2, 1))).reshape(-1)
        self._check(self._L.calipso_hip_small_set(self._h, _pd(kk) if kk is not None else None, _pd(bb) if bb is not None else None), "small_set")

    def solve(self):
        """factor + solve every instance (one launch); returns the launch duration in milliseconds"""
        ms = C.c_double(0.0)
        self._check(self._L.calipso_hip_small_solve(self._h, C.byref(ms)), "small_solve")
        return float(ms.value)

    def get(self):
        """(X (batch, n, nrhs), inertia (batch, 3), number of instances that met an exact zero pivot)"""
        x = np.zeros(self.batch * self.n * self.nrhs)
        inr = np.zeros(3 * self.batch, dtype=np.int64)
        bad = self._check(self._L.calipso_hip_small_get(self._h, _pd(x), _pi(inr)), "small_get")
        return np.transpose(x.reshape(self.batch, self.nrhs, self.n), (0, 2, 1)), inr.reshape(self.batch, 3), bad

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._L.calipso_hip_small_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class SmallNewtonBatch:
    """Solver / initialize! / solve! (src/solver/solver.jl:46-150, initialize.jl:9-48, solve.jl:8-377) for `batch` independent small conic QPs of one shape in ONE
    kernel launch: a workgroup per instance, everything in LDS, every decision of solve! on the device (include/calipso_hip.h, "solve! for a batch of SMALL conic QPs";
    csrc/smallnewton.hip).  QP data as qp_attach: min c x'Px + q'x s.t. Ax = b, h - Gx >= 0 (nonnegative cones)."""

    def __init__(self, nx, ne, nc, batch, device=0, options=None):
        self._L = lib()
        self.nx, self.ne, self.nc, self.batch = int(nx), int(ne), int(nc), int(batch)
        self.N = self.nx + 2 * self.ne + 3 * self.nc
        h = C.c_void_p()
        rc = self._L.calipso_hip_smallnewton_create(self.nx, self.ne, self.nc, self.batch, device, C.byref(h))
        if rc != 0:
            msg = self._L.calipso_hip_smallnewton_last_error(h if h.value else None).decode()
            if h.value:
                self._L.calipso_hip_smallnewton_destroy(h)
            raise CalipsoHipError("calipso_hip_smallnewton_create failed (%d): %s" % (rc, msg))
        self._h = h
        for k, v in (options or {}).items():
            self.set_option(k, v)

    def _check(self, rc, what):
        if rc < 0:
            raise CalipsoHipError("%s failed (%d): %s" % (what, rc, self._L.calipso_hip_smallnewton_last_error(self._h).decode()))
        return rc

    def set_option(self, name, value):
        self._check(self._L.calipso_hip_smallnewton_set_option(self._h, name.encode(), float(value)), "smallnewton_set_option(%s)" % name)

    def set_cones(self, n_nonnegative, second_order_dims=()):
        """cone layout: the first n_nonnegative cone entries nonnegative, then second-order cones of the given dimensions (contiguous)"""
        dims = np.ascontiguousarray(list(second_order_dims), dtype=np.int64)
        self._check(self._L.calipso_hip_smallnewton_set_cones(self._h, int(n_nonnegative), int(dims.size), _pi(dims) if dims.size else None), "smallnewton_set_cones")

    def set_qp(self, P, q, A, b, G, h, objective_scale=0.5, shared=None):
        """arrays of ONE problem (P (nx, nx), q (nx), A (ne, nx), ...) shared by all instances, or stacked along a leading batch axis"""
        P = np.asarray(P, dtype=np.float64)
        if shared is None:
            shared = P.ndim == 2
        K = 1 if shared else self.batch
        cm = lambda M, r, c: np.ascontiguousarray(np.transpose(np.asarray(M, dtype=np.float64).reshape(K, r, c), (0, 2, 1))).reshape(-1) if r * c else np.zeros(1)
        vv = lambda a, n: np.ascontiguousarray(np.asarray(a, dtype=np.float64).reshape(K, n)).reshape(-1) if n else np.zeros(1)
        Pc, Ac, Gc = cm(P, self.nx, self.nx), cm(A, self.ne, self.nx), cm(G, self.nc, self.nx)
        qv, bv, hv = vv(q, self.nx), vv(b, self.ne), vv(h, self.nc)
        self._check(self._L.calipso_hip_smallnewton_set_qp(self._h, _pd(Pc), _pd(qv), _pd(Ac), _pd(bv), _pd(Gc), _pd(hv), float(objective_scale), int(bool(shared))), "smallnewton_set_qp")

    def set_state(self, w=None, dual=None, scalars=None):
        """w: (batch, N) points; dual: (batch, ne) multiplier estimates; scalars: (batch, 3) [central_path, fraction_to_boundary, penalty]"""
        f = lambda a, n: None if a is None else np.ascontiguousarray(np.asarray(a, dtype=np.float64).reshape(self.batch, n)).reshape(-1)
        ww, ll, ss = f(w, self.N), f(dual, max(self.ne, 1)) if (dual is not None and self.ne) else None, f(scalars, 3)
        self._check(self._L.calipso_hip_smallnewton_set_state(self._h, _pd(ww) if ww is not None else None, _pd(ll) if ll is not None else None, _pd(ss) if ss is not None else None), "smallnewton_set_state")

    def initialize(self, x0):
        """initialize!(solver, guess) for every instance: x0 (batch, nx) or one guess for all"""
        x0 = np.asarray(x0, dtype=np.float64)
        w = np.zeros((self.batch, self.N))
        w[:, :self.nx] = x0.reshape(-1, self.nx)
        self.set_state(w=w)

    def get_state(self):
        w = np.zeros(self.batch * self.N); lam = np.zeros(self.batch * max(self.ne, 1)); sc = np.zeros(self.batch * 6); cn = np.zeros(self.batch * 8, dtype=np.int64)
        self._check(self._L.calipso_hip_smallnewton_get_state(self._h, _pd(w), _pd(lam), _pd(sc), _pi(cn)), "smallnewton_get_state")
        names = ("total_iterations", "outer", "factorizations", "refinement_failures", "max_refinement_rounds", "last_refinement_rounds", "newton_steps", "accepted_iterates")
        return dict(solution=w.reshape(self.batch, self.N), dual=lam.reshape(self.batch, -1)[:, :self.ne], scalars=sc.reshape(self.batch, 6),
                    counters={n: cn.reshape(self.batch, 8)[:, i].copy() for i, n in enumerate(names)})

    def keep_trace(self, rows):
        self._trace_rows = int(rows)
        self._check(self._L.calipso_hip_smallnewton_trace(self._h, int(rows), None), "smallnewton_trace")

    def trace(self, rows=None):
        rows = int(rows if rows is not None else self._trace_rows)
        out = np.zeros(self.batch * rows * self.N)
        self._check(self._L.calipso_hip_smallnewton_trace(self._h, rows, _pd(out)), "smallnewton_trace")
        return out.reshape(self.batch, rows, self.N)

    def solve(self):
        """solve! of every instance, one launch: (result (batch,) int32 — 1 converged, 0 caps reached, < 0 see include/calipso_hip.h —, launch milliseconds)"""
        res = np.zeros(self.batch, dtype=np.int32); ms = C.c_double(0.0)
        self._check(self._L.calipso_hip_smallnewton_solve(self._h, res.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(ms)), "smallnewton_solve")
        return res, float(ms.value)

    def steps(self, count, advance=False):
        """`count` Newton steps of every instance in one launch: (info (batch, 8), status (batch,), launch milliseconds)"""
        info = np.zeros(self.batch * 8); st = np.zeros(self.batch, dtype=np.int32); ms = C.c_double(0.0)
        self._check(self._L.calipso_hip_smallnewton_steps(self._h, int(count), int(bool(advance)), _pd(info), st.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(ms)), "smallnewton_steps")
        return info.reshape(self.batch, 8), st, float(ms.value)

    def differentiate(self, jacobian_parameters):
        """differentiate! of every instance at its resident point, one launch (differentiate.jl:1-61): jacobian_parameters (batch, N, p) = dR/dtheta per instance, or
        (N, p) = one matrix for all instances; returns (sensitivity (batch, N, p) = dw/dtheta, status (batch,) — 1: the factorisation's inertia is not
        (nx, ne + nc, 0) —, launch milliseconds)"""
        J = np.asarray(jacobian_parameters, dtype=np.float64)
        N = self.nx + 2 * self.ne + 3 * self.nc
        shared = J.ndim == 2
        if shared:
            J = J[None]
        if J.ndim != 3 or J.shape[0] != (1 if shared else self.batch) or J.shape[1] != N:
            raise ValueError("jacobian_parameters must be (batch, N, p) or (N, p)")
        p = J.shape[2]
        Jc = np.ascontiguousarray(np.transpose(J, (0, 2, 1))).ravel()          # per instance column-major N x p
        out = np.zeros(self.batch * N * p); st = np.zeros(self.batch, dtype=np.int32); ms = C.c_double(0.0)
        self._check(self._L.calipso_hip_smallnewton_differentiate(self._h, int(p), int(shared), _pd(Jc), _pd(out), st.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(ms)), "smallnewton_differentiate")
        return np.transpose(out.reshape(self.batch, p, N), (0, 2, 1)).copy(), st, float(ms.value)

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._L.calipso_hip_smallnewton_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Comm:
    """RCCL communicator of the batched path (include/calipso_hip.h, "multi-GPU exchange"): one process per GPU; the only
    collectives are the post-round all-gather of per-problem status rows and the all-reduce of counters."""

    @staticmethod
    def unique_id():
        buf = (C.c_uint8 * 128)()
        rc = lib().calipso_hip_comm_unique_id(buf)
        if rc != 0:
            raise CalipsoHipError("comm_unique_id failed (%d): %s" % (rc, lib().calipso_hip_comm_last_error(None).decode()))
        return bytes(buf)

    def __init__(self, rank, nranks, unique_id, device=0):
        self._L = lib()
        self.rank, self.nranks = int(rank), int(nranks)
        buf = (C.c_uint8 * 128)(*unique_id)
        h = C.c_void_p()
        rc = self._L.calipso_hip_comm_init(self.rank, self.nranks, buf, device, C.byref(h))
        if rc != 0:
            raise CalipsoHipError("comm_init failed (%d): %s" % (rc, self._L.calipso_hip_comm_last_error(h if h.value else None).decode()))
        self._c = h

    def size(self):
        """(ranks, own rank) as the communicator itself reports them (ncclCommCount / ncclCommUserRank)"""
        out = np.zeros(2, dtype=np.int32)
        rc = self._L.calipso_hip_comm_size(self._c, out.ctypes.data_as(C.POINTER(C.c_int32)))
        if rc < 0:
            raise CalipsoHipError("comm_size failed (%d): %s" % (rc, self._L.calipso_hip_comm_last_error(self._c).decode()))
        return int(out[0]), int(out[1])

    def gather_status(self, rows, capacity):
        rows = np.ascontiguousarray(rows, dtype=np.int32).reshape(-1, 4)
        out = np.zeros((int(capacity), 4), dtype=np.int32)
        counts = np.zeros(self.nranks, dtype=np.int64)
        p32 = lambda a: a.ctypes.data_as(C.POINTER(C.c_int32))
        n = self._L.calipso_hip_comm_gather_status(self._c, p32(rows), rows.shape[0], p32(out), out.shape[0], _pi(counts))
        if n < 0:
            raise CalipsoHipError("comm_gather_status failed (%d): %s" % (n, self._L.calipso_hip_comm_last_error(self._c).decode()))
        return out[:n], counts

    def allreduce_sum(self, values):
        v = np.ascontiguousarray(values, dtype=np.float64).copy()
        rc = self._L.calipso_hip_comm_allreduce_sum(self._c, _pd(v), v.size)
        if rc < 0:
            raise CalipsoHipError("comm_allreduce_sum failed (%d): %s" % (rc, self._L.calipso_hip_comm_last_error(self._c).decode()))
        return v

    def close(self):
        if getattr(self, "_c", None) is not None and self._c.value:
            self._L.calipso_hip_comm_destroy(self._c)
            self._c = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def mfma_f64_peak(device=0):
    """measured fp64 matrix-core ceiling of the device in TFLOP/s (calipso_hip_mfma_f64_peak)"""
    t = C.c_double(0.0)
    rc = lib().calipso_hip_mfma_f64_peak(device, C.byref(t))
    if rc != 0:
        raise CalipsoHipError("mfma_f64_peak failed (%d)" % rc)
    return float(t.value)


def initialize_b(solver, guess):
    """initialize!(solver, guess)  src/solver/initialize.jl:9-13"""
    g = np.ascontiguousarray(guess, dtype=np.float64)
    solver._check(solver._L.calipso_hip_initialize(solver._h, _pd(g)), "initialize!")


def solve_b(solver):
    """solve!(solver)::Bool  src/solver/solve.jl:8-377 (raises on the reference's error() cases)"""
    attached = getattr(solver, "_qp_attached", False)
    rc = solver._L.calipso_hip_solve(solver._h, solver._cb, None)
    solver._check(rc, "solve!")
    return rc == 1
