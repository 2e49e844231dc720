"""Problem definitions used by the parity tests (test infrastructure).

Each problem exposes `evaluate(flags, x, y, z, theta, out)`, the stand-in for the reference's
Symbolics-generated functions called by `evaluate!` (src/solver/evaluate.jl:1-124): it writes the
requested ProblemData fields (reference field names, column-major) through `out(name)`.
The flag bits are the same for the oracle (oracle/calipso_oracle.h) and for the product
(include/calipso_hip.h).

Problem data mirrors the reference's own tests (file:line given per constructor).
"""
import numpy as np

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
ALL_VARIABLE_FLAGS = (1 << 11) - 1


def _put(out, name, arr):
    buf = out(name)
    a = np.asarray(arr, dtype=np.float64)
    if a.ndim == 2:
        a = a.T.reshape(-1)  # column-major
    if buf.size:
        buf[:] = a.reshape(-1)


class SymbolicProblem:
    """f, g, h given as sympy expressions in x (and theta); derivatives by sympy, like the reference's codegen
    (src/solver/codegen.jl:1-90 — out of the hot path; only used to feed the tests)."""

    def __init__(self, nx, objective, equality=None, cone=None, np_=0, nonnegative_indices=None,
                 second_order_indices=None, parameters=None, x0=None, name="symbolic"):
        import sympy as sp
        self.name = name
        self.nx, self.np = nx, np_
        x = sp.symbols("x0:%d" % nx) if nx else ()
        th = sp.symbols("p0:%d" % np_) if np_ else ()
        f = sp.sympify(objective(list(x), list(th)) if np_ else objective(list(x)))
        g = list(equality(list(x), list(th)) if (equality and np_) else (equality(list(x)) if equality else []))
        h = list(cone(list(x), list(th)) if (cone and np_) else (cone(list(x)) if cone else []))
        self.ne, self.nc = len(g), len(h)
        y = sp.symbols("y0:%d" % self.ne) if self.ne else ()
        z = sp.symbols("z0:%d" % self.nc) if self.nc else ()
        X = sp.Matrix(list(x))
        TH = sp.Matrix(list(th)) if np_ else None
        gm = sp.Matrix(g) if g else sp.zeros(0, 1)
        hm = sp.Matrix(h) if h else sp.zeros(0, 1)
        fx = sp.Matrix([f]).jacobian(X).T
        gx = gm.jacobian(X) if g else sp.zeros(0, nx)
        hx = hm.jacobian(X) if h else sp.zeros(0, nx)
        gy = (gm.T * sp.Matrix(list(y)))[0] if g else sp.Integer(0)
        hz = (hm.T * sp.Matrix(list(z)))[0] if h else sp.Integer(0)
        gyx = sp.Matrix([gy]).jacobian(X).T
        hzx = sp.Matrix([hz]).jacobian(X).T
        args = [list(x), list(y), list(z), list(th)]
        lam = lambda e: sp.lambdify(args, e, "numpy")
        self._f = lam(f)
        self._fx = lam(fx)
        self._fxx = lam(fx.jacobian(X))
        self._g = lam(gm)
        self._gx = lam(gx)
        self._gyx = lam(gyx)
        self._gyxx = lam(gyx.jacobian(X))
        self._h = lam(hm)
        self._hx = lam(hx)
        self._hzx = lam(hzx)
        self._hzxx = lam(hzx.jacobian(X))
        if np_:
            self._fxp = lam(fx.jacobian(TH))
            self._gp = lam(gm.jacobian(TH)) if g else None
            self._gyxp = lam(gyx.jacobian(TH))
            self._hp = lam(hm.jacobian(TH)) if h else None
            self._hzxp = lam(hzx.jacobian(TH))
        self.nonnegative_indices = list(range(1, self.nc + 1)) if nonnegative_indices is None else list(nonnegative_indices)
        self.second_order_indices = [[]] if second_order_indices is None else [list(c) for c in second_order_indices]
        self.parameters = np.zeros(np_) if parameters is None else np.asarray(parameters, dtype=np.float64)
        self.x0 = np.zeros(nx) if x0 is None else np.asarray(x0, dtype=np.float64)

    def evaluate(self, flags, x, y, z, theta, out):
        a = (list(x), list(y), list(z), list(theta))
        m = lambda fn, r, c: np.asarray(fn(*a), dtype=np.float64).reshape(r, c) if r * c else np.zeros((r, c))
        nx, ne, nc, npar = self.nx, self.ne, self.nc, self.np
        if flags & OBJECTIVE:
            _put(out, "objective", [float(self._f(*a))])
        if flags & OBJECTIVE_GRADIENT:
            _put(out, "objective_gradient_variables", m(self._fx, nx, 1))
        if flags & OBJECTIVE_HESSIAN:
            _put(out, "objective_jacobian_variables_variables", m(self._fxx, nx, nx))
        if flags & EQUALITY and ne:
            _put(out, "equality_constraint", m(self._g, ne, 1))
        if flags & EQUALITY_JACOBIAN and ne:
            _put(out, "equality_jacobian_variables", m(self._gx, ne, nx))
        if flags & EQUALITY_DUAL_GRADIENT:
            _put(out, "equality_dual_jacobian_variables", m(self._gyx, nx, 1))
        if flags & EQUALITY_DUAL_HESSIAN:
            _put(out, "equality_dual_jacobian_variables_variables", m(self._gyxx, nx, nx))
        if flags & CONE and nc:
            _put(out, "cone_constraint", m(self._h, nc, 1))
        if flags & CONE_JACOBIAN and nc:
            _put(out, "cone_jacobian_variables", m(self._hx, nc, nx))
        if flags & CONE_DUAL_GRADIENT:
            _put(out, "cone_dual_jacobian_variables", m(self._hzx, nx, 1))
        if flags & CONE_DUAL_HESSIAN:
            _put(out, "cone_dual_jacobian_variables_variables", m(self._hzxx, nx, nx))
        if npar:
            if flags & OBJECTIVE_JACOBIAN_PARAMETERS:
                _put(out, "objective_jacobian_variables_parameters", m(self._fxp, nx, npar))
            if flags & EQUALITY_JACOBIAN_PARAMETERS and ne:
                _put(out, "equality_jacobian_parameters", m(self._gp, ne, npar))
            if flags & EQUALITY_DUAL_JACOBIAN_PARAMETERS:
                _put(out, "equality_dual_jacobian_variables_parameters", m(self._gyxp, nx, npar))
            if flags & CONE_JACOBIAN_PARAMETERS and nc:
                _put(out, "cone_jacobian_parameters", m(self._hp, nc, npar))
            if flags & CONE_DUAL_JACOBIAN_PARAMETERS:
                _put(out, "cone_dual_jacobian_variables_parameters", m(self._hzxp, nx, npar))


class ConicQP:
    """min 1/2 x'Px + q'x  s.t.  Ax - b = 0,  h - Gx in K   (numpy, no sympy).
    Same structure as generate_random_qp of test/solver/problem.jl:3-23 (whose objective is x'Px + q'x:
    pass objective_scale=1.0 to reproduce that form)."""

    def __init__(self, P, q, A, b, G, h, nonnegative_indices=None, second_order_indices=None, x0=None,
                 objective_scale=0.5, name="qp"):
        self.name = name
        self.P, self.q, self.A, self.b, self.G, self.h = [np.asarray(v, dtype=np.float64) for v in (P, q, A, b, G, h)]
        self.nx, self.ne, self.nc, self.np = self.P.shape[0], self.A.shape[0], self.G.shape[0], 0
        self.c = objective_scale
        self.Psym = self.c * (self.P + self.P.T)
        self.nonnegative_indices = list(range(1, self.nc + 1)) if nonnegative_indices is None else list(nonnegative_indices)
        self.second_order_indices = [[]] if second_order_indices is None else [list(c) for c in second_order_indices]
        self.parameters = np.zeros(0)
        self.x0 = np.zeros(self.nx) if x0 is None else np.asarray(x0, dtype=np.float64)

    def evaluate(self, flags, x, y, z, theta, out):
        x = np.asarray(x); y = np.asarray(y); z = np.asarray(z)
        if flags & OBJECTIVE:
            _put(out, "objective", [self.c * x @ self.P @ x + self.q @ x])
        if flags & OBJECTIVE_GRADIENT:
            _put(out, "objective_gradient_variables", self.Psym @ x + self.q)
        if flags & OBJECTIVE_HESSIAN:
            _put(out, "objective_jacobian_variables_variables", self.Psym)
        if flags & EQUALITY and self.ne:
            _put(out, "equality_constraint", self.A @ x - self.b)
        if flags & EQUALITY_JACOBIAN and self.ne:
            _put(out, "equality_jacobian_variables", self.A)
        if flags & EQUALITY_DUAL_GRADIENT:
            _put(out, "equality_dual_jacobian_variables", self.A.T @ y if self.ne else np.zeros(self.nx))
        if flags & EQUALITY_DUAL_HESSIAN:
            _put(out, "equality_dual_jacobian_variables_variables", np.zeros((self.nx, self.nx)))
        if flags & CONE and self.nc:
            _put(out, "cone_constraint", self.h - self.G @ x)
        if flags & CONE_JACOBIAN and self.nc:
            _put(out, "cone_jacobian_variables", -self.G)
        if flags & CONE_DUAL_GRADIENT:
            _put(out, "cone_dual_jacobian_variables", -self.G.T @ z if self.nc else np.zeros(self.nx))
        if flags & CONE_DUAL_HESSIAN:
            _put(out, "cone_dual_jacobian_variables_variables", np.zeros((self.nx, self.nx)))


class ParametricConicQP(ConicQP):
    """ConicQP whose linear data moves with parameters  theta = [dq; db; dh]  (np = nx + ne + nc):
        min 1/2 x'Px + (q + dq)'x   s.t.  Ax - (b + db) = 0,   (h + dh) - Gx in K
    so every block of dR/dtheta (residual_jacobian_parameters.jl:1-40) is populated and differentiate! has many right-hand sides."""

    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self.np = self.nx + self.ne + self.nc
        self.parameters = np.zeros(self.np)

    def evaluate(self, flags, x, y, z, theta, out):
        theta = np.asarray(theta, dtype=np.float64)
        nx, ne, nc, npar = self.nx, self.ne, self.nc, self.np
        dq, db, dh = theta[:nx], theta[nx:nx + ne], theta[nx + ne:]
        q0, b0, h0 = self.q, self.b, self.h
        self.q, self.b, self.h = q0 + dq, b0 + db, h0 + dh
        try:
            super().evaluate(flags, x, y, z, theta, out)
        finally:
            self.q, self.b, self.h = q0, b0, h0
        if flags & OBJECTIVE_JACOBIAN_PARAMETERS:
            J = np.zeros((nx, npar)); J[:, :nx] = np.eye(nx)
            _put(out, "objective_jacobian_variables_parameters", J)
        if flags & EQUALITY_JACOBIAN_PARAMETERS and ne:
            J = np.zeros((ne, npar)); J[:, nx:nx + ne] = -np.eye(ne)
            _put(out, "equality_jacobian_parameters", J)
        if flags & EQUALITY_DUAL_JACOBIAN_PARAMETERS:
            _put(out, "equality_dual_jacobian_variables_parameters", np.zeros((nx, npar)))
        if flags & CONE_JACOBIAN_PARAMETERS and nc:
            J = np.zeros((nc, npar)); J[:, nx + ne:] = np.eye(nc)
            _put(out, "cone_jacobian_parameters", J)
        if flags & CONE_DUAL_JACOBIAN_PARAMETERS:
            _put(out, "cone_dual_jacobian_variables_parameters", np.zeros((nx, npar)))


def parametric_conic_qp(nx, ne, n_nn, n_soc, soc_dim, seed=0):
    rng = np.random.default_rng(seed)
    nc = n_nn + n_soc * soc_dim
    Q = rng.standard_normal((nx, nx)) / np.sqrt(nx)
    P = Q.T @ Q + np.eye(nx)
    A = rng.standard_normal((ne, nx)) / np.sqrt(nx)
    G = rng.standard_normal((nc, nx)) / np.sqrt(nx)
    nn = list(range(1, n_nn + 1))
    soc = [list(range(n_nn + 1 + j * soc_dim, n_nn + 1 + (j + 1) * soc_dim)) for j in range(n_soc)] or [[]]
    return ParametricConicQP(P, rng.standard_normal(nx), A, rng.standard_normal(ne), G, rng.random(nc) + 1.0, nonnegative_indices=nn,
                             second_order_indices=soc, name="parametric_conic_qp")


# ---- the reference's test problems ---------------------------------------------------------------
def wachter():
    """README.md:97-121 = test/solver/wachter.jl:3-15; x* = [1, 0, 0.5] (wachter.jl:47).  BASELINE config C1."""
    return SymbolicProblem(3, lambda x: x[0], lambda x: [x[0] ** 2 - x[1] - 1.0, x[0] - x[2] - 0.5],
                           lambda x: [x[1], x[2]], x0=[-2.0, 3.0, 1.0], name="wachter")


def maratos():
    """test/solver/maratos.jl:3-15"""
    return SymbolicProblem(2, lambda x: 2.0 * (x[0] ** 2 + x[1] ** 2 - 1.0) - x[0], lambda x: [x[0] ** 2 + x[1] ** 2 - 1.0],
                           None, x0=[2.0, 1.0], name="maratos")


def test1():
    """test/solver/test1.jl:3-15"""
    return SymbolicProblem(50, lambda x: sum(v * v for v in x), lambda x: [x[i] ** 2 - 1.2 for i in range(30)],
                           lambda x: [x[0] + 10.0, x[1] + 5.0, 20.0 - x[4]], x0=np.ones(50), name="test1")


def test2(x0):
    """test/solver/test2.jl:3-15 (x0 = rand(2) in the reference)"""
    import sympy as sp
    return SymbolicProblem(2, lambda x: -x[0] * x[1] + 2.0 / (3.0 * sp.sqrt(3)), None,
                           lambda x: [-x[0] - x[1] ** 2 + 1.0, x[0] + x[1]], x0=x0, name="test2")


def test3(x0):
    """test/solver/test3.jl:4-16"""
    return SymbolicProblem(2, lambda x: 100.0 * (x[1] - x[0] ** 2) ** 2 + (1.0 - x[0]) ** 2, None,
                           lambda x: [-(x[0] - 1.0) ** 3 + x[1] - 1.0, -x[0] - x[1] + 2.0], x0=x0, name="test3")


def test4(x0):
    """test/solver/test4.jl:3-14"""
    import sympy as sp
    return SymbolicProblem(3, lambda x: x[0] - 2.0 * x[1] + x[2] + sp.sqrt(6), None,
                           lambda x: [1 - x[0] ** 2 - x[1] ** 2 - x[2] ** 2], x0=x0, name="test4")


def knitro():
    """test/solver/knitro.jl:3-20 (MPCC); commented asserts :39-44 give x = [1,0,2,0,0,0,3,6]"""
    return SymbolicProblem(8, lambda x: (x[0] - 5) ** 2 + (2 * x[1] + 1) ** 2,
                           lambda x: [2 * (x[1] - 1) - 1.5 * x[1] + x[2] - 0.5 * x[3] + x[4],
                                      3 * x[0] - x[1] - 3.0 - x[5], -x[0] + 0.5 * x[1] + 4.0 - x[6],
                                      -x[0] - x[1] + 7.0 - x[7], x[2] * x[5], x[3] * x[6], x[4] * x[7]],
                           lambda x: list(x), x0=np.zeros(8), name="knitro")


def friction_cone(v, mu, gamma, x0):
    """test/solver/friction_cone.jl:13-41: one SOC of dimension 3, no nonnegative cones"""
    return SymbolicProblem(3, lambda x: v[0] * x[0] + v[1] * x[1] + v[2] * x[2], lambda x: [x[0] - mu * gamma],
                           lambda x: list(x), nonnegative_indices=[], second_order_indices=[[1, 2, 3]], x0=x0, name="friction")


def portfolio(seed=0, p=10):
    """test/solver/portfolio.jl:6-44: 2 nonnegative + one SOC of dimension p+2"""
    rng = np.random.default_rng(seed)
    E = rng.standard_normal((p, p))
    Sig = E.T @ E
    w, V = np.linalg.eigh(Sig)
    Sh = (V * np.sqrt(w)) @ V.T
    c = np.concatenate([np.zeros(p), [1.0]])
    G1 = np.block([[2.0 * Sh, np.zeros((p, 1))], [np.zeros((1, p)), -np.ones((1, 1))]])
    hvec = np.concatenate([np.zeros(p), [1.0]])
    qv = np.concatenate([np.zeros(p), [1.0]])
    G2 = np.concatenate([np.ones(p), [0.0]])[None, :]
    G3 = np.concatenate([-np.ones(p), [0.0]])[None, :]
    A = np.vstack([G2, G3, -qv[None, :], -G1])
    b = np.concatenate([[1.0, -1.0, 1.0], hvec])
    nx = p + 1
    prob = ConicQP(np.zeros((nx, nx)), c, np.zeros((0, nx)), np.zeros(0), A, b,
                   nonnegative_indices=[1, 2], second_order_indices=[list(range(3, 3 + p + 2))],
                   x0=rng.standard_normal(nx), name="portfolio")
    prob.A_cone, prob.b_cone = A, b
    return prob


def random_qp(nx=10, ne=5, nc=5, seed=0, second_order_indices=None, nonnegative_indices=None):
    """generate_random_qp of test/solver/problem.jl:3-23 (objective z'Pz + q'z, P = B'B)."""
    rng = np.random.default_rng(seed)
    B = rng.standard_normal((nx, nx))
    P = B.T @ B
    q = rng.standard_normal(nx)
    G = rng.standard_normal((nc, nx))
    xb = rng.standard_normal(nx)
    h = G @ xb + rng.random(nc)
    A = rng.standard_normal((ne, nx))
    b = A @ xb
    return ConicQP(P, q, A, b, G, h, nonnegative_indices=nonnegative_indices, second_order_indices=second_order_indices,
                   x0=rng.standard_normal(nx), objective_scale=1.0, name="random_qp")


def qp_equality_parametric(seed=0, nx=10, ne=5):
    """test/solver/qp_equality.jl:2-33: theta = [diag(P); p; vec(A); b]"""
    rng = np.random.default_rng(seed)
    xh = np.maximum(0.0, rng.standard_normal(nx))
    Q = rng.random((nx, nx))
    Pd = np.diag(Q.T @ Q).copy()
    p = rng.standard_normal(nx)
    A = rng.random((ne, nx))
    b = A @ xh
    theta = np.concatenate([Pd, p, A.T.reshape(-1), b])   # vec(A) is column-major
    npar = theta.size

    def obj(x, th):
        return sum(0.5 * th[i] * x[i] ** 2 for i in range(nx)) + sum(th[nx + i] * x[i] for i in range(nx))

    def eq(x, th):
        return [sum(th[2 * nx + i + j * ne] * x[j] for j in range(nx)) - th[2 * nx + ne * nx + i] for i in range(ne)]

    prob = SymbolicProblem(nx, obj, eq, None, np_=npar, parameters=theta, x0=rng.standard_normal(nx), name="qp_equality")
    prob.Pd, prob.p, prob.A, prob.b = Pd, p, A, b
    return prob


def double_integrator(horizon=5, action_guess=None):
    """test/examples/double_integrator.jl:1-90 written in standard form: z = [X1;U1;...;X_{T-1};U_{T-1};X_T] (X in R^2, U in R), equality rows
    [dynamics d_1..d_{T-1}; X1 - x_init; X_T - x_goal] (src/trajectory_optimization/indices.jl:63-80), parameters theta = [theta_1; theta_t...;
    theta_T] with theta_1 = [vec(A); B; diag(Q); R; x_init] (11), theta_t = [vec(A); B; diag(Q); R] (9), theta_T = [diag(Q_T); x_goal] (4).
    The reference draws the action guess with randn (:81); fix it."""
    T = horizon
    nz = 2 * T + (T - 1)
    A = np.array([[1.0, 1.0], [0.0, 1.0]]); B = np.array([0.0, 1.0])
    th1 = np.concatenate([A.T.reshape(-1), B, [1.0, 1.0], [0.1], [0.0, 0.0]])
    tht = np.concatenate([A.T.reshape(-1), B, [1.0, 1.0], [0.1]])
    thT = np.array([10.0, 10.0, 1.0, 0.0])
    theta = np.concatenate([th1] + [tht] * (T - 2) + [thT])
    off = [0]
    for t in range(T):
        off.append(off[-1] + (11 if t == 0 else (9 if t < T - 1 else 4)))

    def X(z, t):
        return z[3 * t: 3 * t + 2]

    def U(z, t):
        return z[3 * t + 2: 3 * t + 3]

    def W(th, t):
        return th[off[t]: off[t + 1]]

    def objective(z, th):
        J = 0
        for t in range(T - 1):
            w = W(th, t)
            J += 0.5 * (w[6] * X(z, t)[0] ** 2 + w[7] * X(z, t)[1] ** 2) + 0.5 * w[8] * U(z, t)[0] ** 2
        w = W(th, T - 1)
        return J + 0.5 * (w[0] * X(z, T - 1)[0] ** 2 + w[1] * X(z, T - 1)[1] ** 2)

    def equality(z, th):
        e = []
        for t in range(T - 1):
            w = W(th, t)                       # A = reshape(w[1:4], 2, 2) column-major, B = w[5:6]
            x, u, y = X(z, t), U(z, t), X(z, t + 1)
            e += [y[0] - (w[0] * x[0] + w[2] * x[1] + w[4] * u[0]), y[1] - (w[1] * x[0] + w[3] * x[1] + w[5] * u[0])]
        w1, wT = W(th, 0), W(th, T - 1)
        e += [X(z, 0)[0] - w1[9], X(z, 0)[1] - w1[10], X(z, T - 1)[0] - wT[2], X(z, T - 1)[1] - wT[3]]
        return e

    x0 = np.zeros(nz)
    for t in range(T):                         # linear_interpolation(state_initial, state_goal, horizon)
        x0[3 * t: 3 * t + 2] = np.array([1.0, 0.0]) * t / (T - 1)
    if action_guess is not None:
        for t in range(T - 1):
            x0[3 * t + 2] = action_guess[t]
    prob = SymbolicProblem(nz, objective, equality, None, np_=theta.size, parameters=theta, x0=x0, name="double_integrator")
    prob.horizon = T
    return prob


def qp_nonnegative_parametric(seed=0, nx=10, ne=5):
    """test/solver/qp_nonnegative.jl:2-49: the parametric QP of qp_equality.jl with the cone constraint x >= 0 (nc = nx nonnegative entries)"""
    base = qp_equality_parametric(seed=seed, nx=nx, ne=ne)
    theta = base.parameters

    def obj(x, th):
        return sum(0.5 * th[i] * x[i] ** 2 for i in range(nx)) + sum(th[nx + i] * x[i] for i in range(nx))

    def eq(x, th):
        return [sum(th[2 * nx + i + j * ne] * x[j] for j in range(nx)) - th[2 * nx + ne * nx + i] for i in range(ne)]

    def cone(x, th):
        return [x[i] for i in range(nx)]

    prob = SymbolicProblem(nx, obj, eq, cone, np_=theta.size, parameters=theta, x0=np.random.default_rng(seed + 100).standard_normal(nx),
                           name="qp_nonnegative")
    prob.Pd, prob.p, prob.A, prob.b = base.Pd, base.p, base.A, base.b
    return prob


class TrajectoryProblem(SymbolicProblem):
    """A SymbolicProblem whose equality-dual Hessian (g'y)xx is scattered the way the reference's trajectory layer does it
    (SURVEY.md quirk B-11): every dynamics stage t emits the structurally non-zero entries of the Hessian of y_t'd_t(X_t, U_t, X_t+1)
    at GLOBAL (i, j) tuples (src/trajectory_optimization/dynamics.jl:81-101,245-260), the per-stage lists are concatenated without
    de-duplication (methods.jl:24-27) and the core scatter ASSIGNS  problem.equality_dual_jacobian_variables_variables[idx...] =
    cache[i]  (src/solver/evaluate.jl:75-77): where consecutive stages share (X_t+1, X_t+1) entries the LAST writer wins (stage
    order 1..T-1).  hessian_mode = "sum" gives the exact Hessian instead (what the reference's own unit test checks with +=,
    test/trajectory_optimization/hessian_lagrangian.jl:297-303)."""

    def set_stages(self, stage_hessian, stage_slices, stage_dual_slices, hessian_mode):
        self._stage_hessian = stage_hessian            # (x, u, y, lam) -> local Hessian of lam'd wrt [x; u; y], and its structural pattern
        self._stage_slices = stage_slices              # per stage: global indices of [x; u; y]
        self._stage_dual_slices = stage_dual_slices    # per stage: slice of the equality duals of its dynamics rows
        self.hessian_mode = hessian_mode

    def evaluate(self, flags, x, y, z, theta, out):
        SymbolicProblem.evaluate(self, flags & ~EQUALITY_DUAL_HESSIAN, x, y, z, theta, out)
        if flags & EQUALITY_DUAL_HESSIAN:
            fn, pattern = self._stage_hessian
            H = np.zeros((self.nx, self.nx))
            xv, yv = np.asarray(x, dtype=np.float64), np.asarray(y, dtype=np.float64)
            for idx, dsl in zip(self._stage_slices, self._stage_dual_slices):
                loc = np.asarray(fn(*xv[idx], *yv[dsl]), dtype=np.float64).reshape(len(idx), len(idx))
                for (i, j) in pattern:
                    if self.hessian_mode == "sum":
                        H[idx[i], idx[j]] += loc[i, j]
                    else:
                        H[idx[i], idx[j]] = loc[i, j]       # last writer wins
            _put(out, "equality_dual_jacobian_variables_variables", H)


def pendulum(T=11, h=0.05, action_guess=None, hessian_mode="last_writer"):
    """README.md:123-189 = test/examples/pendulum.jl:3-59 written directly in standard form (BASELINE config C2).
    z = [X1;U1;...;X10;U10;X11]; equality = [d_1..d_10; X1 - 0; X11 - (pi,0)].
    hessian_mode = "last_writer" (default) reproduces the reference's scatter of the dynamics Hessians (quirk B-11, see
    TrajectoryProblem): at the shared (X_t+1, X_t+1) entries only the later stage's contribution survives; "sum" is the exact
    Lagrangian Hessian.  The converged answer is the same, the iterate path is not."""
    import sympy as sp
    nxs, nu = 2, 1
    nz = nxs * T + nu * (T - 1)
    m_, l_, g_, c_ = 1.0, 0.5, 9.81, 0.1

    def xs(z, t):
        return z[t * (nxs + nu): t * (nxs + nu) + nxs]

    def us(z, t):
        return z[t * (nxs + nu) + nxs: t * (nxs + nu) + nxs + nu]

    def f(mid, u):
        return [mid[1], u[0] / (m_ * l_ * l_) - g_ * sp.sin(mid[0]) / l_ - c_ * mid[1] / (m_ * l_ * l_)]

    def dyn(X, U, Y):
        mid = [0.5 * (X[0] + Y[0]), 0.5 * (X[1] + Y[1])]
        fm = f(mid, U)
        return [Y[0] - (X[0] + h * fm[0]), Y[1] - (X[1] + h * fm[1])]

    def objective(z):
        J = 0
        for t in range(T - 1):
            X, U = xs(z, t), us(z, t)
            J += 0.1 * (X[0] ** 2 + X[1] ** 2) + 0.1 * U[0] ** 2
        XT = xs(z, T - 1)
        return J + 0.1 * (XT[0] ** 2 + XT[1] ** 2)

    def equality(z):
        e = []
        for t in range(T - 1):
            e += dyn(xs(z, t), us(z, t), xs(z, t + 1))
        X1, XT = xs(z, 0), xs(z, T - 1)
        e += [X1[0] - 0.0, X1[1] - 0.0, XT[0] - sp.pi, XT[1] - 0.0]
        return e

    x0 = np.zeros(nz)
    for t in range(T):  # linear_interpolation(x1, xT, T)  (trajectory_optimization/utilities.jl:10-19)
        x0[t * (nxs + nu): t * (nxs + nu) + nxs] = np.array([np.pi, 0.0]) * t / (T - 1)
    if action_guess is not None:
        for t in range(T - 1):
            x0[t * (nxs + nu) + nxs] = action_guess[t]
    prob = TrajectoryProblem(nz, objective, equality, None, x0=x0, name="pendulum")
    prob.T = T
    # per-stage Hessian of lam'd(y, x, u) wrt [x; u; y] (dynamics.jl:81-101) and its structural non-zeros
    lx, lu, ly, ll = sp.symbols("a0:2"), sp.symbols("b0:1"), sp.symbols("c0:2"), sp.symbols("l0:2")
    d = dyn(list(lx), list(lu), list(ly))
    dl = sum(li * di for li, di in zip(ll, d))
    loc = list(lx) + list(lu) + list(ly)
    Hs = sp.hessian(dl, loc)
    pattern = [(i, j) for j in range(len(loc)) for i in range(len(loc)) if Hs[i, j] != 0]     # column-major, like findnz
    fn = sp.lambdify(loc + list(ll), Hs, "numpy")
    stage_idx = [list(range(t * (nxs + nu), t * (nxs + nu) + nxs + nu + nxs)) for t in range(T - 1)]
    stage_dual = [slice(nxs * t, nxs * t + nxs) for t in range(T - 1)]
    prob.set_stages((fn, pattern), stage_idx, stage_dual, hessian_mode)
    return prob


# ---- synthetic conic QP of SURVEY.md 8(d) (SplitMix64; bit-identical on every side) -----------------
STREAMS = dict(B=1, q=2, A=3, G=4, xbar=5, cone_point_tail=6, x=10, r=11, y=12, z=13, lam=14, s_nn=15, t_nn=16,
               s_tail=17, t_tail=18, hpos=19)


def synthetic_layout(nx, ne, n_nn, n_soc, soc_dim):
    nc = n_nn + n_soc * soc_dim
    nonneg = list(range(1, n_nn + 1))
    soc = [list(range(n_nn + k * soc_dim + 1, n_nn + (k + 1) * soc_dim + 1)) for k in range(n_soc)]
    if not soc:
        soc = [[]]
    return nc, nonneg, soc


def synthetic_conic_qp(uniform, problem_id, nx, ne, n_nn, n_soc, soc_dim):
    """SURVEY.md 8(d).  `uniform(problem_id, stream_id, lo, hi, count)` is the SplitMix64 stream function of the
    side under test (oracle or HIP library).  Returns (ConicQP, point dict, lam)."""
    nc, nonneg, soc = synthetic_layout(nx, ne, n_nn, n_soc, soc_dim)
    U = lambda name, lo, hi, cnt: uniform(problem_id, STREAMS[name], lo, hi, cnt)
    B = U("B", -1, 1, nx * nx).reshape(nx, nx).T            # column-major fill
    P = (B + B.T) / (2.0 * np.sqrt(nx)) + 2.0 * np.eye(nx)
    q = U("q", -1, 1, nx)
    A = (U("A", -1, 1, ne * nx) / np.sqrt(nx)).reshape(nx, ne).T
    G = (U("G", -1, 1, nc * nx) / np.sqrt(nx)).reshape(nx, nc).T
    xbar = U("xbar", -1, 1, nx)
    b = A @ xbar
    cp = np.zeros(nc)                                        # an interior cone point
    cp[:n_nn] = U("hpos", 0.5, 1.5, n_nn)
    tails = U("cone_point_tail", -0.3, 0.3, nc)
    for c in soc:
        if c:
            idx = np.array(c) - 1
            cp[idx[1:]] = tails[idx[1:]]
            cp[idx[0]] = 1.0 + np.linalg.norm(cp[idx[1:]])
    h = G @ xbar + cp
    prob = ConicQP(P, q, A, b, G, h, nonnegative_indices=nonneg, second_order_indices=soc, name="synthetic")
    pt = dict(x=U("x", -1, 1, nx), r=0.1 * U("r", -1, 1, ne), y=U("y", -1, 1, ne), z=U("z", -1, 1, nc))
    s = np.zeros(nc); t = np.zeros(nc)
    s[:n_nn] = U("s_nn", 0.5, 1.5, n_nn); t[:n_nn] = U("t_nn", 0.5, 1.5, n_nn)
    st = U("s_tail", -0.3, 0.3, nc); tt = U("t_tail", -0.3, 0.3, nc)
    for c in soc:
        if c:
            idx = np.array(c) - 1
            s[idx[1:]] = st[idx[1:]]; t[idx[1:]] = tt[idx[1:]]
            s[idx[0]] = 1.0 + np.linalg.norm(s[idx[1:]]); t[idx[0]] = 1.0 + np.linalg.norm(t[idx[1:]])
    pt["s"], pt["t"] = s, t
    lam = U("lam", -1, 1, ne)
    return prob, pt, lam


def staged_conic_qp(uniform, problem_id, T, nv, nd, n_nn_stage, n_soc_stage, soc_dim):
    """The synthetic conic QP of SURVEY.md 8(d) with the stage structure of a trajectory-optimisation problem (the reference's
    src/trajectory_optimization layer: variables ordered stage by stage, indices.jl:41-180): T stages of nv variables; the
    Hessian block (stage, stage) only; nd "dynamics" equality rows per stage pair (t, t+1); per stage n_nn_stage nonnegative rows
    and n_soc_stage second-order cones of dimension soc_dim on that stage's variables.  Same SplitMix64 streams as
    synthetic_conic_qp, entries outside the structure set to zero.  Returns (ConicQP, point dict, lam)."""
    nx, ne = T * nv, (T - 1) * nd
    n_nn, n_soc = T * n_nn_stage, T * n_soc_stage
    nc, nonneg, soc = synthetic_layout(nx, ne, n_nn, n_soc, soc_dim)
    U = lambda name, lo, hi, cnt: uniform(problem_id, STREAMS[name], lo, hi, cnt)
    stage_of_var = np.arange(nx) // nv
    B = U("B", -1, 1, nx * nx).reshape(nx, nx).T
    P = (B + B.T) / (2.0 * np.sqrt(nv))
    P = P * (stage_of_var[:, None] == stage_of_var[None, :]) + 2.0 * np.eye(nx)
    q = U("q", -1, 1, nx)
    A = (U("A", -1, 1, ne * nx) / np.sqrt(2 * nv)).reshape(nx, ne).T
    if ne:
        st_row = np.arange(ne) // nd
        A = A * ((stage_of_var[None, :] == st_row[:, None]) | (stage_of_var[None, :] == st_row[:, None] + 1))
    G = (U("G", -1, 1, nc * nx) / np.sqrt(nv)).reshape(nx, nc).T
    st_cone = np.zeros(nc, dtype=int)
    st_cone[:n_nn] = np.arange(n_nn) // max(1, n_nn_stage)
    for j, c in enumerate(soc):
        if c:
            st_cone[np.array(c) - 1] = j // max(1, n_soc_stage)
    if nc:
        G = G * (stage_of_var[None, :] == st_cone[:, None])
    xbar = U("xbar", -1, 1, nx)
    b = A @ xbar
    cp = np.zeros(nc)
    cp[:n_nn] = U("hpos", 0.5, 1.5, n_nn)
    tails = U("cone_point_tail", -0.3, 0.3, nc)
    for c in soc:
        if c:
            idx = np.array(c) - 1
            cp[idx[1:]] = tails[idx[1:]]
            cp[idx[0]] = 1.0 + np.linalg.norm(cp[idx[1:]])
    h = G @ xbar + cp
    prob = ConicQP(P, q, A, b, G, h, nonnegative_indices=nonneg, second_order_indices=soc, name="staged")
    pt = dict(x=U("x", -1, 1, nx), r=0.1 * U("r", -1, 1, ne), y=U("y", -1, 1, ne), z=U("z", -1, 1, nc))
    sl = np.zeros(nc); t = np.zeros(nc)
    sl[:n_nn] = U("s_nn", 0.5, 1.5, n_nn); t[:n_nn] = U("t_nn", 0.5, 1.5, n_nn)
    stl = U("s_tail", -0.3, 0.3, nc); tt = U("t_tail", -0.3, 0.3, nc)
    for c in soc:
        if c:
            idx = np.array(c) - 1
            sl[idx[1:]] = stl[idx[1:]]; t[idx[1:]] = tt[idx[1:]]
            sl[idx[0]] = 1.0 + np.linalg.norm(sl[idx[1:]]); t[idx[0]] = 1.0 + np.linalg.norm(t[idx[1:]])
    pt["s"], pt["t"] = sl, t
    lam = U("lam", -1, 1, ne)
    prob.half_bandwidth = 2 * nv - 1                     # |i - j| of two variables of adjacent stages
    return prob, pt, lam


def cartpole_mpc(horizon=10, h=0.05, perturb=0.05):
    """BASELINE config C5 shape: cart-pole MPC of examples/autotuning/cartpole.jl:85-146 (model examples/autotuning/models/cartpole.jl:2-33):
    H = 10 stages, 4 states, 1 action => nx = 49, ne = 40 (36 explicit-midpoint dynamics + x_1 - x_init), nc = 0, p = 102 parameters
    theta_t = [xbar(4); ubar(1); w_Q(4); w_R(1); x_init(4) if t = 1], theta_T = [xbar(4); w_Q(4)].
    The tracking reference is synthetic (the reference's comes from a prior swing-up solve)."""
    import sympy as sp
    ns, na = 4, 1
    T = horizon
    nz = ns * T + na * (T - 1)
    mc, mp_, l, g = 1.0, 0.2, 0.5, 9.81

    def fcont(x, u):
        s_, c_ = sp.sin(x[1]), sp.cos(x[1])
        H11, H12, H22 = mc + mp_, mp_ * l * c_, mp_ * l ** 2
        det = H11 * H22 - H12 * H12
        Cqd0 = -mp_ * x[3] * l * s_ * x[3]
        r0 = Cqd0 + 0 - u[0]
        r1 = 0 + mp_ * g * l * s_ - 0
        qdd0 = -(H22 * r0 - H12 * r1) / det
        qdd1 = -(-H12 * r0 + H11 * r1) / det
        return [x[2], x[3], qdd0, qdd1]

    def fdisc(x, u):
        k1 = fcont(x, u)
        xm = [x[i] + 0.5 * h * k1[i] for i in range(4)]
        k2 = fcont(xm, u)
        return [x[i] + h * k2[i] for i in range(4)]

    xs = lambda z, t: z[t * (ns + na): t * (ns + na) + ns]
    us = lambda z, t: z[t * (ns + na) + ns: t * (ns + na) + ns + na]
    # parameter offsets
    offs = [0]
    for t in range(T - 1):
        offs.append(offs[-1] + (14 if t == 0 else 10))
    npar = offs[-1] + 8

    def objective(z, th):
        J = 0
        for t in range(T - 1):
            w = th[offs[t]:]
            X, U = xs(z, t), us(z, t)
            J += sum(0.5 * w[5 + i] ** 2 * (X[i] - w[i]) ** 2 for i in range(4)) + 0.5 * w[9] ** 2 * (U[0] - w[4]) ** 2
        w = th[offs[T - 1]:]
        XT = xs(z, T - 1)
        return J + sum(0.5 * w[4 + i] ** 2 * (XT[i] - w[i]) ** 2 for i in range(4))

    def equality(z, th):
        e = []
        for t in range(T - 1):
            X, U, Y = xs(z, t), us(z, t), xs(z, t + 1)
            fd = fdisc(X, U)
            e += [Y[i] - fd[i] for i in range(4)]
        X1 = xs(z, 0)
        e += [X1[i] - th[10 + i] for i in range(4)]
        return e

    theta = np.zeros(npar)
    x0 = np.zeros(nz)
    for t in range(T):
        xbar = np.array([0.0, np.pi * t / (T - 1), 0.0, 0.0])
        if t < T - 1:
            theta[offs[t]:offs[t] + 4] = xbar
            theta[offs[t] + 4] = 0.0
            theta[offs[t] + 5:offs[t] + 9] = 1.0
            theta[offs[t] + 9] = 1.0
            if t == 0:
                theta[10:14] = xbar + perturb * np.array([1.0, -1.0, 0.5, 0.25])
        else:
            theta[offs[t]:offs[t] + 4] = xbar
            theta[offs[t] + 4:offs[t] + 8] = 1.0
        x0[t * (ns + na): t * (ns + na) + ns] = xbar
    prob = SymbolicProblem(nz, objective, equality, None, np_=npar, parameters=theta, x0=x0, name="cartpole_mpc")
    prob.horizon = T
    return prob


def declared_structure(prob):
    """what a caller who knows the sparsity of the problem up front (the reference's methods.*_sparsity lists) hands to calipso_hip_create_structured:
    per row of [equality; cone] its first / last non-zero column (1-based, inclusive; first > last for an empty row) and the first columns of the
    diagonal blocks of the Lagrangian Hessian — here read off the QP data"""
    rows = []
    if prob.ne:
        rows.append(np.asarray(prob.A) != 0.0)
    if prob.nc:
        rows.append(np.asarray(prob.G) != 0.0)
    Zp = np.vstack(rows) if rows else np.zeros((0, prob.nx), dtype=bool)
    first = np.ones(Zp.shape[0], dtype=np.int64); last = np.zeros(Zp.shape[0], dtype=np.int64)
    for k in range(Zp.shape[0]):
        nz = np.nonzero(Zp[k])[0]
        if nz.size:
            first[k], last[k] = nz[0] + 1, nz[-1] + 1
    Hp = (np.asarray(prob.P) != 0.0) | (np.asarray(prob.P).T != 0.0)
    starts, reach = [1], -1
    for j in range(prob.nx):
        if j > starts[-1] - 1 and reach < j:
            starts.append(j + 1)
        nz = np.nonzero(Hp[:, j])[0]
        reach = max(reach, j, int(nz.max()) if nz.size else j)
    return dict(row_first=first, row_last=last, hessian_block_start=np.array(starts, dtype=np.int64))


def structure_from_pattern(prob, samples=3, seed=0):
    """the declared structure of ANY problem object with evaluate(): the union of the non-zero patterns of its Jacobians and Lagrangian Hessian over a few random
    points (what the reference holds as methods.*_sparsity, src/trajectory_optimization/sparsity.jl:28-129) -> the arguments of calipso_hip_create_structured"""
    rng = np.random.default_rng(seed)
    Zp = np.zeros((prob.ne + prob.nc, prob.nx), dtype=bool)
    Hp = np.zeros((prob.nx, prob.nx), dtype=bool)
    theta = np.asarray(getattr(prob, "parameters", np.zeros(0)), dtype=np.float64)
    sizes = {"objective": 1, "objective_gradient_variables": prob.nx, "equality_constraint": prob.ne, "cone_constraint": prob.nc,
             "equality_dual_jacobian_variables": prob.nx, "cone_dual_jacobian_variables": prob.nx, "equality_jacobian_variables": prob.ne * prob.nx,
             "cone_jacobian_variables": prob.nc * prob.nx, "objective_jacobian_variables_variables": prob.nx ** 2,
             "equality_dual_jacobian_variables_variables": prob.nx ** 2, "cone_dual_jacobian_variables_variables": prob.nx ** 2}
    for _ in range(samples):
        out = {}
        getter = lambda name: out.setdefault(name, np.zeros(sizes[name]))          # (evaluate writes column-major flat buffers)
        prob.evaluate(ALL_VARIABLE_FLAGS, rng.standard_normal(prob.nx), rng.standard_normal(prob.ne), rng.standard_normal(prob.nc), theta, getter)
        rows = []
        if prob.ne:
            rows.append(np.asarray(out["equality_jacobian_variables"]).reshape(prob.nx, prob.ne).T != 0.0)
        if prob.nc:
            rows.append(np.asarray(out["cone_jacobian_variables"]).reshape(prob.nx, prob.nc).T != 0.0)
        if rows:
            Zp |= np.vstack(rows)
        for name in ("objective_jacobian_variables_variables", "equality_dual_jacobian_variables_variables", "cone_dual_jacobian_variables_variables"):
            if name in out:
                M = np.asarray(out[name]).reshape(prob.nx, prob.nx) != 0.0
                Hp |= M | M.T
    first = np.ones(Zp.shape[0], dtype=np.int64); last = np.zeros(Zp.shape[0], dtype=np.int64)
    for k in range(Zp.shape[0]):
        nz = np.nonzero(Zp[k])[0]
        if nz.size:
            first[k], last[k] = nz[0] + 1, nz[-1] + 1
    starts, reach = [1], -1
    for j in range(prob.nx):
        if j > starts[-1] - 1 and reach < j:
            starts.append(j + 1)
        nz = np.nonzero(Hp[:, j])[0]
        reach = max(reach, j, int(nz.max()) if nz.size else j)
    return dict(row_first=first, row_last=last, hessian_block_start=np.array(starts, dtype=np.int64))
