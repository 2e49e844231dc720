"""GPU: groups — several handles of one shape stepped in lockstep through the same launches (include/calipso_hip.h "groups";
BASELINE config C4 runs many independent instances per GPU).  Per member the arithmetic must be exactly that of
stepping the member alone: every comparison below is bit-for-bit."""
import numpy as np
import pytest

import problems as pr
from helpers import load_pkg

pytestmark = pytest.mark.gpu

SHAPE = (300, 140, 40, 20, 3)


def build(pkg, pid, shape=SHAPE, indefinite=0.0):
    nx, ne, n_nn, n_soc, dim = shape
    prob, pt, lam = pr.synthetic_conic_qp(pkg.splitmix_uniform, pid, nx, ne, n_nn, n_soc, dim)
    if indefinite:
        prob.P = prob.P - indefinite * np.eye(nx)      # negative curvature: the inertia test fails at IC-1 and the regularisation loop runs
    s = pkg.Solver(prob, prob.nx, 0, prob.ne, prob.nc, nonnegative_indices=prob.nonnegative_indices, second_order_indices=prob.second_order_indices)
    s.set("solution", np.concatenate([pt[k] for k in "xrsyzt"]))
    s.set("dual", lam)
    for name, v in (("central_path", 0.17), ("penalty", 52.0), ("fraction_to_boundary", 0.99)):
        s.set(name, [v])
    s.qp_attach(prob.P, prob.q, prob.A, prob.b, prob.G, prob.h, 0.5)
    fl = pkg.FLAGS
    s.qp_evaluate(fl["objective"] | fl["equality_constraint"] | fl["cone_constraint"], 0)
    s.cone(product=True, target=True)
    s.synchronize()
    return s


def same(a, b):
    return np.array_equal(np.asarray(a), np.asarray(b))


def test_group_step_is_bitwise_the_single_step():
    pkg = load_pkg()
    ids = [3, 4, 5, 6]
    singles = [build(pkg, p) for p in ids]
    members = [build(pkg, p) for p in ids]
    g = pkg.Group(members)
    ref = [s.newton_step(advance=False) for s in singles]
    got = g.newton_step(advance=False)
    for r, q, s, m in zip(ref, got, singles, members):
        assert r == q, (r, q)
        assert same(s.data("step").all, m.data("step").all)
        assert same(s.data("residual").all, m.data("residual").all)
        assert same(s.solution.all, m.solution.all)                      # restored iterate
    # a second identical call gives identical results (benchmark mode restores the state)
    again = g.newton_step(advance=False)
    assert again == got
    g.close()


def test_group_advancing_iterates_stay_bitwise_equal():
    pkg = load_pkg()
    ids = [11, 12, 13]
    singles = [build(pkg, p) for p in ids]
    members = [build(pkg, p) for p in ids]
    g = pkg.Group(members)
    for it in range(4):
        ref = [s.newton_step(advance=True) for s in singles]
        got = g.newton_step(advance=True)
        for r, q, s, m in zip(ref, got, singles, members):
            assert r == q, (it, r, q)
            assert same(s.solution.all, m.solution.all), it
    # members remain usable on their own afterwards
    r = singles[1].newton_step(advance=True)
    q = members[1].newton_step(advance=True)
    assert r == q and same(singles[1].solution.all, members[1].solution.all)
    g.close()


def test_group_members_take_different_paths():
    """one member needs the regularisation loop of inertia_correction! (inertia.jl:30-80), the others do not: the re-factorisation
    rounds run on the sub-list of members that need them"""
    pkg = load_pkg()
    spec = [(21, 0.0), (22, 6.0), (23, 0.0), (24, 40.0)]
    singles = [build(pkg, p, indefinite=v) for p, v in spec]
    members = [build(pkg, p, indefinite=v) for p, v in spec]
    g = pkg.Group(members)
    ref = [s.newton_step(advance=True) for s in singles]
    got = g.newton_step(advance=True)
    assert [r["factorizations"] for r in ref][0] == 1 and max(r["factorizations"] for r in ref) > 1
    for r, q, s, m in zip(ref, got, singles, members):
        assert r == q, (r, q)
        assert same(s.solution.all, m.solution.all)
        assert s.get("primal_regularization", 1)[0] == m.get("primal_regularization", 1)[0]
        assert s.stats() == m.stats()
    g.close()


def test_group_of_one_and_shape_mismatch():
    pkg = load_pkg()
    a, b = build(pkg, 31), build(pkg, 31)
    g = pkg.Group([b])
    assert a.newton_step(advance=False) == g.newton_step(advance=False)[0]
    g.close()
    c = build(pkg, 32, shape=(200, 90, 30, 20, 3))
    with pytest.raises(pkg.CalipsoHipError):
        pkg.Group([a, c])


def build_for_solve(pkg, pid, shape=(120, 50, 20, 10, 3), **opts):
    nx, ne, n_nn, n_soc, dim = shape
    prob, pt, lam = pr.synthetic_conic_qp(pkg.splitmix_uniform, pid, nx, ne, n_nn, n_soc, dim)
    s = pkg.Solver(prob, prob.nx, 0, prob.ne, prob.nc, nonnegative_indices=prob.nonnegative_indices, second_order_indices=prob.second_order_indices,
                   options=opts)
    s.qp_attach(prob.P, prob.q, prob.A, prob.b, prob.G, prob.h, 0.5)
    s.set("solution", np.concatenate([pt["x"], np.zeros(s.N - nx)]))
    return prob, s


def test_group_solve_is_bitwise_the_single_solves():
    """solve! (solve.jl:8-377) in lockstep: members converge after different numbers of inner / outer iterations and drop out of the
    launches one by one; every member ends exactly where its stand-alone solve ends"""
    pkg = load_pkg()
    ids = [40, 41, 42, 43, 44]
    singles = [build_for_solve(pkg, p) for p in ids]
    members = [build_for_solve(pkg, p) for p in ids]
    ref = [pkg.solve_b(s) for _, s in singles]
    g = pkg.Group([s for _, s in members])
    got = g.solve()
    assert [int(r) for r in ref] == got and all(ref)
    its = [s.stats()["total_iterations"] for _, s in singles]
    assert len(set(its)) > 1                                     # the members do not all take the same path
    for (prob, s), (_, m) in zip(singles, members):
        assert s.stats() == m.stats()
        assert same(s.solution.all, m.solution.all)
        assert same(s.get("dual", prob.ne), m.get("dual", prob.ne))
        for name in ("central_path", "penalty", "primal_regularization"):
            assert s.get(name, 1)[0] == m.get(name, 1)[0], name
        assert np.abs(prob.A @ m.solution.variables - prob.b).max() < 1e-4
    g.close()


def test_group_solve_with_iteration_caps():
    """members stopped by max_residual_iterations / max_outer_iterations report 0 like the single solve"""
    pkg = load_pkg()
    opts = dict(max_outer_iterations=2, max_residual_iterations=3)
    singles = [build_for_solve(pkg, p, **opts) for p in (50, 51)]
    members = [build_for_solve(pkg, p, **opts) for p in (50, 51)]
    ref = [int(pkg.solve_b(s)) for _, s in singles]
    g = pkg.Group([s for _, s in members])
    assert g.solve() == ref and ref == [0, 0]
    for (_, s), (_, m) in zip(singles, members):
        assert s.stats() == m.stats() and same(s.solution.all, m.solution.all)
    g.close()


def test_group_of_eight_mid_size():
    """8 x (nx = 1100, three solve blocks): group launches with thousands of trailing-update tiles and the group-sized Schur tile
    shape (128 x 128 instead of the single handle's choice) produce the bits of the stand-alone steps"""
    pkg = load_pkg()
    shape = (1100, 400, 100, 50, 3)
    ids = list(range(60, 68))
    singles = [build(pkg, p, shape=shape) for p in ids]
    members = [build(pkg, p, shape=shape) for p in ids]
    g = pkg.Group(members)
    ref = [s.newton_step(advance=True) for s in singles]
    got = g.newton_step(advance=True)
    for r, q, s, m in zip(ref, got, singles, members):
        assert r == q and r["status"] == 0
        assert same(s.data("step").all, m.data("step").all)
        assert same(s.solution.all, m.solution.all)
    g.close()


@pytest.mark.parametrize("shape", [(40, 15, 0, 0, 3), (50, 0, 10, 4, 3), (64, 20, 12, 0, 3), (130, 30, 0, 6, 5)])
def test_group_degenerate_shapes(shape):
    """no cones / no equalities / no second-order cones / only second-order cones: group == single, bit for bit"""
    pkg = load_pkg()
    ids = [70, 71, 72]
    singles = [build(pkg, p, shape=shape) for p in ids]
    members = [build(pkg, p, shape=shape) for p in ids]
    g = pkg.Group(members)
    for it in range(2):
        ref = [s.newton_step(advance=True) for s in singles]
        got = g.newton_step(advance=True)
        for r, q, s, m in zip(ref, got, singles, members):
            assert r == q, (it, r, q)
            assert same(s.solution.all, m.solution.all)
    g.close()


def test_group_solve_through_host_callbacks():
    """members without a device evaluator (the general case: user functions behind the evaluation callback): four pendulum
    swing-ups (BASELINE config C2 shape, different initial guesses) solved in lockstep end exactly where their stand-alone
    solves end"""
    pkg = load_pkg()
    rng = np.random.default_rng(8)
    guesses = [np.zeros(10), 0.1 * rng.standard_normal(10), rng.standard_normal(10), 0.5 * np.ones(10)]

    options = [dict(), dict(central_path_initial=0.3), dict(penalty_initial=5.0), dict(residual_tolerance=1e-6, equality_tolerance=1e-6)]

    def make(g, o):
        prob = pr.pendulum(action_guess=g)
        s = pkg.Solver(prob, prob.nx, prob.np, prob.ne, prob.nc, options=o)
        pkg.initialize_b(s, prob.x0)
        return s

    singles = [make(g, o) for g, o in zip(guesses, options)]
    members = [make(g, o) for g, o in zip(guesses, options)]
    ref = [int(pkg.solve_b(s)) for s in singles]
    grp = pkg.Group(members)
    got = grp.solve()
    assert got == ref and all(ref)
    assert len({s.stats()["total_iterations"] for s in singles}) > 1
    for s, m in zip(singles, members):
        assert s.stats() == m.stats()
        assert same(s.solution.all, m.solution.all)
    grp.close()


def test_group_solve_with_parameters_differentiates_each_member():
    """differentiate=true and np > 0 (test/solver/qp_equality.jl): the sensitivities of every member of a lockstep solve are those of
    its stand-alone solve, bit for bit (including the member that is the group's base handle)"""
    pkg = load_pkg()
    opts = dict(residual_tolerance=1e-8, equality_tolerance=1e-6, complementarity_tolerance=1e-6, differentiate=1)

    def make(seed):
        prob = pr.qp_equality_parametric(seed=seed)
        s = pkg.Solver(prob, prob.nx, prob.np, prob.ne, prob.nc, parameters=prob.parameters, options=opts)
        pkg.initialize_b(s, prob.x0)
        return s

    singles = [make(k) for k in (5, 6, 7)]
    members = [make(k) for k in (5, 6, 7)]
    ref = [int(pkg.solve_b(s)) for s in singles]
    grp = pkg.Group(members)
    assert grp.solve() == ref and all(ref)
    for s, m in zip(singles, members):
        assert same(s.solution.all, m.solution.all)
        S = s.data("solution_sensitivity")
        assert np.abs(S).max() > 0 and same(S, m.data("solution_sensitivity"))
    grp.close()


@pytest.mark.parametrize("n", [5, 7, 13, 16, 27, 32])
def test_group_sizes_on_the_flattened_grids(n):
    """the group launches of the Schur complement and of the trailing updates run on ONE flattened grid whose workgroups are dealt to the
    XCDs by (instance, tile) ranges (schur.hip, ldl.hip): sizes that are not multiples of 8, and the largest group, give the stand-alone bits"""
    pkg = load_pkg()
    shape = (700, 250, 60, 30, 3) if n <= 16 else (300, 140, 40, 20, 3)
    ids = list(range(200, 200 + n))
    singles = [build(pkg, p, shape=shape) for p in ids]
    members = [build(pkg, p, shape=shape) for p in ids]
    g = pkg.Group(members)
    for it in range(2):
        ref = [s.newton_step(advance=True) for s in singles]
        got = g.newton_step(advance=True)
        for r, q, s, m in zip(ref, got, singles, members):
            assert r == q and r["status"] == 0, (it, r, q)
            assert same(s.data("step").all, m.data("step").all)
            assert same(s.solution.all, m.solution.all)
    g.close()
    if n == 32:                                            # (the limit is MAX_BATCH = 128 of internal.hpp: tests/test_gpu_blocks.py takes groups of 64 and 128)
        extra = build(pkg, 999, shape=shape)
        g33 = pkg.Group(members + [extra])
        g33.close()


def test_group_with_wide_second_order_cones_is_bitwise_the_single_step():
    """cones of dimension 12 (the reference's portfolio size) take the wave-per-cone kernels of csrc/soc_wide.hip: also per member of a group"""
    pkg = load_pkg()
    shape = (200, 60, 24, 6, 12)          # 24 nonnegative rows + 6 cones of dimension 12
    ids = [21, 22, 23]
    singles = [build(pkg, p, shape) for p in ids]
    members = [build(pkg, p, shape) for p in ids]
    g = pkg.Group(members)
    for it in range(2):
        ref = [s.newton_step(advance=True) for s in singles]
        got = g.newton_step(advance=True)
        for r, q, s, m in zip(ref, got, singles, members):
            assert r == q and r["status"] >= 0, (it, r, q)
            assert same(s.data("step").all, m.data("step").all)
            assert same(s.solution.all, m.solution.all)
    g.close()


def test_solve_block_option_keeps_group_and_single_bitwise_and_agrees_across_settings():
    """"opt.solve_block" (512 / 1024: the widest diagonal block of L whose inverse is assembled for the triangular solves) — for each setting a member of
    a group gets the bits of the same handle stepped alone; the two settings agree to rounding; anything else is refused"""
    pkg = load_pkg()
    shape = (1700, 300, 40, 20, 3)                     # NP = 2048: one block of 2048, two of 1024, or four of 512
    steps = {}
    for blockw in (2048, 1024, 512):
        singles = [build(pkg, p, shape) for p in (31, 32)]
        members = [build(pkg, p, shape) for p in (31, 32)]
        for h in singles + members:
            h.set_option("solve_block", blockw)
        g = pkg.Group(members)
        ref = [s.newton_step(advance=False) for s in singles]
        got = g.newton_step(advance=False)
        for r, q, s, m in zip(ref, got, singles, members):
            assert r == q and r["status"] == 0
            assert same(s.data("step").all, m.data("step").all)
        steps[blockw] = singles[0].data("step").all.copy()
        g.close()
    assert np.abs(steps[512] - steps[1024]).max() <= 1e-9 * max(1.0, np.abs(steps[1024]).max())
    assert np.abs(steps[2048] - steps[1024]).max() <= 1e-9 * max(1.0, np.abs(steps[1024]).max())
    with pytest.raises(pkg.CalipsoHipError, match="512, 1024 or 2048"):
        singles[0].set_option("solve_block", 256)
    members[1].set_option("solve_block", 1024)            # members[0] stays at 512
    g = pkg.Group(members)
    with pytest.raises(pkg.CalipsoHipError, match="agree on opt.solve_block"):
        g.newton_step(advance=False)
    g.close()


def test_newton_steps_in_one_call_are_the_steps_called_one_by_one():
    """calipso_hip_newton_steps: K steps without returning to the host language in between — same infos, same iterates, bit for bit"""
    pkg = load_pkg()
    a, b = build(pkg, 9), build(pkg, 9)
    one = [a.newton_step(advance=True) for _ in range(3)]
    many = b.newton_steps(3, advance=True)
    assert one == many
    assert same(a.solution.all, b.solution.all) and same(a.data("step").all, b.data("step").all)
    assert b.newton_steps(0) == []
    again = b.newton_steps(2, advance=False)                       # benchmark mode: the iterate is restored after every step
    assert again[0] == again[1]
    assert same(a.solution.all, b.solution.all)


def test_lanes_of_a_batch_run_side_by_side():
    """calipso_hip_streams_concurrent / calipso_hip_rebind_stream: whether the streams of two handles overlap is measured (a long kernel on one, a short one on the
    other, both ways), a colliding handle gets a new stream and stays fully usable; BatchSolver probes the leaders of its lanes at creation and leaves no colliding
    pair (BASELINE config 4's dense batch lost 14 % to such a collision, by the accident of how many streams the process had created before)."""
    pkg = load_pkg()
    from calipso_jl_amd.batch import BatchSolver
    shape = (300, 60, 12, 6, 3)
    hs = [build(pkg, 70 + k, shape) for k in range(6)]
    ok, t_ab, t_ba, t_long, t_alone, t_both = hs[0].streams_concurrent(hs[1])
    assert t_long > 50.0 and min(t_ab, t_ba) > 0.0 and t_both >= 0.5 * t_alone > 0.0                       # the long kernel really is long (us), the short one was timed
    ref = build(pkg, 72, shape)                                          # the same problem as hs[2], its stream untouched
    hs[2].rebind_stream()                                                # a new stream, same priority class: the handle steps as before, bit for bit
    hs[2].rebind_stream(1)
    a = hs[2].newton_step(advance=True); b = ref.newton_step(advance=True)
    assert a["status"] >= 0 and a == b
    assert np.array_equal(hs[2].solution.all, ref.solution.all)
    groups = [pkg.Group(hs[0:2]), pkg.Group(hs[2:4]), pkg.Group(hs[4:6])]
    bs = BatchSolver(groups, lanes=3)
    rep = bs.stream_report
    assert rep is not None and rep["pairs"] == 3 and rep["left"] == 0, rep
    for i in range(3):
        for j in range(i):
            assert hs[2 * i].streams_concurrent(hs[2 * j])[0]
    infos = bs.newton_step(advance=False)
    assert all(m["status"] >= 0 for g in infos for m in g)
    bs.close()
    with pytest.raises(pkg.CalipsoHipError):
        hs[0].streams_concurrent(hs[0])
