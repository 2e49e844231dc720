"""GPU (-m gpu): solve! for a batch of small conic QPs in ONE kernel launch (csrc/smallnewton.hip, calipso_hip_smallnewton_*): every instance's accepted iterates,
iteration counts, factorisation counts and result agree with the ORACLE's solve! of the same problem (solve.jl:8-377; per accepted iterate 1e-8), with the general
device path (calipso_hip_solve with the attached QP evaluator) and — in benchmark mode — a non-advancing step leaves the state untouched."""
import numpy as np
import pytest

import problems as pr
from helpers import load_pkg
from test_oracle_solve import run as run_oracle

pytestmark = pytest.mark.gpu


def rel(a, b):
    return np.abs(np.asarray(a) - np.asarray(b)).max() / max(1.0, np.abs(np.asarray(b)).max())


def make_batch(pkg, probs, **opts):
    p0 = probs[0]
    sn = pkg.SmallNewtonBatch(p0.nx, p0.ne, p0.nc, len(probs), options=opts)
    st = lambda name: np.stack([np.asarray(getattr(p, name), dtype=np.float64) for p in probs])
    sn.set_qp(st("P"), st("q"), st("A"), st("b"), st("G"), st("h"), objective_scale=p0.c, shared=False)
    sn.initialize(np.stack([p.x0 for p in probs]))
    return sn


# threads per instance: 0 = the library's choice by the LDS footprint (csrc/smallnewton.hip: sn_threads), else the build of the kernel with that workgroup size
THREADS = [0, 64, 128, 256]


@pytest.mark.parametrize("threads", THREADS)
@pytest.mark.parametrize("shape", [(10, 4, 6), (12, 0, 9), (9, 5, 0), (49, 40, 0), (30, 12, 24), (70, 20, 10)])
def test_batched_solve_matches_the_oracle_per_accepted_iterate(oracle_mod, shape, threads):
    pkg = load_pkg()
    nx, ne, nc = shape
    probs = [pr.random_qp(nx, ne, nc, seed=100 + k, nonnegative_indices=list(range(1, nc + 1))) for k in range(6)]
    sn = make_batch(pkg, probs, threads=threads)
    sn.keep_trace(64)
    res, ms = sn.solve()
    st = sn.get_state()
    tr = sn.trace()
    for k, prob in enumerate(probs):
        o, status = run_oracle(oracle_mod, prob)
        assert status == int(res[k]) == 1, (k, status, res[k])
        os_ = o.stats()
        assert st["counters"]["total_iterations"][k] == os_["total_iterations"], (k, st["counters"]["total_iterations"][k], os_["total_iterations"])
        assert st["counters"]["outer"][k] == os_["outer"]
        assert 1 <= st["counters"]["factorizations"][k] <= os_["factorizations"]      # (the oracle also counts the reference's hidden re-factorisation before every solve, linear_solver.jl:53)
        assert st["counters"]["max_refinement_rounds"][k] == os_["max_refinement_rounds"]
        ot = o.trace()
        rows = int(st["counters"]["accepted_iterates"][k])
        assert rows == ot.shape[0] and rows >= 3
        for r in range(min(rows, 64)):
            assert rel(tr[k, r], ot[r]) <= 1e-8, (k, r, rel(tr[k, r], ot[r]))
        assert rel(st["solution"][k], o.point()["all"]) <= 1e-8
        assert rel(st["dual"][k], o.buf("dual")) <= 1e-8 if ne else True
        assert abs(st["scalars"][k, 0] - o.buf("central_path")[0]) <= 1e-15 and abs(st["scalars"][k, 2] - o.buf("penalty")[0]) <= 1e-9
    sn.close()


@pytest.mark.parametrize("threads", THREADS)
@pytest.mark.parametrize("layout", [(12, 4, 4, (3, 3)), (20, 8, 0, (4, 3, 3)), (16, 5, 6, (5,)), (30, 10, 3, (3, 3, 3, 3)), (10, 3, 8, (3,)), (14, 6, 10, (4,))])
def test_batched_solve_with_second_order_cones_matches_the_oracle(oracle_mod, layout, threads):
    """nonnegative entries followed by second-order cones (the friction-cone / portfolio shapes of the reference's tests): the arrow blocks, the closed-form inverses of
    cones/second_order.jl:50-65 with their first-row quirk, the triu-symmetrised cone block (which makes the first solve inexact: refinement rounds > 1) — per accepted
    iterate against the oracle, 1e-8"""
    pkg = load_pkg()
    nx, ne, q, dims = layout
    nc = q + sum(dims)
    soc, at = [], q + 1
    for dm in dims:
        soc.append(list(range(at, at + dm))); at += dm
    probs = [pr.random_qp(nx, ne, nc, seed=500 + k, nonnegative_indices=list(range(1, q + 1)), second_order_indices=soc) for k in range(5)]
    p0 = probs[0]
    sn = pkg.SmallNewtonBatch(nx, ne, nc, len(probs), options={"threads": threads})
    sn.set_cones(q, dims)
    st_ = lambda name: np.stack([np.asarray(getattr(p, name), dtype=np.float64) for p in probs])
    sn.set_qp(st_("P"), st_("q"), st_("A"), st_("b"), st_("G"), st_("h"), objective_scale=p0.c, shared=False)
    sn.initialize(np.stack([p.x0 for p in probs]))
    sn.keep_trace(96)
    res, _ = sn.solve()
    st = sn.get_state()
    tr = sn.trace()
    rounds_seen, compared = 0, 0
    for k, prob in enumerate(probs):
        o, status = run_oracle(oracle_mod, prob)
        os_ = o.stats()
        if os_["lu_fallbacks"] > 0:                       # the reference fell back to H \ residual: this path stops there and says so
            assert res[k] == -102, (k, res[k])
            continue
        assert status == int(res[k]) == 1, (k, status, res[k])
        compared += 1
        assert st["counters"]["total_iterations"][k] == os_["total_iterations"] and st["counters"]["outer"][k] == os_["outer"]
        assert st["counters"]["max_refinement_rounds"][k] == os_["max_refinement_rounds"]
        rounds_seen = max(rounds_seen, int(os_["max_refinement_rounds"]))
        ot = o.trace()
        rows = int(st["counters"]["accepted_iterates"][k])
        assert rows == ot.shape[0]
        for r in range(min(rows, 96)):
            assert rel(tr[k, r], ot[r]) <= 1e-8, (k, r, rel(tr[k, r], ot[r]))
        assert rel(st["solution"][k], o.point()["all"]) <= 1e-8
    assert compared == 0 or rounds_seen >= 2              # the triu-symmetrised cone blocks do make refinement work (cold-started cone problems often end in the reference's fallback: compared == 0)
    sn.close()


def test_batched_solve_agrees_with_the_general_device_path():
    """the same problems through calipso_hip_solve (one handle each, attached QP evaluator): same iteration counts, solutions to 1e-8"""
    pkg = load_pkg()
    probs = [pr.random_qp(14, 6, 8, seed=300 + k, nonnegative_indices=list(range(1, 9))) for k in range(4)]
    sn = make_batch(pkg, probs)
    res, _ = sn.solve()
    st = sn.get_state()
    for k, prob in enumerate(probs):
        s = pkg.Solver(prob, prob.nx, 0, prob.ne, prob.nc, nonnegative_indices=prob.nonnegative_indices)
        s.qp_attach(prob.P, prob.q, prob.A, prob.b, prob.G, prob.h, prob.c)
        pkg.initialize_b(s, prob.x0)
        assert pkg.solve_b(s) and res[k] == 1
        assert s.stats()["total_iterations"] == st["counters"]["total_iterations"][k] and s.stats()["factorizations"] == st["counters"]["factorizations"][k]
        assert rel(st["solution"][k], s.solution.all) <= 1e-8
    sn.close()


def test_one_problem_shared_by_the_batch_and_benchmark_steps():
    """shared problem data (one copy for all instances), different starting points; steps(advance=False) repeats the same step and leaves the state as it was"""
    pkg = load_pkg()
    prob = pr.random_qp(20, 8, 10, seed=7, nonnegative_indices=list(range(1, 11)))
    B = 33
    sn = pkg.SmallNewtonBatch(prob.nx, prob.ne, prob.nc, B)
    sn.set_qp(prob.P, prob.q, prob.A, prob.b, prob.G, prob.h, objective_scale=prob.c)
    rng = np.random.default_rng(0)
    sn.initialize(prob.x0[None, :] + 0.1 * rng.standard_normal((B, prob.nx)))
    res, _ = sn.solve()
    assert (res == 1).all()
    sols = sn.get_state()["solution"]
    assert rel(sols, np.repeat(sols[:1], B, axis=0)) <= 1e-6              # one convex problem: every start reaches the same solution (to the solver's tolerances)
    # benchmark steps from an interior state: put every instance back to a strictly feasible interior point with a large central-path parameter
    w = sols.copy()
    n_, ne, nc = prob.nx + prob.ne + prob.nc, prob.ne, prob.nc
    w[:, prob.nx + ne:prob.nx + ne + nc] += 0.5; w[:, -nc:] += 0.5
    sn.set_state(w=w, scalars=np.tile([0.17, 0.99, 52.0], (B, 1)))
    before = sn.get_state()
    info1, st1, _ = sn.steps(1, advance=False)
    info3, st3, _ = sn.steps(3, advance=False)
    after = sn.get_state()
    assert (st1 == 0).all() and (st3 == 0).all() and (info1[:, 6] == 0).all()
    assert np.array_equal(info1, info3)                                       # the third repetition of the step is the first
    assert np.array_equal(before["solution"], after["solution"]) and np.array_equal(before["scalars"][:, :3], after["scalars"][:, :3])
    info_a, st_a, _ = sn.steps(1, advance=True)
    assert np.array_equal(info_a[:, :6], info1[:, :6]) and not np.array_equal(sn.get_state()["solution"], before["solution"])
    sn.close()


def test_nonconvex_instances_run_the_regularisation_loop_like_the_oracle(oracle_mod):
    """P with negative curvature: IC-1 fails its inertia test and inertia_correction! (inertia.jl:30-80) factors again with growing primal regularisation — on the device,
    per instance (the convex members of the same batch take one factorisation per step): same iterates, iteration counts and final regularisation as the oracle"""
    pkg = load_pkg()
    probs = []
    for k, shift in enumerate((0.0, 3.0, 0.0, 8.0)):
        p = pr.random_qp(12, 5, 7, seed=700 + k, nonnegative_indices=list(range(1, 8)))
        p = pr.ConicQP(p.P - shift * np.eye(12), p.q, p.A, p.b, p.G, p.h, nonnegative_indices=p.nonnegative_indices, x0=p.x0, objective_scale=p.c)
        probs.append(p)
    sn = make_batch(pkg, probs, max_outer_iterations=6)
    sn.keep_trace(64)
    res, _ = sn.solve()
    st = sn.get_state()
    tr = sn.trace()
    facts = []
    for k, prob in enumerate(probs):
        o, status = run_oracle(oracle_mod, prob, max_outer_iterations=6)
        os_ = o.stats()
        if os_["lu_fallbacks"] > 0:
            assert res[k] == -102
            continue
        assert status == int(res[k]), (k, status, res[k])                      # 1 converged or 0: the iteration caps of a problem that is unbounded below
        assert st["counters"]["total_iterations"][k] == os_["total_iterations"]
        ot = o.trace()
        rows = int(st["counters"]["accepted_iterates"][k])
        assert rows == ot.shape[0]
        for r in range(min(rows, 64)):
            assert rel(tr[k, r], ot[r]) <= 1e-7, (k, r, rel(tr[k, r], ot[r]))
        assert abs(st["scalars"][k, 4] - o.buf("primal_regularization_last")[0]) <= 1e-12 * max(1.0, abs(o.buf("primal_regularization_last")[0]))
        facts.append(int(st["counters"]["factorizations"][k]) - int(st["counters"]["newton_steps"][k]))
    assert max(facts) > 0 and min(facts) == 0                                  # some instances re-factored, the convex ones never
    sn.close()


def test_filter_capacity_option(oracle_mod):
    """options.max_filter (filter.jl:7-13): a smaller filter array per instance — the solve does not depend on the capacity as long as the pairs fit"""
    pkg = load_pkg()
    probs = [pr.random_qp(10, 4, 6, seed=100 + k, nonnegative_indices=list(range(1, 7))) for k in range(3)]
    a = make_batch(pkg, probs)
    b = make_batch(pkg, probs, max_filter=16)
    ra, _ = a.solve(); rb, _ = b.solve()
    assert (ra == 1).all() and (rb == 1).all()
    assert np.array_equal(a.get_state()["solution"], b.get_state()["solution"])
    with pytest.raises(pkg.CalipsoHipError):
        a.set_option("max_filter", 0)
    a.close(); b.close()


def test_unconstrained_and_single_instance():
    pkg = load_pkg()
    prob = pr.random_qp(15, 0, 0, seed=9, nonnegative_indices=[])
    sn = make_batch(pkg, [prob])
    res, _ = sn.solve()
    assert res[0] == 1
    x = sn.get_state()["solution"][0, :15]
    xs = np.linalg.solve(2.0 * prob.c * prob.P, -prob.q)                       # min c x'Px + q'x
    assert rel(x, xs) <= 1e-4                                                  # (to the solver's tolerances: residual_tolerance 1e-4, primal regularisation 1e-7)
    sn.close()


def test_limits_are_refused():
    pkg = load_pkg()
    with pytest.raises(pkg.CalipsoHipError, match="LDS"):
        pkg.SmallNewtonBatch(100, 150, 100, 2)
    sn = pkg.SmallNewtonBatch(5, 2, 3, 2)
    with pytest.raises(pkg.CalipsoHipError, match="no problem data"):
        sn.solve()
    with pytest.raises(pkg.CalipsoHipError):
        sn.set_option("residual_norm", 2.0)
    sn.close()


@pytest.mark.parametrize("threads", [0, 256])
@pytest.mark.parametrize("layout", [(12, 5, 6, 0, 0), (49, 40, 0, 0, 0), (24, 9, 11, 0, 0)])
def test_batched_differentiate_matches_the_oracle(oracle_mod, layout, threads):
    """differentiate! of a batch in one launch (differentiate.jl:1-61; calipso_hip_smallnewton_differentiate): parametric conic QPs theta = [dq; db; dh]
    (np = nx + ne + nc: every entry of dR/dtheta's three blocks is exercised), solved by the batched kernel, then sensitivities of every instance against the
    ORACLE's differentiate! at its own solution (1e-7: the two solutions agree to 1e-8 and the condensed matrix at the solution is what it is) — R+ only,
    mixed with second-order cones, C5's (49, 40, 0)."""
    pkg = load_pkg()
    nx, ne, nnn, nsoc, sdim = layout
    probs = [pr.parametric_conic_qp(nx, ne, nnn, nsoc, sdim, seed=300 + k) for k in range(5)]
    p0 = probs[0]
    nc = p0.nc
    opts = dict(residual_tolerance=1e-6, optimality_tolerance=1e-6, equality_tolerance=1e-6, complementarity_tolerance=1e-6, slack_tolerance=1e-6)
    sn = pkg.SmallNewtonBatch(nx, ne, nc, len(probs), options=dict(threads=threads, **opts))
    if nsoc:
        sn.set_cones(nnn, [sdim] * nsoc)
    st_ = lambda name: np.stack([np.asarray(getattr(p, name), dtype=np.float64) for p in probs])
    sn.set_qp(st_("P"), st_("q"), st_("A"), st_("b"), st_("G"), st_("h"), objective_scale=p0.c, shared=False)
    sn.initialize(np.stack([p.x0 for p in probs]))
    res, _ = sn.solve()
    N, npar = nx + 2 * ne + 3 * nc, nx + ne + nc
    J = np.zeros((len(probs), N, npar))
    oy, oz = nx + ne + nc, nx + 2 * ne + nc                                       # point.jl: x | r | s | y | z | t
    J[:, :nx, :nx] = np.eye(nx)
    J[:, oy:oy + ne, nx:nx + ne] = -np.eye(ne)
    J[:, oz:oz + nc, nx + ne:] = np.eye(nc)
    S, st, ms = sn.differentiate(J)
    sol = sn.get_state()["solution"].copy()
    compared, ok, So_all = 0, [], {}
    for k, prob in enumerate(probs):
        o, status = run_oracle(oracle_mod, prob, differentiate=1, **opts)
        if o.stats()["lu_fallbacks"] > 0 or res[k] != 1:      # (cold-started cone problems may end in the reference's fallback to H \\ residual: this path stops there, -102)
            assert res[k] == -102 or status != 1, (k, res[k], status)
            continue
        assert status == 1 and st[k] == 0
        compared += 1
        assert rel(sol[k], o.point()["all"]) <= 1e-8
        assert np.abs(J[k] - o.mat("jacobian_parameters", o.N, prob.np)).max() == 0.0
        So_all[k] = o.mat("solution_sensitivity", o.N, prob.np).copy()
        # at its OWN solution: the points agree to 1e-8 absolute, but the duals of inactive and the slacks of active constraints are ~ kappa themselves and the
        # condensed matrix carries their ratios: the sensitivities agree as far as that allows
        assert rel(S[k], So_all[k]) <= 1e-3, (k, rel(S[k], So_all[k]))
        sol[k] = o.point()["all"].copy()
        ok.append(k)
    assert compared == len(probs)
    # at the ORACLE's solution (the same point to the bit; the scalars the batch's own solve! left — kappa, rho and the regularisation are equal): to rounding
    sn.set_state(w=sol)
    S, st1, _ = sn.differentiate(J)
    for k in ok:
        assert st1[k] == 0 and rel(S[k], So_all[k]) <= 1e-9, (k, rel(S[k], So_all[k]))
    # a second call (a subset of the columns) reuses the buffers and gives the same columns; ONE matrix for all instances (the model is the same) as well
    S2, st2, _ = sn.differentiate(J[:, :, :3])
    assert np.array_equal(S2, S[:, :, :3])
    S3, st3, _ = sn.differentiate(J[0])
    assert np.array_equal(S3, S)
    sn.close()


@pytest.mark.parametrize("layout", [(20, 8, 4, 2, 3), (30, 10, 0, 3, 4), (16, 5, 6, 1, 5), (12, 4, 5, 0, 0)])
def test_batched_differentiate_at_interior_points_with_second_order_cones(oracle_mod, layout):
    """the same entry at INTERIOR points (well conditioned; cold-started cone problems mostly end in the reference's fallback, so there is no solution to stand on):
    a non-advancing Newton step forms the cone Jacobians there (what differentiate! then finds, quirk B-12), and the sensitivities are compared column by column with
    the ORACLE's residual_jacobian_variables! / factorize! / search_direction_symmetric! at the same point and scalars — second-order cones included, where the
    reference's solve is the unrefined one with triu-symmetrised cone blocks (quirk B-3) and this path follows it; 1e-8"""
    from helpers import interior_point
    pkg = load_pkg()
    nx, ne, nnn, nsoc, sdim = layout
    probs = [pr.parametric_conic_qp(nx, ne, nnn, nsoc, sdim, seed=700 + k) for k in range(4)]
    p0 = probs[0]
    nc = p0.nc
    kappa, tau, rho = 0.17, 0.99, 52.0
    sn = pkg.SmallNewtonBatch(nx, ne, nc, len(probs))
    if nsoc:
        sn.set_cones(nnn, [sdim] * nsoc)
    st_ = lambda name: np.stack([np.asarray(getattr(p, name), dtype=np.float64) for p in probs])
    sn.set_qp(st_("P"), st_("q"), st_("A"), st_("b"), st_("G"), st_("h"), objective_scale=p0.c, shared=False)
    N, npar = nx + 2 * ne + 3 * nc, nx + ne + nc
    oy, oz = nx + ne + nc, nx + 2 * ne + nc
    J = np.zeros((len(probs), N, npar))
    J[:, :nx, :nx] = np.eye(nx)
    J[:, oy:oy + ne, nx:nx + ne] = -np.eye(ne)
    J[:, oz:oz + nc, nx + ne:] = np.eye(nc)
    oracles, W, LAM = [], [], []
    for k, prob in enumerate(probs):
        pt, lam = interior_point(prob, seed=40 + k)
        o = oracle_mod.OracleSolver(prob.nx, prob.np, prob.ne, prob.nc, prob.nonnegative_indices, prob.second_order_indices)
        op = o.point()
        for f in "xrsyzt":
            op[f][:] = pt[f]
        o.buf("dual")[:] = lam
        for name, v in (("central_path", kappa), ("penalty", rho), ("primal_regularization", 1.0e-7), ("dual_regularization", 1.0e-7), ("fraction_to_boundary", tau)):
            o.buf(name)[0] = v
        prob.evaluate(pr.ALL_VARIABLE_FLAGS, op["x"], op["y"], op["z"], prob.parameters, o.buf)
        o.cone(product=True, jacobian=True, target=True)
        o.residual_jacobian_variables(); o.residual_jacobian_variables_symmetric()
        oracles.append(o); W.append(op["all"].copy()); LAM.append(lam)
    sn.set_state(w=np.stack(W), dual=np.stack(LAM) if ne else None, scalars=np.tile([kappa, tau, rho], (len(probs), 1)))
    info, stp, _ = sn.steps(1, advance=False)                  # forms the cone Jacobians at these points, leaves the points where they are
    assert (stp == 0).all(), stp
    S, st, _ = sn.differentiate(J)
    assert (st == 0).all()
    for k, o in enumerate(oracles):
        for j in range(npar):
            o.buf("residual")[:] = J[k][:, j]
            o.search_direction_symmetric(0, fact=(j == 0))
            assert rel(S[k][:, j], -1.0 * o.buf("step")) <= 1e-8, (k, j, rel(S[k][:, j], -1.0 * o.buf("step")))
    sn.close()
