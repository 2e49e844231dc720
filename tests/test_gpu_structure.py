"""GPU: stage-banded structure (csrc/structure.hip).  Trajectory-optimisation problems order their variables stage
by stage, the Schur complement onto x is then banded and the device path skips everything outside the band.  The banded
treatment must give the bits of the dense treatment, and the dense treatment is what the rest of the suite checks against the oracle."""
import numpy as np
import pytest

import problems as pr
from helpers import load_pkg

pytestmark = pytest.mark.gpu


def build(pkg, pid, T, nv, nd, nn, nsoc, dim):
    prob, pt, lam = pr.staged_conic_qp(pkg.splitmix_uniform, pid, T, nv, nd, nn, nsoc, dim)
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
    return prob, s


STAGED = [
    # T, nv, nd, nonnegative rows / stage, second-order cones / stage, cone dimension
    (24, 30, 20, 4, 2, 3),        # nx = 720 (NP = 1024): band of 1 block
    (41, 56, 40, 6, 3, 2),        # nx = 2296: the quadruped-gait size of BASELINE config C4, band of 2 blocks
    (10, 100, 60, 10, 4, 4),      # wide stages: band of 4 blocks
    (6, 40, 0, 5, 2, 3),          # no dynamics: block-diagonal S
]


@pytest.mark.parametrize("shape", STAGED)
def test_banded_treatment_is_bitwise_the_dense_one(shape):
    pkg = load_pkg()
    prob, dense = build(pkg, 5, *shape)
    _, banded = build(pkg, 5, *shape)
    info = banded.analyze_structure()
    assert info["half_bandwidth"] == (prob.half_bandwidth if shape[2] else shape[1] - 1)
    NP = ((prob.nx + 511) // 512) * 512
    assert 0 < info["band_blocks"] < NP // 64 - 1
    for it in range(3):
        a = dense.newton_step(advance=True)
        b = banded.newton_step(advance=True)
        assert a == b and a["status"] == 0, (it, a, b)
        assert np.array_equal(dense.data("step").all, banded.data("step").all)
        assert np.array_equal(dense.solution.all, banded.solution.all)
    assert dense.factorize() == banded.factorize()
    # back to the dense treatment
    banded.clear_structure()
    assert dense.newton_step(advance=True) == banded.newton_step(advance=True)
    assert np.array_equal(dense.solution.all, banded.solution.all)


def test_dense_problem_has_nothing_to_skip():
    pkg = load_pkg()
    import test_gpu_group as tg
    a, b = tg.build(pkg, 3, shape=(700, 120, 40, 20, 3)), tg.build(pkg, 3, shape=(700, 120, 40, 20, 3))
    info = b.analyze_structure()
    assert info["half_bandwidth"] == 699 and (info["band_blocks"] == 0 or info["band_blocks"] * 64 >= 699)
    assert info["equality_rows_per_group"] == 120 and info["cone_rows_per_group"] == 100
    assert a.newton_step(advance=True) == b.newton_step(advance=True)
    assert np.array_equal(a.solution.all, b.solution.all)


def test_group_of_banded_members():
    pkg = load_pkg()
    shape = STAGED[0]
    singles = [build(pkg, p, *shape)[1] for p in (7, 8, 9)]
    members = [build(pkg, p, *shape)[1] for p in (7, 8, 9)]
    for m in members:
        m.analyze_structure()
    g = pkg.Group(members)
    for it in range(2):
        ref = [s.newton_step(advance=True) for s in singles]
        got = g.newton_step(advance=True)
        for r, q, s, m in zip(ref, got, singles, members):
            assert r == q and np.array_equal(s.solution.all, m.solution.all)
    g.close()


@pytest.mark.parametrize("stage_parallel", [False, True])
@pytest.mark.parametrize("shape", [STAGED[0], STAGED[2], (12, 40, 30, 4, 2, 3)])
def test_banded_step_matches_the_oracle(oracle_mod, shape, stage_parallel):
    """f1 against the ORACLE (not against the dense device path): one inner Newton iteration of a stage-structured problem with the
    banded treatment on — step, residual, inertia, refinement rounds, cone step sizes — equals the CPU restatement, whose sparse
    up-looking LDL^T (qdldl.jl:400-589 restated) factors the same structured K.  Tolerances of SURVEY.md 8(c)."""
    pkg = load_pkg()
    prob, s = build(pkg, 11, *shape)
    info = s.analyze_structure()
    assert info["band_blocks"] > 0
    if stage_parallel:                      # S through the multifrontal sparse LDL^T over a nested dissection of its pattern
        if shape == STAGED[2]:
            with pytest.raises(pkg.CalipsoHipError):      # stages of 100 variables: fronts of 300 rows do not fit one CU's LDS — refused, blocked path stays
                s.set_stage_parallel(True)
        else:
            assert s.set_stage_parallel(True)["largest_front"] <= 196
    w = s.get("solution", s.N)
    lam = s.get("dual", s.ne)
    step_info = s.newton_step(advance=False)
    assert step_info["status"] == 0
    step, R = s.data("step").all, s.data("residual").all
    o = oracle_mod.OracleSolver(prob.nx, 0, prob.ne, prob.nc, prob.nonnegative_indices, prob.second_order_indices)
    o.point()["all"][:] = w
    o.buf("dual")[:] = lam
    o.buf("central_path")[0] = 0.17; o.buf("penalty")[0] = 52.0; o.buf("fraction_to_boundary")[0] = 0.99
    o.set_int("linear_solve_refactor", 0)
    op = o.point()
    prob.evaluate(pr.ALL_VARIABLE_FLAGS, op["x"], op["y"], op["z"], np.zeros(0), o.buf)
    o.cone(product=True, jacobian=True, target=True, barrier=True, barrier_gradient=True)
    o.residual()
    assert o.search_direction() == 0
    so, Ro = np.array(o.buf("step")), np.array(o.buf("residual"))
    assert np.abs(R - Ro).max() <= 1e-12 * max(1.0, np.abs(Ro).max())
    assert np.abs(step - so).max() <= 1e-8 * max(1.0, np.abs(so).max())
    assert o.stats()["last_refinement_rounds"] == step_info["refinement_rounds"]
    assert tuple(o.compute_inertia()) == (prob.nx, prob.ne + prob.nc, 0)
    inertia, warn = s.factorize()
    assert inertia == tuple(o.compute_inertia()) and warn == 0
    # cone step sizes: identical shrinking counts
    for name, a_g in (("s", step_info["step_size"]), ("t", step_info["step_size_cone_slack_dual"])):
        vec = op[name]
        dv = so[o.index("cone_slack" if name == "s" else "cone_slack_dual") - 1]
        a = 1.0
        while o.cone_violation(vec - a * dv, vec, 0.99):
            a *= 0.5
        if name == "t":
            assert a == a_g
        else:
            assert a_g <= a              # the reported x/r/s step size may have been shortened further by the filter line search
    # the structured K the oracle factored is what the device's banded factorisation represents: the banded linear solve agrees
    rng = np.random.default_rng(3)
    b = rng.standard_normal(o.n)
    s.set("residual_symmetric", b)
    s.linear_solve()
    assert np.abs(s.get("step_symmetric", o.n) - o.linear_solve(b, fact=False)).max() <= 1e-8 * max(1.0, np.abs(o.linear_solve(b, fact=False)).max())


@pytest.mark.parametrize("shape", [STAGED[0], STAGED[1], (16, 24, 16, 3, 1, 3)])
def test_stage_parallel_factorisation_matches_the_blocked_one(shape):
    """calipso_hip_set_stage_parallel: S factored by the multifrontal sparse LDL^T over a nested dissection of its pattern (log2(stages) launches) instead
    of the blocked LDL^T.  Another elimination order: same inertia and decisions, values to rounding."""
    pkg = load_pkg()
    prob, ref = build(pkg, 5, *shape)
    _, par = build(pkg, 5, *shape)
    ref.analyze_structure()
    par.analyze_structure()
    info = par.set_stage_parallel(True)
    T = shape[0]
    assert info["levels"] <= int(np.ceil(np.log2(T))) + 4 and info["largest_front"] <= 196
    assert ref.factorize() == par.factorize()                        # (inertia, regularisation path)
    for it in range(3):
        a = ref.newton_step(advance=True)
        b = par.newton_step(advance=True)
        for key in ("status", "factorizations", "refinement_rounds", "cone_halvings_s", "cone_halvings_t", "line_search_iterations"):
            if key in a:
                assert a[key] == b[key], (it, key, a, b)
        sa, sb = ref.data("step").all, par.data("step").all
        assert np.abs(sa - sb).max() <= 1e-9 * max(1.0, np.abs(sa).max())
        assert np.abs(ref.solution.all - par.solution.all).max() <= 1e-9 * max(1.0, np.abs(ref.solution.all).max())
    # and back: the blocked factorisation again, bit for bit
    par.set_stage_parallel(False)
    par.set("solution", ref.solution.all); par.set("dual", ref.get("dual", ref.ne))
    par.qp_evaluate(pkg.FLAGS["objective"] | pkg.FLAGS["equality_constraint"] | pkg.FLAGS["cone_constraint"], 0)
    a, b = ref.newton_step(advance=False), par.newton_step(advance=False)
    assert a["status"] == b["status"] == 0


def test_group_led_by_a_stage_parallel_handle():
    pkg = load_pkg()
    shape = STAGED[0]
    singles = [build(pkg, p, *shape)[1] for p in (7, 8, 9)]
    members = [build(pkg, p, *shape)[1] for p in (7, 8, 9)]
    for m in singles + members:
        m.analyze_structure()
    for s in singles:
        s.set_stage_parallel(True)
    members[0].set_stage_parallel(True, batch=3)
    g = pkg.Group(members)
    for it in range(2):
        ref = [s.newton_step(advance=True) for s in singles]
        got = g.newton_step(advance=True)
        for a, b, s, m in zip(ref, got, singles, members):
            assert a == b and a["status"] == 0
            assert np.array_equal(s.solution.all, m.solution.all)        # a member gets the bits of the same handle stepped alone
    g.close() if hasattr(g, "close") else None


def test_whole_solve_with_the_three_treatments():
    """solve! (src/solver/solve.jl:8-377) of a stage-structured conic QP evaluated on the device: dense treatment, stage-banded blocked factorisation,
    stage-parallel multifrontal factorisation — the same iteration counts and the same solution (bitwise for the first two)"""
    pkg = load_pkg()
    shape = (16, 24, 16, 3, 1, 3)
    sols, stats = [], []
    for mode in ("dense", "banded", "stage_parallel", "stage_blocks"):
        prob, s = build(pkg, 21, *shape)
        if mode != "dense":
            s.analyze_structure()
        if mode in ("stage_parallel", "stage_blocks"):
            s.set_stage_parallel(True)
        if mode == "stage_blocks":              # packed blocks, block mat-vecs, Schur complement by segment pairs (csrc/blocks.hip)
            assert s.set_stage_blocks(True)["hessian_blocks"] == shape[0]
        assert pkg.solve_b(s)
        st = s.stats()
        sols.append(s.solution.all.copy()); stats.append((st["total_iterations"], st["outer"]))
    assert stats[0] == stats[1] == stats[2] == stats[3], stats
    assert np.array_equal(sols[0], sols[1])
    assert np.abs(sols[2] - sols[0]).max() <= 1e-7 * max(1.0, np.abs(sols[0]).max())
    assert np.abs(sols[3] - sols[0]).max() <= 1e-7 * max(1.0, np.abs(sols[0]).max())


def test_stage_parallel_upload_outside_the_skyline_is_caught_even_without_a_band():
    """A handle small enough that the band covers everything (band_blocks == 0: one 64-wide block) still factors S through its SKYLINE once the
    stage-parallel mode is on; an upload with an entry outside that skyline (zero at analysis time, non-zero later) must send the handle back to
    the dense treatment instead of being dropped from the factorisation."""
    pkg = load_pkg()
    shape = (4, 14, 8, 2, 1, 3)                   # nx = 56: NP = 64, a single block
    prob, s = build(pkg, 11, *shape)
    _, ref = build(pkg, 11, *shape)
    info = s.analyze_structure()
    assert info["band_blocks"] == 0               # nothing for the band treatment to skip ...
    s.set_stage_parallel(True)                    # ... but the multifrontal plan reads S through the skyline only
    nx = prob.nx
    colmajor = lambda M: np.ascontiguousarray(M.T).reshape(-1)
    H = 0.5 * (prob.P + prob.P.T)
    # an upload INSIDE the skyline keeps the mode: same result as the dense handle to rounding
    for h in (s, ref):
        h.set("lagrangian_hessian", colmajor(H))
    a, b = s.newton_step(advance=False), ref.newton_step(advance=False)
    assert a["status"] == b["status"] == 0 and a["factorizations"] == b["factorizations"]
    assert np.abs(s.data("step").all - ref.data("step").all).max() <= 1e-9 * max(1.0, np.abs(ref.data("step").all).max())
    assert s.set_stage_parallel(True)["levels"] >= 1
    # first and last stage coupled: outside the skyline
    H2 = H.copy()
    H2[0, nx - 1] += 0.41; H2[nx - 1, 0] += 0.41
    for h in (s, ref):
        h.set("lagrangian_hessian", colmajor(H2))
    a, b = s.newton_step(advance=False), ref.newton_step(advance=False)
    assert a == b and a["status"] == 0
    assert np.array_equal(s.data("step").all, ref.data("step").all)          # both handles ran the dense launches
    R = ref.data("residual").all
    assert np.abs(R - ref.jacobian_variables_mul(ref.data("step").all)).max() <= 1e-8 * max(1.0, np.abs(R).max())   # the new entries are in the system that was solved


def test_group_larger_than_the_reserved_stage_parallel_batch_falls_back_to_the_blocked_factorisation():
    """calipso_hip_set_stage_parallel(batch = 1) on the leader of a group of three: the multifrontal storage covers one member, so the group step
    falls back to the blocked LDL^T — whose padded rows need their unit pivots although the Schur kernel skipped them for the multifrontal path
    (round-2 defect: NaN steps).  The fallback gives the bits of the banded blocked treatment."""
    pkg = load_pkg()
    shape = (24, 30, 20, 4, 2, 3)                 # nx = 720, NP = 1024: 304 padded rows
    ref = [build(pkg, p, *shape)[1] for p in (3, 4, 5)]
    members = [build(pkg, p, *shape)[1] for p in (3, 4, 5)]
    for m in ref + members:
        m.analyze_structure()
    members[0].set_stage_parallel(True, batch=1)  # too small for the group of three
    g, gref = pkg.Group(members), pkg.Group(ref)
    for it in range(2):
        a, b = gref.newton_step(advance=True), g.newton_step(advance=True)
        for x, y, r, m in zip(a, b, ref, members):
            assert x == y and x["status"] == 0, (it, x, y)
            assert np.all(np.isfinite(m.solution.all)) and np.array_equal(r.solution.all, m.solution.all)
    g.close(); gref.close()


def test_stage_parallel_switched_on_after_a_blocked_step_on_the_same_handle():
    """A blocked factorisation publishes its inertia counts under a sequence number; a stage-parallel factorisation on the SAME handle afterwards must not
    wait for that (long consumed) number — launch_ldl forgets the state of the previous factorisation first (round-3 advisory: this order used to end in
    'scalar read-back did not arrive')."""
    pkg = load_pkg()
    shape = STAGED[1]                                   # NP = 2560: the blocked path takes the two-stream schedule and publishes
    prob, ref = build(pkg, 7, *shape)
    _, s = build(pkg, 7, *shape)
    for h in (ref, s):
        h.analyze_structure()
    a = s.newton_step(advance=False)                    # blocked LDL^T (publishes)
    assert a["status"] >= 0
    s.set_stage_parallel(True)
    b = s.newton_step(advance=False)                    # multifrontal on the same handle
    assert b["status"] >= 0 and b["factorizations"] == a["factorizations"]
    r = ref.newton_step(advance=False)
    assert np.abs(ref.data("step").all - s.data("step").all).max() <= 1e-9 * max(1.0, np.abs(ref.data("step").all).max())
    s.set_stage_parallel(False)
    c = s.newton_step(advance=False)                    # and back to the blocked one
    assert c == r
