"""GPU: stage blocks (csrc/blocks.hip): the blocks of [gx; hx] and of the Lagrangian Hessian of a stage-structured problem packed
contiguously, the mat-vecs of the Newton step and the Schur complement computed block by block.  Checked against the ORACLE (tolerances of SURVEY.md
8(c)), against the dense treatment of the same handle data (to rounding: the sums run block by block), group-vs-single bit for bit, and for the
fallbacks (an upload outside the blocks, a mixed group)."""
import numpy as np
import pytest

import problems as pr
from helpers import load_pkg

pytestmark = pytest.mark.gpu

SHAPES = [
    # T, nv, nd, nonnegative rows / stage, second-order cones / stage, cone dimension
    (12, 40, 30, 4, 2, 3),
    (24, 30, 20, 4, 2, 3),        # nx = 720 (NP = 1024)
    (41, 56, 54, 6, 6, 2),        # BASELINE config C4's size with the structure of a trajectory problem (bench.py: C4T)
    (9, 70, 40, 5, 3, 6),         # stages wider than one 64-column segment; cones of dimension 6 (the wave-per-cone kernels + W blocks in the Schur operand)
    (6, 40, 0, 5, 2, 3),          # no dynamics: block-diagonal S
]


def build(pkg, pid, T, nv, nd, nn, nsoc, dim, blocks=False, stage_parallel=True):
    prob, pt, lam = pr.staged_conic_qp(pkg.splitmix_uniform, pid, T, nv, nd, nn, nsoc, dim)
    s = pkg.Solver(prob, prob.nx, 0, prob.ne, prob.nc, nonnegative_indices=prob.nonnegative_indices, second_order_indices=prob.second_order_indices)
    s.set("solution", np.concatenate([pt[k] for k in "xrsyzt"]))
    s.set("dual", lam)
    for name, v in (("central_path", 0.17), ("penalty", 52.0), ("fraction_to_boundary", 0.99)):
        s.set(name, [v])
    s.qp_attach(prob.P, prob.q, prob.A, prob.b, prob.G, prob.h, 0.5)
    if blocks:
        s.analyze_structure()
        if stage_parallel:
            try:
                s.set_stage_parallel(True)
            except pkg.CalipsoHipError:
                pass                               # fronts too large for LDS: the blocked factorisation stays (blocks work with both)
        info = s.set_stage_blocks(True)
        assert info["hessian_blocks"] == T and info["packed_doubles"] > 0
    fl = pkg.FLAGS
    s.qp_evaluate(fl["objective"] | fl["equality_constraint"] | fl["cone_constraint"] | fl["objective_gradient_variables"] |
                  fl["equality_dual_jacobian_variables"] | fl["cone_dual_jacobian_variables"], 0)
    s.cone(product=True, target=True)
    s.synchronize()
    return prob, s


def close(a, b, tol):
    a, b = np.asarray(a), np.asarray(b)
    return np.abs(a - b).max() <= tol * max(1.0, np.abs(b).max())


@pytest.mark.parametrize("shape", SHAPES)
def test_block_matvecs_and_step_agree_with_the_dense_treatment(shape):
    pkg = load_pkg()
    prob, dense = build(pkg, 5, *shape)
    _, blk = build(pkg, 5, *shape, blocks=True)
    # the evaluator's mat-vecs: f, g, h, fx, (g'y)x, (h'z)x
    for name, n in (("equality_constraint", prob.ne), ("cone_constraint", prob.nc), ("objective_gradient_variables", prob.nx),
                    ("equality_dual_jacobian_variables", prob.nx), ("cone_dual_jacobian_variables", prob.nx)):
        if n:
            assert close(blk.get(name, n), dense.get(name, n), 1e-13), name
    assert abs(blk.scalar("objective") - dense.scalar("objective")) <= 1e-12 * max(1.0, abs(dense.scalar("objective")))
    for it in range(2):
        a, b = dense.newton_step(advance=True), blk.newton_step(advance=True)
        assert a["status"] == b["status"] == 0 and a["refinement_rounds"] == b["refinement_rounds"] and a["factorizations"] == b["factorizations"]
        assert a["step_size"] == b["step_size"] and a["step_size_cone_slack_dual"] == b["step_size_cone_slack_dual"]
        assert close(blk.data("residual").all, dense.data("residual").all, 1e-12)
        assert close(blk.data("step").all, dense.data("step").all, 1e-8)
    assert dense.factorize()[0] == blk.factorize()[0]
    # back to the dense-layout kernels (Lsym, which the blocks had borrowed, is rebuilt): same inertia, same first solve to the bit on the same data
    blk.set_stage_blocks(False)
    blk.clear_structure()
    dense.set("solution", blk.solution.all)
    fl = pkg.FLAGS
    for h in (dense, blk):
        h.qp_evaluate(fl["objective"] | fl["equality_constraint"] | fl["cone_constraint"] | fl["objective_gradient_variables"] |
                      fl["equality_dual_jacobian_variables"] | fl["cone_dual_jacobian_variables"], 0)
        h.cone(product=True, target=True)
        h.residual()
    assert np.array_equal(dense.data("residual").all, blk.data("residual").all)
    assert dense.factorize() == blk.factorize()
    for h in (dense, blk):
        h.search_direction_symmetric(0)
    assert np.array_equal(dense.data("step").all, blk.data("step").all)


@pytest.mark.parametrize("stage_parallel", [True, False])
@pytest.mark.parametrize("shape", [SHAPES[0], SHAPES[1], SHAPES[3]])
def test_block_step_matches_the_oracle(oracle_mod, shape, stage_parallel):
    """one inner Newton iteration with stage blocks on (mat-vecs, Schur complement by segment pairs, multifrontal or blocked LDL^T of S) against the CPU
    restatement of the reference: step, residual, inertia, refinement rounds"""
    pkg = load_pkg()
    prob, s = build(pkg, 11, *shape, blocks=True, stage_parallel=stage_parallel)
    w, lam = s.get("solution", s.N), s.get("dual", s.ne)
    info = s.newton_step(advance=False)
    assert info["status"] == 0
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
    assert close(R, Ro, 1e-12)
    assert close(step, so, 1e-8)
    assert o.stats()["last_refinement_rounds"] == info["refinement_rounds"]
    inertia, warn = s.factorize()
    assert inertia == tuple(o.compute_inertia()) == (prob.nx, prob.ne + prob.nc, 0) and warn == 0
    # a second factorisation of the same matrix (the blocked LDL^T works in place: what it left between the tiles must not leak into the next one)
    assert s.newton_step(advance=False)["status"] == 0
    assert np.array_equal(s.data("step").all, step)


def test_group_with_blocks_is_bitwise_the_single_step_and_mixed_groups_are_refused():
    pkg = load_pkg()
    shape = SHAPES[1]
    singles = [build(pkg, p, *shape, blocks=True)[1] for p in (7, 8, 9)]
    members = [build(pkg, p, *shape, blocks=True)[1] for p in (7, 8, 9)]
    members[0].set_stage_parallel(True, batch=3)
    members[0].set_stage_blocks(True)              # (set_stage_parallel re-analysed nothing, but the leader's blocks are re-made to be safe)
    g = pkg.Group(members)
    for it in range(2):
        ref = [s.newton_step(advance=True) for s in singles]
        got = g.newton_step(advance=True)
        for a, b, s, m in zip(ref, got, singles, members):
            assert a == b and a["status"] == 0
            assert np.array_equal(s.solution.all, m.solution.all)
    g.close()
    # one member without blocks: the group refuses to step
    mixed = [build(pkg, p, *shape, blocks=(p != 8))[1] for p in (7, 8, 9)]
    g2 = pkg.Group(mixed)
    with pytest.raises(pkg.CalipsoHipError, match="agree on calipso_hip_set_stage_blocks"):
        g2.newton_step(advance=False)
    g2.close()


def test_upload_outside_the_blocks_falls_back_to_the_dense_treatment():
    pkg = load_pkg()
    shape = SHAPES[0]
    prob, s = build(pkg, 11, *shape, blocks=True)
    _, ref = build(pkg, 11, *shape)
    nx = prob.nx
    colmajor = lambda M: np.ascontiguousarray(M.T).reshape(-1)
    H = 0.5 * (prob.P + prob.P.T)
    # inside the blocks: the mode stays, the packed copy follows the upload
    H1 = H.copy(); H1[0, 1] += 0.2; H1[1, 0] += 0.2
    for h in (s, ref):
        h.set("lagrangian_hessian", colmajor(H1))
    a, b = s.newton_step(advance=False), ref.newton_step(advance=False)
    assert a["status"] == b["status"] == 0 and close(s.data("step").all, ref.data("step").all, 1e-8)
    assert s.set_stage_blocks(True)["hessian_blocks"] == shape[0]
    # first and last stage coupled: outside every Hessian block
    H2 = H.copy(); H2[0, nx - 1] += 0.41; H2[nx - 1, 0] += 0.41
    for h in (s, ref):
        h.set("lagrangian_hessian", colmajor(H2))
    a, b = s.newton_step(advance=False), ref.newton_step(advance=False)
    assert a["status"] == b["status"] == 0 and close(s.data("step").all, ref.data("step").all, 1e-8)
    R = s.data("residual").all
    assert np.abs(R - s.jacobian_variables_mul(s.data("step").all)).max() <= 1e-8 * max(1.0, np.abs(R).max())     # the new entries are in the system that was solved
    with pytest.raises(pkg.CalipsoHipError, match="analyze_structure first"):      # the analysed structure is gone: the handle is back to the dense treatment
        s.set_stage_blocks(True)
    # a Jacobian row that reaches outside its block
    _, s2 = build(pkg, 11, *shape, blocks=True)
    A = prob.A.copy(); A[0, nx - 1] = 0.5
    for h in (s2, ref):
        h.set("lagrangian_hessian", colmajor(H))
        h.set("equality_jacobian_variables", colmajor(A))
    a, b = s2.newton_step(advance=False), ref.newton_step(advance=False)
    assert a["status"] == b["status"] == 0 and close(s2.data("step").all, ref.data("step").all, 1e-8)
    with pytest.raises(pkg.CalipsoHipError, match="analyze_structure first"):
        s2.set_stage_blocks(True)


def test_a_dense_problem_has_no_blocks():
    pkg = load_pkg()
    prob, pt, lam = pr.synthetic_conic_qp(pkg.splitmix_uniform, 3, 120, 50, 20, 10, 3)
    s = pkg.Solver(prob, prob.nx, 0, prob.ne, prob.nc, nonnegative_indices=prob.nonnegative_indices, second_order_indices=prob.second_order_indices)
    s.qp_attach(prob.P, prob.q, prob.A, prob.b, prob.G, prob.h, 0.5)
    with pytest.raises(pkg.CalipsoHipError):
        s.set_stage_blocks(True)                   # no analysis yet
    s.analyze_structure()
    with pytest.raises(pkg.CalipsoHipError, match="no block structure"):
        s.set_stage_blocks(True)


def build_structured(pkg, pid, T, nv, nd, nn, nsoc, dim):
    """a STRUCTURED handle (calipso_hip_create_structured): no dense Lxx / [gx; hx] / S on the device at all"""
    prob, pt, lam = pr.staged_conic_qp(pkg.splitmix_uniform, pid, T, nv, nd, nn, nsoc, dim)
    s = pkg.Solver(prob, prob.nx, 0, prob.ne, prob.nc, nonnegative_indices=prob.nonnegative_indices, second_order_indices=prob.second_order_indices,
                   structure=pr.declared_structure(prob))
    s.set("solution", np.concatenate([pt[k] for k in "xrsyzt"]))
    s.set("dual", lam)
    for name, v in (("central_path", 0.17), ("penalty", 52.0), ("fraction_to_boundary", 0.99)):
        s.set(name, [v])
    s.qp_attach(prob.P, prob.q, prob.A, prob.b, prob.G, prob.h, 0.5)
    fl = pkg.FLAGS
    s.qp_evaluate(fl["objective"] | fl["equality_constraint"] | fl["cone_constraint"] | fl["objective_gradient_variables"] |
                  fl["equality_dual_jacobian_variables"] | fl["cone_dual_jacobian_variables"], 0)
    s.cone(product=True, target=True)
    s.synchronize()
    return prob, s


@pytest.mark.parametrize("shape", [SHAPES[0], SHAPES[2], SHAPES[4]])
def test_structured_handle_gets_the_bits_of_the_blocks_on_a_dense_handle_in_a_fraction_of_the_memory(shape):
    pkg = load_pkg()
    prob, ref = build(pkg, 5, *shape, blocks=True)
    _, s = build_structured(pkg, 5, *shape)
    assert s.device_bytes() * 8 < ref.device_bytes()                       # (C4T's size: 240 MB -> a few MB)
    # the blocks round-trip through the dense host layout of ProblemData
    cm = lambda M: np.ascontiguousarray(M.T).reshape(-1)
    assert np.array_equal(s.get("lagrangian_hessian", prob.nx ** 2), cm(prob.P))          # Lxx = 2 c P with c = 1/2
    if prob.ne:
        assert np.array_equal(s.get("equality_jacobian_variables", prob.ne * prob.nx), cm(prob.A))
    assert np.array_equal(s.get("cone_jacobian_variables", prob.nc * prob.nx), cm(-prob.G))
    for it in range(3):
        a, b = ref.newton_step(advance=True), s.newton_step(advance=True)
        assert a == b and a["status"] == 0, (it, a, b)
        assert np.array_equal(ref.data("step").all, s.data("step").all)
        assert np.array_equal(ref.solution.all, s.solution.all)
    assert ref.factorize() == s.factorize()
    # H v through the blocks, the dense K for inspection through temporaries
    v = np.random.default_rng(3).standard_normal(s.N)
    assert close(s.jacobian_variables_mul(v), ref.jacobian_variables_mul(v), 1e-12)
    assert close(s.jacobian_variables_symmetric(), ref.jacobian_variables_symmetric(), 1e-12)
    # what a structured handle refuses
    with pytest.raises(pkg.CalipsoHipError):
        s.analyze_structure()
    with pytest.raises(pkg.CalipsoHipError):
        s.clear_structure()
    H = 0.5 * (prob.P + prob.P.T); H[0, prob.nx - 1] = 0.3; H[prob.nx - 1, 0] = 0.3
    with pytest.raises(pkg.CalipsoHipError, match="outside the Hessian blocks"):
        s.set("lagrangian_hessian", cm(H))


def test_structured_group_and_whole_solve():
    pkg = load_pkg()
    shape = SHAPES[1]
    singles = [build_structured(pkg, p, *shape)[1] for p in (7, 8, 9)]
    members = [build_structured(pkg, p, *shape)[1] for p in (7, 8, 9)]
    g = pkg.Group(members)
    for it in range(2):
        ref = [x.newton_step(advance=True) for x in singles]
        got = g.newton_step(advance=True)
        for a, b, x, m in zip(ref, got, singles, members):
            assert a == b and a["status"] == 0
            assert np.array_equal(x.solution.all, m.solution.all)
    g.close()
    # solve! to the end on a structured handle: the iteration counts and the solution of the dense treatment
    shape = (16, 24, 16, 3, 1, 3)
    prob, dense = build(pkg, 21, *shape)
    _, s = build_structured(pkg, 21, *shape)
    assert pkg.solve_b(dense) and pkg.solve_b(s)
    assert (dense.stats()["total_iterations"], dense.stats()["outer"]) == (s.stats()["total_iterations"], s.stats()["outer"])
    assert close(s.solution.all, dense.solution.all, 1e-7)


@pytest.mark.parametrize("members", [64, 128])
def test_large_structured_groups_are_bitwise_the_single_steps(members):
    """Groups of 64 and 128 structured handles (internal.hpp: MAX_BATCH = 128; the members' scalars travel in a device table, group.hip): every member gets
    the bits of the same handle stepped alone — sampled members over two advancing steps, so the table is re-used, then replaced when scalars move."""
    pkg = load_pkg()
    shape = SHAPES[0]
    ids = list(range(100, 100 + members))
    mem = [build_structured(pkg, p, *shape)[1] for p in ids]
    sample = sorted({0, 1, 31, 32, 33, members // 2, members - 2, members - 1})
    singles = {k: build_structured(pkg, ids[k], *shape)[1] for k in sample}
    # members with different scalars (the table holds them per member)
    for k in (1, members - 1):
        for h in (mem[k], singles[k]):
            h.set("central_path", [0.05]); h.set("penalty", [7.0])
    g = pkg.Group(mem)
    for it in range(2):
        got = g.newton_step(advance=True)
        for k in sample:
            ref = singles[k].newton_step(advance=True)
            assert ref == got[k] and ref["status"] == 0, (it, k, ref, got[k])
            assert np.array_equal(singles[k].data("step").all, mem[k].data("step").all)
            assert np.array_equal(singles[k].solution.all, mem[k].solution.all)
    g.close()
    with pytest.raises(pkg.CalipsoHipError):
        pkg.Group(mem + [build_structured(pkg, 999, *shape)[1]] * (129 - members))       # 129 members: above the limit


def test_group_of_forty_dense_members_is_bitwise_the_single_steps():
    pkg = load_pkg()
    from test_gpu_group import build as build_dense, same
    shape = (300, 120, 30, 10, 3)
    ids = list(range(200, 240))
    mem = [build_dense(pkg, p, shape) for p in ids]
    sample = [0, 17, 31, 32, 39]
    singles = {k: build_dense(pkg, ids[k], shape) for k in sample}
    g = pkg.Group(mem)
    for it in range(2):
        got = g.newton_step(advance=True)
        for k in sample:
            ref = singles[k].newton_step(advance=True)
            assert ref == got[k] and ref["status"] == 0
            assert same(singles[k].data("step").all, mem[k].data("step").all) and same(singles[k].solution.all, mem[k].solution.all)
    g.close()
