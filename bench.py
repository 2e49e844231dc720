#!/usr/bin/env python3
"""bench.py — Newton steps/s of the CALIPSO KKT hot path on MI355X (BASELINE.json metric).

Headline (`value`): ONE n = 5000 KKT system per GPU stepped sequentially — "Newton steps/sec (n~5k KKT) at 1 GPU".
A "step" = one inner Newton iteration of solve! (src/solver/solve.jl:98-353): evaluate (QP mat-vecs on the device) -> cone! ->
residual! -> inertia-corrected LDL^T of the condensed KKT matrix -> condensed solve + step recovery -> >= 1 refinement round
against the unreduced system -> cone fraction-to-boundary search -> candidate merit / violation -> filter line-search decision.
Inputs are resident in HBM before the timed region.  Workload = BASELINE config C3 (synthetic dense conic QP, nx=2500, ne=1500,
nc=400 R+ + 200 x SOC3 => n = 5000 condensed, N = 8500 unreduced; SplitMix64 streams); problem id = rank.

Batched figure (`config.batched`, BASELINE's "batched problems/sec"): B independent instances per GPU (default 36 = 3 groups of 12;
a group steps its members in lockstep through the same kernel launches, three groups in flight on three HIP streams), measured in
a second timed region of the same run; aggregated over ranks.

  python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run, one rank per GPU)

Multi-GPU: problems are independent => ranks share nothing on the data path (a single system is "replicas only", the batched path
is sharded block-contiguously: weak scaling); torch.distributed (RCCL) is used for the barrier, the max-over-ranks time and the
post-round gather of status rows / counters.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

CONFIGS = {
    # name: (nx, ne, n_nonneg, n_soc, soc_dim)
    "C3": (2500, 1500, 400, 200, 3),
    "C4": (2302, 2208, 244, 240, 2),
    "small": (600, 300, 100, 50, 3),
}
# stage-structured variants (tests/problems.py: staged_conic_qp): name -> (T, nv, nd, nonnegative rows / stage, SOCs / stage, SOC dim).
# C4T has the size of C4 (nx = 2296, ne = 2160, nc = 738) with the block structure of a 41-stage trajectory problem; the handle
# analyses the pattern (calipso_hip_analyze_structure) unless --dense-structure is given.
STAGED = {
    "C4T": (41, 56, 54, 6, 6, 2),
    "smallT": (12, 40, 30, 4, 2, 3),
}
FP64_MFMA_PEAK_TFLOPS = 78.6   # MI355X fp64 matrix peak (datasheet); the measured ceiling is reported beside it (calipso_hip_mfma_f64_peak)


def make_instance(pkg, pr, pid, shape, device, staged=None, analyze=True, structured=False):
    nx, ne, n_nn, n_soc, dim = shape
    if staged is not None:
        prob, pt, lam = pr.staged_conic_qp(pkg.splitmix_uniform, pid, *staged)
    else:
        prob, pt, lam = pr.synthetic_conic_qp(pkg.splitmix_uniform, pid, nx, ne, n_nn, n_soc, dim)
    # structured handle (calipso_hip_create_structured): the sparsity is declared up front, only the stage blocks live on the device
    s = pkg.Solver(prob, prob.nx, 0, prob.ne, prob.nc, nonnegative_indices=prob.nonnegative_indices,
                   second_order_indices=prob.second_order_indices, device=device, structure=pr.declared_structure(prob) if structured else None)
    w = np.concatenate([pt[k] for k in "xrsyzt"])
    s.set("solution", w)
    s.set("dual", lam)
    for name, v in (("central_path", 0.17), ("penalty", 52.0), ("fraction_to_boundary", 0.99)):
        s.set(name, [v])
    s.qp_attach(prob.P, prob.q, prob.A, prob.b, prob.G, prob.h, 0.5)
    if staged is not None and analyze and not structured:
        s.analyze_structure()
    fl = pkg.FLAGS
    s.qp_evaluate(fl["objective"] | fl["equality_constraint"] | fl["cone_constraint"], 0)
    s.cone(product=True, target=True)
    s.synchronize()
    return prob, pt, lam, w, s


def staged_shape(st):
    T, nv, nd, nn, nsoc, dim = st
    return (T * nv, (T - 1) * nd, T * nn, T * nsoc, dim)


def _oracle_mod():
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle
    return oracle


def cpu_solve_baseline(prob, opts=None, reps=3, differentiate_reps=0):
    """The oracle (oracle/: the repo's single-thread C++ restatement of the reference's CPU path — kind "port", NOT the Julia reference) on the SAME problem, same options,
    same host evaluation functions (tests/problems.py), on this box's host: one solve! (median of `reps`), and differentiate! alone where asked.  Reported next to the
    GPU figure whichever of the two wins."""
    oracle = _oracle_mod()

    def fresh():
        o = oracle.OracleSolver(prob.nx, prob.np, prob.ne, prob.nc, prob.nonnegative_indices, prob.second_order_indices)
        for k, v in (opts or {}).items():
            (o.set_opt if isinstance(v, float) else o.set_int)(k, v)
        if prob.np:
            o.buf("parameters")[:] = prob.parameters
        o.point()["x"][:] = prob.x0
        return o
    ts, st, its, o = [], 0, 0, None
    for _ in range(max(1, reps)):
        o = fresh()
        t0 = time.perf_counter()
        st = o.solve(prob)
        ts.append(1e3 * (time.perf_counter() - t0))
        its = o.stats()["total_iterations"]
    out = {"kind": "port", "cores": 1, "unit": "ms per solve!", "value": float(np.median(ts)), "solve_ms_all": ts, "solved": bool(st == 1), "newton_iterations": int(its),
           "ms_per_newton_iteration": float(np.median(ts)) / max(1, int(its)),
           "sample": "%d solve!s of the same problem by the oracle on one host core (evaluation = the same Python functions the GPU path calls back into)" % len(ts)}
    if differentiate_reps > 0:
        o.differentiate(prob)
        t0 = time.perf_counter()
        for _ in range(differentiate_reps):
            o.differentiate(prob)
        dms = 1e3 * (time.perf_counter() - t0) / differentiate_reps
        out["differentiate_ms"] = dms
        out["back_solves_per_s"] = prob.np / (dms * 1e-3)
    return out


def cpu_step_baseline(shape, staged=None, samples=1):
    """B0(i) alone (one Newton step of problem 0 by the oracle, ONE factorisation, one host core) for a configuration other than the headline's: config.c4"""
    oracle = _oracle_mod()
    import problems as pr
    nx, ne, n_nn, n_soc, dim = shape
    if staged is not None:
        prob, pt, lam = pr.staged_conic_qp(oracle.splitmix_uniform, 0, *staged)
    else:
        prob, pt, lam = pr.synthetic_conic_qp(oracle.splitmix_uniform, 0, nx, ne, n_nn, n_soc, dim)
    o = oracle.OracleSolver(prob.nx, 0, prob.ne, prob.nc, prob.nonnegative_indices, prob.second_order_indices)
    o.point()["all"][:] = np.concatenate([pt[k] for k in "xrsyzt"])
    o.buf("dual")[:] = lam
    o.buf("central_path")[0] = 0.17
    o.buf("penalty")[0] = 52.0
    o.set_int("linear_solve_refactor", 0)
    ts, rc = [], 0
    for _ in range(max(1, samples)):
        t0 = time.perf_counter()
        prob.evaluate(pr.ALL_VARIABLE_FLAGS, pt["x"], pt["y"], pt["z"], np.zeros(0), o.buf)
        o.cone(product=True, jacobian=True, target=True, barrier=True, barrier_gradient=True)
        o.residual()
        rc = o.search_direction()
        ts.append(time.perf_counter() - t0)
    t = float(np.median(ts))
    return {"kind": "port", "cores": 1, "unit": "Newton steps/s", "value": 1.0 / t, "seconds_per_step": t, "status": int(rc),
            "sample": "%d Newton step(s) of problem 0 by the oracle, one factorisation per step (favourable to the reference)%s" % (
                len(ts), "" if staged is None else "; the port assembles and factors the stage blocks as DENSE blocks — the reference's sparse LDL^T would exploit the stage structure, "
                "so this row understates what the reference's CPU path does on a trajectory problem")}


def config_c3_solve(pkg, pr, device, shape, cpu=True):
    """A REAL solve! of one C3-shaped problem (advancing iterates: central-path / penalty updates, inertia-correction retries, line searches — not the repeated
    non-advancing step of the headline): solve.jl:8-377 on the device with the attached QP evaluator, cold start (initialize_slacks! / initialize_duals!)."""
    nx, ne, n_nn, n_soc, dim = shape
    prob, pt, lam = pr.synthetic_conic_qp(pkg.splitmix_uniform, 1, nx, ne, n_nn, n_soc, dim)
    out = {"workload": "one solve! (cold start, default options) of C3 problem 1: nx = %d, ne = %d, nc = %d" % (prob.nx, prob.ne, prob.nc)}
    ms, st = [], None
    for rep in range(2):                    # (the first solve pays the plan scan / first-use allocations of a fresh handle)
        s = pkg.Solver(prob, prob.nx, 0, prob.ne, prob.nc, nonnegative_indices=prob.nonnegative_indices, second_order_indices=prob.second_order_indices, device=device)
        s.qp_attach(prob.P, prob.q, prob.A, prob.b, prob.G, prob.h, 0.5)
        pkg.initialize_b(s, pt["x"])
        s.synchronize()
        t0 = time.perf_counter()
        try:
            ok = pkg.solve_b(s)
            err = None
        except pkg.CalipsoHipError as e:
            ok, err = False, str(e)
        s.synchronize()
        ms.append(1e3 * (time.perf_counter() - t0))
        st = s.stats()
        kt = s.kernel_times()
        s.__del__()
    out.update({"solved": bool(ok), "error": err, "solve_ms": ms[-1], "solve_ms_first": ms[0], "newton_iterations": int(st["total_iterations"]), "newton_steps_with_a_direction": int(st["newton_steps"]),
                "outer_iterations": int(st["outer"]), "factorizations": int(st["factorizations"]), "max_refinement_rounds": int(st["max_refinement_rounds"]),
                "refinement_failures": int(st["refinement_failures"]), "lu_fallbacks": int(st["fallbacks"]),
                "newton_steps_per_s": st["newton_steps"] / (ms[-1] * 1e-3) if ms[-1] > 0 else None, "left_looking_schedule": int(kt[6])})
    return out


def config_small_newton(pkg, pr, device, cpu=True, batch=4096, steps=20):
    """The batched small-problem path (csrc/smallnewton.hip, calipso_hip_smallnewton_*): `batch` QPs of C5's shape (nx = 49, ne = 40: the cart-pole MPC problem of
    examples/autotuning/cartpole.jl:85-146 is that size; here random strictly convex QPs, test/solver/problem.jl:3-23) — whole solve!s in ONE launch, and `steps`
    non-advancing Newton steps per instance in one launch (the headline's step, for the batch).  The oracle solves a sample of the same problems on one host core."""
    nx, ne, nc = 49, 40, 0
    nprob = 32
    probs = [pr.random_qp(nx, ne, nc, seed=1000 + k, nonnegative_indices=[]) for k in range(nprob)]
    idx = np.arange(batch) % nprob
    st = lambda name: np.stack([np.asarray(getattr(probs[i], name), dtype=np.float64) for i in idx])
    sn = pkg.SmallNewtonBatch(nx, ne, nc, batch, device=device)
    sn.set_qp(st("P"), st("q"), st("A"), st("b"), st("G"), st("h"), objective_scale=probs[0].c, shared=False)
    rng = np.random.default_rng(0)
    x0 = np.stack([probs[i].x0 for i in idx]) + 0.01 * rng.standard_normal((batch, nx))
    ms_all = []
    for _ in range(3):
        sn.initialize(x0)
        res, ms = sn.solve()
        ms_all.append(ms)
    stt = sn.get_state()
    its, nst = stt["counters"]["total_iterations"], stt["counters"]["newton_steps"]
    ms = min(ms_all)
    import ctypes as C
    from calipso_jl_amd._lib import lib
    dsc = np.zeros(4); fdsc = lib().calipso_hip_debug_smallnewton_describe; fdsc.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
    fdsc(sn._h, dsc.ctypes.data_as(C.POINTER(C.c_double)))
    out = {"workload": "%d random convex QPs of C5's shape (nx = %d, ne = %d, n = %d), one workgroup per instance, one launch" % (batch, nx, ne, nx + ne),
           "kernel": {"threads_per_instance": int(dsc[0]), "lds_bytes_per_instance": int(dsc[1]), "instances_per_compute_unit": int(dsc[2]), "compute_units": int(dsc[3])},
           "solve": {"launch_ms": ms, "converged": int((res == 1).sum()), "solves_per_s": batch / (ms * 1e-3), "newton_steps_per_s": float(nst.sum()) / (ms * 1e-3),
                     "mean_newton_iterations": float(its.mean()), "max_refinement_rounds": int(stt["counters"]["max_refinement_rounds"].max())}}
    # differentiate! of the whole batch at its solutions, one launch (differentiate.jl:1-61): C5's 102 parameter columns, dR/dtheta the same for all instances
    try:
        N, pcol = nx + 2 * ne + 3 * nc, 102
        Jp = np.zeros((N, pcol)); Jp[:nx, :nx] = np.eye(nx); Jp[nx + ne + nc:nx + 2 * ne + nc, nx:nx + ne] = -np.eye(ne); Jp[:nx, nx + ne:] = rng.standard_normal((nx, pcol - nx - ne))
        msd, ok = 1e30, 0
        for _ in range(3):
            Sens, std, m1 = sn.differentiate(Jp)
            msd = min(msd, m1); ok = int((std == 0).sum())
        out["differentiate"] = {"columns": pcol, "launch_ms": msd, "differentiates_per_s": batch / (msd * 1e-3), "back_solves_per_s": batch * pcol / (msd * 1e-3), "inertia_ok": ok,
                                "finite": bool(np.isfinite(Sens).all()), "note": "kernel time (HIP events); every column refined (iterative_refinement.jl:1-52); the general path's differentiate! of ONE C5 "
                                "problem: config.c5 (GPU 0.6 ms, oracle 4.9 ms)"}
        del Sens
    except Exception as e:
        out["differentiate"] = {"error": repr(e)}
    w = stt["solution"].copy()
    w[:, :nx] += 0.05 * rng.standard_normal((batch, nx))
    sn.set_state(w=w, scalars=np.tile([0.17, 0.99, 52.0], (batch, 1)))
    sn.steps(2, advance=False)
    msk = min(sn.steps(steps, advance=False)[2] for _ in range(3))
    info, stat, _ = sn.steps(1, advance=False)
    bytes_inst = 8.0 * (nx * nx + ne * nx + nx + ne + 2 * (nx + 2 * ne))
    out["steps"] = {"count_per_instance": steps, "launch_ms": msk, "newton_steps_per_s": batch * steps / (msk * 1e-3), "stepped": int(((stat == 0) & (info[:, 6] == 0)).sum()),
                    "refinement_rounds": float(info[:, 2].mean()),
                    "roofline": {"bound": "latency (neither HBM nor MFMA: ~75 dependent workgroup phases per step; throughput = resident instances / latency of one, %d instances per compute unit)" % int(dsc[2]),
                                 "us_per_step_of_a_resident_instance": msk * 1e3 / steps / max(1.0, batch / max(1.0, dsc[2] * dsc[3])),
                                 "hbm_frac": batch * bytes_inst / (msk * 1e-3) / 8e12, "note": "an instance's data is read once per LAUNCH (%.0f KB), not per step" % (bytes_inst / 1e3)}}
    sn.close()
    if cpu:
        try:
            oracle = _oracle_mod()
            ts = []
            for k in range(6):
                o = oracle.OracleSolver(nx, 0, ne, nc, probs[k].nonnegative_indices, probs[k].second_order_indices)
                o.point()["x"][:] = probs[k].x0
                t0 = time.perf_counter(); stc = o.solve(probs[k]); ts.append(time.perf_counter() - t0)
            out["cpu_baseline"] = {"kind": "port", "cores": 1, "unit": "solve!s/s", "value": 1.0 / float(np.median(ts)), "solve_ms": 1e3 * float(np.median(ts)),
                                   "sample": "%d solve!s of the same problems by the oracle, one host core, evaluation through Python callbacks" % len(ts)}
            out["solve"]["gpu_over_cpu"] = out["solve"]["solves_per_s"] / out["cpu_baseline"]["value"]
        except Exception as e:
            out["cpu_baseline"] = {"error": repr(e)}
    return out


def config_c2(pkg, pr, device, cpu=True):
    """BASELINE config 2: the pendulum swing-up (T = 11; test/examples/pendulum.jl) as ONE full solve!: every inner Newton iteration of solve.jl:98-353 on the device,
    the Symbolics-generated evaluate! replaced by the restated problem functions on the host (callback: tests/problems.py).  Iterations are checked against the
    oracle-made golden trace (tests/golden/c2_pendulum_trace.npz)."""
    prob = pr.pendulum(action_guess=np.zeros(10))
    ms = []
    its = 0
    accepted = []
    for rep in range(4):
        s = pkg.Solver(prob, prob.nx, prob.np, prob.ne, prob.nc, device=device)
        accepted = []
        if rep == 0:      # (the untimed first solve counts the accepted iterates the way the golden trace was recorded: one callback_inner call each)
            s.set_callbacks(inner=lambda sv: accepted.append(1))
        pkg.initialize_b(s, prob.x0)
        s.synchronize()
        t0 = time.perf_counter()
        ok = pkg.solve_b(s)
        s.synchronize()
        ms.append(1e3 * (time.perf_counter() - t0))
        its = int(s.stats()["total_iterations"])
        if rep == 0: trace_rows = len(accepted)
        sol = s.get("solution", s.N)
        s.close() if hasattr(s, "close") else None
    out = {"workload": "pendulum swing-up T = 11 (nx = %d, ne = %d, nc = %d), one solve! with host evaluation callbacks" % (prob.nx, prob.ne, prob.nc), "solved": bool(ok),
           "newton_iterations": its, "solve_ms": float(np.median(ms[1:])), "solve_ms_all": ms, "ms_per_newton_iteration": float(np.median(ms[1:])) / max(1, its),
           "note": "a 56 x 56 condensed system: the time is launch latency + host callbacks, not device throughput (the batched LDS-resident path of config.c5 is what many such systems take)"}
    try:
        g = np.load(os.path.join(ROOT, "tests", "golden", "c2_pendulum_trace.npz"))
        out["accepted_iterates"] = trace_rows
        out["golden_trace_accepted_iterates"] = int(g["trace"].shape[0])
        out["same_iterations_as_golden_trace"] = bool(trace_rows == int(g["trace"].shape[0]))
        out["matches_golden_solution_1e-6"] = bool(np.abs(sol - g["solution"]).max() <= 1e-6 * max(1.0, np.abs(g["solution"]).max()))
    except Exception as e:      # (fixture missing: say so, do not fail the bench)
        out["golden"] = "unavailable: %s" % e
    if cpu:
        try:
            out["cpu_baseline"] = cpu_solve_baseline(prob)
            out["gpu_over_cpu"] = out["cpu_baseline"]["value"] / out["solve_ms"]      # > 1: the GPU path is faster; a 56 x 56 system is launch latency against arithmetic in cache
        except Exception as e:
            out["cpu_baseline"] = {"error": repr(e)}
    return out


def config_c5(pkg, pr, device, cpu=True):
    """BASELINE config 5: cart-pole auto-tuning sensitivities dw*/dtheta (examples/autotuning/cartpole.jl:179-227, src/solver/differentiate.jl:1-61): nx = 49, ne = 40,
    102 parameter columns.  (a) differentiate! on one handle (dense and declared as the trajectory problem it is); (b) the batched LDS-resident path (csrc/small.hip:
    calipso_hip_small_*) for the back-solves of 1024 MPC steps at once, priced against HBM (its inputs and outputs are read / written once)."""
    prob = pr.cartpole_mpc()
    opts = dict(residual_tolerance=1e-3, optimality_tolerance=1e-3, equality_tolerance=1e-3, complementarity_tolerance=1e-3, slack_tolerance=1e-3, differentiate=1)
    out = {"workload": "cart-pole MPC sensitivities: nx = %d, ne = %d, %d parameter columns" % (prob.nx, prob.ne, prob.np)}
    for name, st in (("dense_handle", None), ("structured_handle", pr.structure_from_pattern(prob))):
        s = pkg.Solver(prob, prob.nx, prob.np, prob.ne, prob.nc, parameters=prob.parameters, options=opts, structure=st, device=device)
        pkg.initialize_b(s, prob.x0)
        ok = pkg.solve_b(s)
        s.differentiate(); s.synchronize()
        t0 = time.perf_counter()
        reps = 20
        for _ in range(reps):
            s.differentiate()
        s.synchronize()
        dms = 1e3 * (time.perf_counter() - t0) / reps
        out[name] = {"solved": bool(ok), "newton_iterations": int(s.stats()["total_iterations"]), "differentiate_ms": dms, "back_solves_per_s": prob.np / (dms * 1e-3),
                     "device_bytes": int(s.device_bytes())}
    if cpu:
        try:
            out["cpu_baseline"] = cpu_solve_baseline(prob, opts, reps=2, differentiate_reps=5)
            out["gpu_over_cpu_differentiate"] = out["cpu_baseline"]["differentiate_ms"] / out["dense_handle"]["differentiate_ms"]
        except Exception as e:
            out["cpu_baseline"] = {"error": repr(e)}
    n, nrhs, batch = prob.nx + prob.ne + prob.nc, prob.np, 1024
    rng = np.random.default_rng(0)
    Q = rng.standard_normal((n, n))
    K = Q @ Q.T + n * np.eye(n)
    K[prob.nx:, prob.nx:] = -K[prob.nx:, prob.nx:]; K[:prob.nx, prob.nx:] *= 0.1; K[prob.nx:, :prob.nx] = K[:prob.nx, prob.nx:].T
    Bm = rng.standard_normal((batch, n, nrhs))
    sb = pkg.SmallBatch(n, nrhs, batch)
    sb.set(np.repeat(K[None], batch, axis=0), Bm)
    sb.solve()
    ms = min(sb.solve() for _ in range(5))
    X, inr, bad = sb.get()
    okb = bad == 0 and float(np.abs(K @ X[batch - 1] - Bm[batch - 1]).max()) < 1e-8
    sb.close()
    bytes_io = 8.0 * batch * (n * n + 2 * n * nrhs)
    out["batched_small_systems"] = {"n": n, "right_hand_sides": nrhs, "batch": batch, "launch_ms": ms, "instances_per_s": batch / (ms * 1e-3), "back_solves_per_s": batch * nrhs / (ms * 1e-3),
                                    "residual_ok": bool(okb),
                                    "roofline": {"bound": "hbm", "bytes": bytes_io, "achieved_GBs": bytes_io / (ms * 1e-3) * 1e-9, "frac": bytes_io / (ms * 1e-3) * 1e-9 / 8000.0,
                                                 "flops": batch * (n ** 3 / 3.0 + 2.0 * n * n * nrhs), "note": "one workgroup per instance, the matrix and its right-hand sides in LDS: K, B read once, X written once; "
                                                 "the work per instance is a dependent chain of %d pivots — latency, not bandwidth, is what bounds a launch" % n}}
    return out


def cpu_baseline(shape, name="C3", staged=None, samples=3, full=False):
    """CPU rows of SURVEY.md 8(d) on the GPU box's host, same C3 problem 0 (ONE Newton step each):
      B0(i)   the oracle (faithful single-thread restatement of the reference's CPU path: assemble + sparse up-looking LDL^T in QDLDL's
              operation order + solves + refinement) with ONE factorisation per step — favourable to the reference; `value`, 3 samples
      B0(ii)  the reference's real behaviour: every linear_solve! re-factorises (linear_solver.jl:52-57, fact=true): 1 + (1 + n_r)
              factorisations per step.  Default: B0(i) + (1 + n_r) x the separately timed factorisation; --cpu-baseline-full runs it
      B1      NOT the reference: LAPACK dsytrf/dsytrs (Bunch-Kaufman) of the dense condensed K on all host cores — the strongest CPU
              baseline the box offers for the factor + solves part."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle
    import problems as pr
    nx, ne, n_nn, n_soc, dim = shape
    if staged is not None:
        prob, pt, lam = pr.staged_conic_qp(oracle.splitmix_uniform, 0, *staged)
    else:
        prob, pt, lam = pr.synthetic_conic_qp(oracle.splitmix_uniform, 0, nx, ne, n_nn, n_soc, dim)

    def fresh(refactor):
        o = oracle.OracleSolver(prob.nx, 0, prob.ne, prob.nc, prob.nonnegative_indices, prob.second_order_indices)
        o.point()["all"][:] = np.concatenate([pt[k] for k in "xrsyzt"])
        o.buf("dual")[:] = lam
        o.buf("central_path")[0] = 0.17
        o.buf("penalty")[0] = 52.0
        o.set_int("linear_solve_refactor", refactor)
        return o

    def one_step(o):
        t0 = time.perf_counter()
        prob.evaluate(pr.ALL_VARIABLE_FLAGS, pt["x"], pt["y"], pt["z"], np.zeros(0), o.buf)
        o.cone(product=True, jacobian=True, target=True, barrier=True, barrier_gradient=True)
        o.residual()
        rc = o.search_direction()
        return time.perf_counter() - t0, rc

    o = fresh(0)
    times, rc = [], 0
    for _ in range(max(1, samples)):
        dt, rc = one_step(o)
        times.append(dt)
    st = o.stats()
    n_r = st["last_refinement_rounds"]
    t0 = time.perf_counter()
    o.factorize(update=True)                       # one more factorisation of the same matrix, timed alone
    t_fact = time.perf_counter() - t0
    t_i = float(np.median(times))
    extra = 1 + n_r                                # hidden re-factorisations: one per linear_solve! (first solve + n_r refinement solves)
    # (not derived any more: a figure that was not measured does not go into the line; one factorisation timed alone says what it would cost)
    b0ii = dict(value=None, measured=False, factorizations_per_step=1 + extra, one_factorisation_s=t_fact,
                how="not run by default (%d factorisations of %.1f s per step): --cpu-baseline-full measures it" % (1 + extra, t_fact))
    if full:
        of = fresh(1)
        dt, _ = one_step(of)
        b0ii = dict(value=1.0 / dt, unit="Newton steps/s", cores=1, factorizations_per_step=int(of.stats()["factorizations"]), measured=True,
                    how="one full step with linear_solve_refactor = 1 (%.1f s)" % dt)
    # B1: LAPACK on all cores, factor + (1 + n_r) solves of the dense condensed K (upper triangle symmetrised, as QDLDL sees it)
    b1 = None
    try:
        from scipy.linalg import lapack
        o.residual_jacobian_variables(); o.residual_jacobian_variables_symmetric()
        K = np.array(o.K_dense(), order="F")
        K = np.triu(K) + np.triu(K, 1).T
        b = np.array(o.buf("residual_symmetric"))
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            ldu, ipiv, info = lapack.dsytrf(K, lower=0)
            for _k in range(1 + n_r):
                x, info2 = lapack.dsytrs(ldu, ipiv, b, lower=0)
            ts.append(time.perf_counter() - t0)
        try:
            import threadpoolctl
            thr = max([p.get("num_threads", 1) for p in threadpoolctl.threadpool_info() if p.get("user_api") == "blas"] or [os.cpu_count()])
        except Exception:
            thr = os.cpu_count()
        b1 = dict(value=1.0 / float(np.median(ts)), unit="factor+solves/s", cores=int(thr), kind="lapack dsytrf/dsytrs (not the reference)",
                  sample="dense K n=%d, 1 dsytrf + %d dsytrs, median of 3: %.3f s; excludes assembly / residuals / refinement mat-vecs" % (
                      K.shape[0], 1 + n_r, float(np.median(ts))))
    except Exception as e:   # pragma: no cover
        b1 = dict(error=repr(e))
    # B2: the reference itself, only where a julia with CALIPSO's dependencies exists (none in this project's containers)
    import shutil
    import subprocess
    b2 = "julia unavailable"
    if staged is None and shutil.which("julia") and os.environ.get("CALIPSO_JL_PROJECT"):
        try:
            r = subprocess.run(["julia", "--project=" + os.environ["CALIPSO_JL_PROJECT"], os.path.join(ROOT, "bench", "ref_julia.jl")] + [str(v) for v in shape],
                               capture_output=True, text=True, timeout=3600)
            b2 = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
        except Exception as e:   # pragma: no cover
            b2 = "julia run failed: %r" % (e,)
    note = "" if staged is None else ("(the port assembles and factors the blocks densely: it does not exploit the stage structure, which the "
                                      "reference's sparse LDL^T would) ")
    return dict(value=1.0 / t_i, unit="Newton steps/s", cores=1, kind="port",
                sample=note + "B0(i): %d sample(s) of 1 Newton step (evaluate + cone + residual + search_direction: 1 LDL^T factorisation, %d solves) of "
                       "%s problem 0, median %.1f s, all %s s" % (len(times), 1 + n_r, name, t_i, ["%.1f" % t for t in times]),
                samples_s=times, status=int(rc), host_cores=os.cpu_count(),
                B0_ii_reference_refactorisation=b0ii, B1_lapack_all_cores=b1, B2_julia_reference=b2)


def free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def spawn_command(n, argv):
    """the launch `python bench.py --gpus N` turns itself into when it is not already a rank of a torch.distributed job"""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
            "--master-port", str(free_port()), os.path.abspath(__file__)] + list(argv)


def spawn_ranks(args):
    """`python bench.py --gpus N` with N > 1 and no rank environment: re-exec under torch.distributed.run, one rank per GPU (RCCL over xGMI),
    so that the plain command really runs N ranks and prints n_gpus = N.  Under torchrun (RANK / WORLD_SIZE set) this is a no-op."""
    if args.gpus <= 1 or "WORLD_SIZE" in os.environ or "RANK" in os.environ:
        return
    if args.force_device < 0 and not args.spawn_check:
        import torch
        have = torch.cuda.device_count()
        if have < args.gpus:
            sys.exit("bench.py: --gpus %d asked for, %d HIP device(s) visible" % (args.gpus, have))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = spawn_command(args.gpus, sys.argv[1:])
    sys.stdout.flush(); sys.stderr.flush()
    os.execve(sys.executable, cmd, env)


class Exchange:
    """The post-round exchange (outside the data path): per-problem status rows all-gathered in global problem-id order, step counters
    all-reduced.  With RCCL ranks (and with one rank) it goes through the PRODUCT's C entry points calipso_hip_comm_* (csrc/comm.hip:
    ncclAllGather / ncclAllReduce); with the gloo backend (CPU-side tests: two ranks sharing one GPU cannot form an RCCL communicator)
    through torch.distributed (calipso.jl_amd/batch.py: gather_results)."""

    def __init__(self, pkg, dist, backend, rank, world, device, try_comm=None):
        self.pkg, self.dist, self.rank, self.world = pkg, dist, rank, world
        self.comm, self.path = None, "torch.distributed (%s)" % backend
        self.ranks_reported = None          # the size the PRODUCT's RCCL communicator itself reports (ncclCommCount); None: the exchange went through torch.distributed
        if try_comm is None:
            try_comm = world == 1 or backend == "nccl"
        if try_comm:
            uid = [pkg.Comm.unique_id() if rank == 0 else None]
            if world > 1:
                dist.broadcast_object_list(uid, src=0)
            ok, why = 1, ""
            try:
                self.comm = pkg.Comm(rank, world, uid[0], device=device)
            except Exception as e:                                    # (the exchange is outside the data path: never let it take the measurement down)
                ok, why = 0, str(e)
            if world > 1:                                             # every rank takes the same path
                import torch
                flag = torch.tensor([ok], dtype=torch.int32, device=("cuda:%d" % device) if backend == "nccl" else "cpu")
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                ok = int(flag.item())
            if ok:
                try:
                    self.ranks_reported = int(self.comm.size()[0])
                except Exception:
                    self.ranks_reported = None
                self.path = "calipso_hip_comm_gather_status / calipso_hip_comm_allreduce_sum (RCCL ncclAllGather / ncclAllReduce, csrc/comm.hip)"
            else:
                self.close()
                self.path += "; the product's communicator could not be created on every rank (%s)" % (why or "another rank failed")

    def gather(self, status, counters):
        status = np.ascontiguousarray(status, dtype=np.int32).reshape(-1, 4)
        if self.comm is not None:
            rows, counts = self.comm.gather_status(status, self.world * max(1, status.shape[0]) + 16)
            return rows, self.comm.allreduce_sum(counters), counts
        from calipso_jl_amd.batch import gather_results
        rows, tot = gather_results(status, counters)
        return rows, tot, None

    def close(self):
        if self.comm is not None:
            self.comm.close()
            self.comm = None


class Workload:
    """B independent instances of one configuration on this rank's GPU.  Instance 0 is the rank's single system (headline region); the B
    instances form B / G groups of G (the members of a group are stepped in lockstep through the same launches), `lanes` groups in flight."""

    def __init__(self, pkg, pr, name, rank, world, device, B, G, lanes, dense_structure=False, no_stage_parallel=False, no_stage_blocks=False, dense_buffers=False):
        from calipso_jl_amd.batch import BatchSolver, shard_range
        self.pkg, self.name, self.B, self.G, self.world = pkg, name, max(0, B), max(1, G), world
        self.staged = STAGED.get(name)
        self.shape = staged_shape(self.staged) if self.staged else CONFIGS[name]
        assert self.B % self.G == 0, "--batch must be a multiple of --group"
        nb = max(self.B, 1)
        self.ids = list(shard_range(world * nb, rank, world))      # block-contiguous problem ids of this rank
        # creation order: the first member of every unit first, so that the streams that carry the launches get distinct priority
        # classes (handles take class = creation index mod 3, calipso_hip_create)
        G_ = self.G
        order = [k for k in range(nb) if k % G_ == 0] + [k for k in range(nb) if k % G_ != 0]
        made = {}
        self.structured = self.staged is not None and not (dense_structure or no_stage_parallel or no_stage_blocks or dense_buffers)
        for k in order:
            inst = make_instance(pkg, pr, self.ids[k], self.shape, device, self.staged, not dense_structure, self.structured)
            # the dense host copies of the problem data (~100 MB per C3 instance) are only needed until they are on the device
            made[k] = inst if k == 0 else (None, None, None, None, inst[4])
            if k != 0:
                inst[4].problem = None
                inst[4].methods = None
        self.prob0 = made[0][0]
        self.solvers = [made[k][4] for k in range(nb)]
        self.stage_parallel = None
        if self.structured:
            self.stage_parallel = dict(structured_handle=True)
        elif self.staged is not None and not dense_structure and not no_stage_parallel:
            # the Schur complement through the multifrontal sparse LDL^T over a nested dissection of its pattern (calipso_hip_set_stage_parallel):
            # on the handle that leads each unit (its storage covers the unit's G members)
            try:
                for k in range(0, nb, G_):
                    self.stage_parallel = self.solvers[k].set_stage_parallel(True, batch=G_)
            except pkg.CalipsoHipError as e:                      # a front exceeds one CU's LDS: the blocked factorisation stays
                self.stage_parallel = dict(refused=str(e))
        self.stage_blocks = None
        if self.structured:
            self.stage_blocks = dict(structured_handle=True)
        elif self.staged is not None and not dense_structure and not no_stage_blocks:
            # stage blocks (calipso_hip_set_stage_blocks): packed blocks of [gx; hx] / Lxx, block mat-vecs, Schur complement by segment pairs
            try:
                for sv in self.solvers:
                    self.stage_blocks = sv.set_stage_blocks(True)
            except pkg.CalipsoHipError as e:
                self.stage_blocks = dict(refused=str(e))
        self.single = self.solvers[0]
        self.units = ([pkg.Group(self.solvers[k:k + G_]) for k in range(0, self.B, G_)] if G_ > 1 else self.solvers[:self.B]) if self.B else []
        self.batch = BatchSolver(self.units, lanes=lanes) if self.units else None

    def sync(self):
        for s in self.solvers:
            s.synchronize()

    def batched_pass(self):
        out = self.batch.newton_step(advance=False)               # units run concurrently, one HIP stream each
        return [i for u in out for i in u] if self.G > 1 else out

    def batched_passes(self, P):
        out = self.batch.newton_steps(P, advance=False)           # lanes run free: independent problems need no pass-level synchronisation
        return [i for u in out for i in u] if self.G > 1 else out

    def kind(self, dense_structure):
        st = self.staged
        if not st:
            return "dense "
        return "stage-structured (%d stages, %s treatment) " % (st[0], "dense" if dense_structure else (
            "structured handle: stage blocks only, multifrontal LDL^T of S" if self.structured else
            ("stage blocks, " if self.stage_blocks and "z_blocks" in self.stage_blocks else "banded, ") +
            ("stage-parallel multifrontal LDL^T of S" if self.stage_parallel and "levels" in self.stage_parallel else "blocked LDL^T of S")))

    def describe(self):
        nx, ne, n_nn, n_soc, dim = self.shape
        nc = n_nn + n_soc * dim
        return "nx=%d ne=%d nc=%d (%d R+ + %d x SOC%d), n=%d condensed, N=%d unreduced" % (nx, ne, nc, n_nn, n_soc, dim, nx + ne + nc, nx + 2 * ne + 3 * nc)

    def close(self):
        if self.batch is not None:
            self.batch.close()
        for u in self.units:
            if hasattr(u, "close"):
                u.close()
        self.units, self.batch, self.single = [], None, None
        for s in self.solvers:
            s.__del__()
        self.solvers = []


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1, help="N > 1 outside torch.distributed.run: bench.py re-launches itself as N ranks (one per GPU)")
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=36, help="batched region: independent problem instances per GPU (B), arranged in groups of --group\n"
                    "members; --lanes groups are in flight at a time, see calipso.jl_amd/batch.py.  0 skips the batched region")
    ap.add_argument("--lanes", type=int, default=3, help="host threads / HIP streams driving the units (groups or single instances) concurrently")
    ap.add_argument("--group", type=int, default=12, help="instances per group: the members of a group are stepped in lockstep through the same\n"
                    "kernel launches (calipso_hip_group_*); --batch must be a multiple of it")
    ap.add_argument("--batched-passes", type=int, default=10, help="passes over all B instances in the batched timed region")
    ap.add_argument("--lockstep-passes", action="store_true", help="batched region: synchronise the lanes after every pass (round-1 behaviour) instead of\n"
                    "letting every lane run its passes back to back")
    ap.add_argument("--config", default="C3", choices=list(CONFIGS) + list(STAGED))
    ap.add_argument("--dense-structure", action="store_true", help="stage-structured configs: keep the dense treatment (no calipso_hip_analyze_structure)")
    ap.add_argument("--no-stage-parallel", action="store_true", help="stage-structured configs: keep the blocked banded LDL^T of S (no calipso_hip_set_stage_parallel)")
    ap.add_argument("--no-stage-blocks", action="store_true", help="stage-structured configs: keep the dense-layout mat-vecs and Schur kernel (no calipso_hip_set_stage_blocks)")
    ap.add_argument("--dense-buffers", action="store_true", help="stage-structured configs: handles made by calipso_hip_create + analyze_structure + set_stage_parallel + set_stage_blocks\n"
                    "(the dense Lxx / [gx; hx] / S buffers exist beside the blocks) instead of structured handles (calipso_hip_create_structured)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-c2-c5", action="store_true", help="skip config.c2 (pendulum solve!) and config.c5 (cart-pole sensitivities): BASELINE configs 2 and 5, rank 0 at N = 1 only, ~10 s")
    ap.add_argument("--cpu-baseline-full", action="store_true", help="measure B0(ii) (re-factorisation before every solve) instead of deriving it (~1 min more)")
    ap.add_argument("--cpu-samples", type=int, default=3)
    ap.add_argument("--no-single", action="store_true", help="profiling runs: skip the single-system region (every launch in the trace then carries a\n"
                    "whole group); the headline then falls back to the batched rate")
    ap.add_argument("--no-c4", action="store_true", help="skip the config.c4 block (BASELINE config 4: C4 dense and C4T stage-structured, 32 instances per GPU\n"
                    "in groups of 16); it only runs with --config C3")
    ap.add_argument("--c4-batch", type=int, default=32, help="config.c4: instances per GPU (BASELINE config 4: 256 problems over 8 GPUs)")
    ap.add_argument("--c4-group", type=int, default=16)
    ap.add_argument("--c4-all", type=int, default=256, help="config.c4: also time ALL the problems of BASELINE config 4 (256) on this one GPU for the stage-structured variant\n"
                    "(groups of up to 128 members, two in flight); 0 skips it")
    ap.add_argument("--c4-configs", default=None, help="the configurations of the config.c4 block (default: C4,C4T with --config C3, none otherwise)")
    ap.add_argument("--dist-backend", default="nccl", help="testing only: gloo lets two ranks share one GPU")
    ap.add_argument("--force-device", type=int, default=-1, help="testing only: every rank uses this device ordinal")
    ap.add_argument("--spawn-check", action="store_true", help="testing only (runs without a GPU): start the ranks, all-gather their ids and print\n"
                    "{\"spawn_check\": true, \"n_gpus\": N, \"ranks\": [...]} instead of benchmarking")
    args = ap.parse_args()
    spawn_ranks(args)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    dist = None
    if args.force_device >= 0:
        local_rank = args.force_device
    if args.spawn_check:
        # No GPU is touched: the ranks report in, then run the post-round exchange of config 4 on synthetic status rows — the block-contiguous shard of
        # world x --c4-batch problem ids, one row [ok, problem id, rank, group] per instance, gathered in global id order.  CALIPSO_BENCH_FAKE_COMM_FAIL_RANKS
        # ("1,3") makes the product communicator "fail" on those ranks: every rank must then agree to fall back to torch.distributed.
        ranks, extra = [rank], {}
        if world > 1:
            import torch.distributed as dist
            dist.init_process_group("gloo")
            ranks = [None] * world
            dist.all_gather_object(ranks, (rank, int(os.environ.get("LOCAL_RANK", "0"))))
            from __graft_entry__ import load_package
            load_package()
            from calipso_jl_amd.batch import shard_range
            failing = [int(v) for v in os.environ.get("CALIPSO_BENCH_FAKE_COMM_FAIL_RANKS", "").split(",") if v.strip()]

            class _StubComm:                              # stands for calipso_hip_comm_init (needs a GPU): succeeds or fails as the test asks
                def __init__(self, r, w, uid, device=0):
                    if r in failing:
                        raise RuntimeError("no communicator on rank %d (simulated)" % r)

                @staticmethod
                def unique_id():
                    return b"stub"

                def close(self):
                    pass

            class _StubPkg:
                Comm = _StubComm
            ids = list(shard_range(world * args.c4_batch, rank, world))
            ex = Exchange(_StubPkg, dist, "gloo", rank, world, 0, try_comm=bool(failing))
            rows, tot, _ = ex.gather([[1, pid, rank, k // max(1, args.c4_group)] for k, pid in enumerate(ids)], [float(len(ids))])
            extra = {"c4_shard": [ids[0], ids[-1], len(ids)] if ids else [], "gathered_ids": [int(v) for v in rows[:, 1]], "gathered_ranks": [int(v) for v in rows[:, 2]],
                     "gathered_groups": [int(v) for v in rows[:, 3]], "counter_total": float(tot[0]), "exchange_path": ex.path, "comm_left_open": ex.comm is not None}
            ex.close()
            dist.destroy_process_group()
        if rank == 0:
            print(json.dumps(dict({"spawn_check": True, "n_gpus": world, "gpus_argument": args.gpus, "ranks": ranks,
                                   "launched_by": "torch.distributed.run" if "TORCHELASTIC_RUN_ID" in os.environ else "direct"}, **extra)))
        return
    torch.cuda.set_device(local_rank)
    if world > 1:
        import torch.distributed as dist
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))    # RCCL over xGMI
        else:
            dist.init_process_group(args.dist_backend)

    from __graft_entry__ import load_package
    pkg = load_package()
    import problems as pr
    exchange = Exchange(pkg, dist, args.dist_backend, rank, world, local_rank)

    def barrier(wl):
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        wl.sync()

    def max_over_ranks(t):
        if dist is None:
            return t
        tt = torch.tensor([t], dtype=torch.float64, device="cuda" if args.dist_backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return float(tt.item())

    def run_batched(wl, P, lockstep=False):
        """timed region over P passes of all B instances of the rank + the post-round exchange; returns (elapsed, infos, k_schur ms samples)"""
        sch = []
        # the lanes' streams are probed once more right in front of the timed region (a few milliseconds; BatchSolver probed them at creation): what was created or
        # destroyed in between may have re-shuffled the hardware queues
        if wl.batch is not None and wl.batch.stream_report is not None:
            again = wl.batch.spread_streams()
            if again is not None:
                wl.batch.stream_report = dict(wl.batch.stream_report, recheck_collisions=again["collisions"], recheck_rebinds=again["rebinds"], left=again["left"])
        barrier(wl)
        t0 = time.perf_counter()
        if lockstep:
            for _ in range(P):
                infos = wl.batched_pass()
                sch.append(wl.solvers[0].phase_times()[7])
        else:
            infos = wl.batched_passes(P)
            sch.append(wl.solvers[0].phase_times()[7])
        barrier(wl)
        elapsed = max_over_ranks(time.perf_counter() - t0)
        assert all(i["status"] >= 0 for i in infos), "a Newton step of the batched region failed"
        status = [[int(i["status"] >= 0), P, i["refinement_rounds"], i["factorizations"]] for i in infos]
        all_status, counters, counts = exchange.gather(status, [float(len(infos) * P)])
        assert all_status.shape[0] == world * wl.B and int(round(float(counters[0]))) == world * wl.B * P
        assert all_status[:, 0].all(), "failed Newton steps must not count towards the reported rate"
        return elapsed, infos, sch

    def run_single(wl, K):
        """timed region: K Newton steps back to back, nothing else between them; returns (elapsed, infos)"""
        infos = []
        barrier(wl)
        t0 = time.perf_counter()
        if os.environ.get("CALIPSO_BENCH_STEP_CALLS", "0") == "1":      # one C call per step from this loop (what the bench did up to round 4: + the interpreter between the steps)
            for _ in range(K):
                infos.append(wl.single.newton_step(advance=False))
        else:                                                          # the K steps in ONE call of the C ABI (calipso_hip_newton_steps): the loop a caller of the library runs natively
            infos = wl.single.newton_steps(K, advance=False)
        barrier(wl)
        elapsed = max_over_ranks(time.perf_counter() - t0)
        assert all(i["status"] >= 0 for i in infos), "a Newton step of the timed region failed"
        return elapsed, infos

    def single_phases(wl, K):
        """a SECOND, untimed-for-the-headline region of K steps in which the handle's HIP-event phase times are read after every step (the queries
        synchronise on the events: they do not belong between the steps of the timed region)"""
        sch, ldl, chain, sd, tot, cw, mv = [], [], [], [], [], [], []
        kt_ = np.zeros(8)
        for _ in range(K):
            wl.single.newton_step(advance=False)
            pt_ = wl.single.phase_times()
            kt_ = wl.single.kernel_times()
            sch.append(pt_[7]); ldl.append(pt_[3]); sd.append(pt_[2]); tot.append(pt_[6]); cw.append(pt_[1])
            chain.append(kt_[0]); mv.append(kt_[4])
        # (read here: later the handle is the base of a group and its figures are the group's) [1] launches of the panel steps, [6] left-looking schedule, [7] its buffers
        return dict(schur=sch, ldl=ldl, chain=chain, sd=sd, total=tot, cone=cw, matvec=mv, kt=[float(v) for v in kt_])

    # =================================================================== headline workload ======================================
    wl = Workload(pkg, pr, args.config, rank, world, local_rank, args.batch, args.group, args.lanes, args.dense_structure, args.no_stage_parallel, args.no_stage_blocks, args.dense_buffers)
    B, G = wl.B, wl.G
    shape, staged = wl.shape, wl.staged
    # "opt.solve_block" (a tuning knob of the device factorisation, not an option of the reference): the widest diagonal block of L whose inverse
    # is assembled for the triangular solves.  ONE system wants 1024 (10 instead of 18 dependent launches per solve); a GROUP wants 512 (its solves are
    # bandwidth-bound and the extra assembly level costs it 0.5 ms per step).  Set on every member: the first member's setting governs a group's launches, and a
    # member stepped alone with the same setting gets the same bits.
    def solve_block(handles, value, wform=None):
        if staged is None:
            for h in handles:
                h.set_option("solve_block", value)
                if wform is not None:
                    h.set_option("solve_wform", wform)
    # ---- warm-up: W steps of the single system (captures its launch graphs) ---------------------------------------------------
    # "opt.solve_wform": the solves through the stacked [Tinv_b; W_b] blocks (two dependent launches per solve block instead of four): on for ONE system, off
    # for the members of a group (bandwidth-bound solves; the products would only cost them)
    solve_block([wl.single], int(os.environ.get("CALIPSO_BENCH_SOLVE_BLOCK", "1024")), int(os.environ.get("CALIPSO_BENCH_SOLVE_WFORM", "1")))
    for _ in range(args.warmup):
        if not args.no_single:
            wl.single.newton_step(advance=False)
    peak_measured = pkg.mfma_f64_peak(local_rank) if rank == 0 else None

    # ---- timed region 1 (headline): K sequential Newton steps of ONE system per GPU ------------------------------------------------
    K = args.steps
    single_elapsed, single_infos, ph = None, [], None
    if not args.no_single:
        single_elapsed, single_infos = run_single(wl, K)
        ph = single_phases(wl, max(3, min(K, 10)))

    # ---- warm-up of the batched pass; unit 0 alone (one group of G instances): its launches have the device to themselves => clean per-launch figures --
    alone, alone_chain, unit_rate = [], [], None
    if wl.batch is not None:
        solve_block(wl.solvers if G > 1 else [], int(os.environ.get("CALIPSO_BENCH_GROUP_SOLVE_BLOCK", "512")), int(os.environ.get("CALIPSO_BENCH_GROUP_SOLVE_WFORM", "0")))
        for _ in range(args.warmup):
            wl.batched_pass()
        barrier(wl)
        n_alone = max(3, min(10, K))
        ts = time.perf_counter()
        for _ in range(n_alone):
            wl.units[0].newton_step(advance=False)
            alone.append(wl.units[0].phase_times())
            alone_chain.append(wl.solvers[0].kernel_times()[0])
        wl.units[0].synchronize()
        unit_rate = G * n_alone / (time.perf_counter() - ts)

    # ---- timed region 2 (batched): P passes over all B instances of the rank ------------------------------------------------------
    P = max(1, args.batched_passes)
    batched_elapsed, infos, sch_conc = None, None, []
    if wl.batch is not None:
        batched_elapsed, infos, sch_conc = run_batched(wl, P, args.lockstep_passes)

    info = single_infos[-1] if single_infos else infos[0]
    nx, ne, n_nn, n_soc, dim = shape
    nc = n_nn + n_soc * dim
    m = ne + nc
    NP = wl.single.padded_nx()
    if single_elapsed is not None:
        value, elapsed, steps_timed = world * K / single_elapsed, single_elapsed, K
    else:                                                     # --no-single (profiling): the batched rate stands in
        value, elapsed, steps_timed = world * B * P / batched_elapsed, batched_elapsed, P

    # ---- roofline of the DOMINANT kernel = the kernel with the largest share of the headline step -----------------------------------
    # Two candidates carry matrix-core work (everything else is a few microseconds per launch): the panel-step kernel of the LDL^T of S
    # (k_ldl_step: one launch per 64 pivots; its first workgroup carries the sequential pivot chain) and the Schur-complement kernel k_schur.
    # Both are timed live with HIP events on the handle's stream (calipso_hip_kernel_times / phase_times) over the launches of the TIMED
    # region; whichever takes the larger share of the step is `roofline`, the other goes to roofline.secondary.  Algorithmic flops:
    #   k_ldl_step  nx^3 / 3 per factorisation (a dense LDL^T of the nx x nx Schur complement), spread over its NP / 64 - 1 launches
    #   k_schur     multiply-adds of the lower triangle of S incl. diagonal: per constraint row with w non-zero columns w (w + 1); dense: nx (nx + 1)
    prob0 = wl.prob0
    wrow = np.concatenate([np.count_nonzero(prob0.A, axis=1), np.count_nonzero(prob0.G, axis=1)]).astype(np.float64)
    flops_schur = float(np.sum(wrow * (wrow + 1.0)))
    flops_ldl = nx ** 3 / 3.0
    n_ldl_launch = max(1, NP // 64 - 1)
    pmc = {}
    for name in ("r06_pmc_summary.json", "r05_pmc_summary.json", "r04_pmc_summary.json", "r03_pmc_summary.json", "r02_pmc_summary.json"):
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", name)))
            pmc["_file"] = "profiles/" + name
            break
        except Exception:
            pmc = {}

    def pmc_traffic(section, kernel_prefix, scale=1.0):
        """HBM bytes per launch (PMC passes, profiles/) of the kernel VARIANT that carries the section's work: among the counter rows whose name starts with
        the prefix, the one with the most bytes over its sampled launches (a group's factorisation runs k_ldl_step<2> with a few <1> launches beside it, one system <0>)"""
        if args.config != "C3":
            return None
        best, best_total = None, -1.0
        for kname, e in pmc.get(section, {}).items():
            if kname.startswith(kernel_prefix) and "hbm_bytes_per_launch" in e:
                total = e["hbm_bytes_per_launch"] * max(1, e.get("launches_sampled", 1))
                if total > best_total:
                    best, best_total = e, total
        return best["hbm_bytes_per_launch"] * scale if best else None

    def entry(kernel, flops_per_factor, launches, ms_total, inst, section, prefix, step_ms):
        ms_launch = ms_total / launches
        e = {"kernel": kernel, "bound": "mfma", "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
             "achieved": inst * flops_per_factor / (ms_total * 1e-3) * 1e-12, "flops_per_launch": inst * flops_per_factor / launches,
             "launches_per_step": launches, "instances_per_launch": inst, "avg_launch_ms": ms_launch, "ms_per_step": ms_total,
             "share_of_step": ms_total / step_ms if step_ms else None, "traffic": pmc_traffic(section, prefix)}
        e["frac"] = e["achieved"] / FP64_MFMA_PEAK_TFLOPS
        return e
    K_LDL = "k_ldl_step (LDL^T of the nx x nx Schur complement S, one launch per 64 pivots: trailing update on v_mfma_f64_16x16x4_f64, its first workgroup factors the next diagonal block)"
    K_LFAC = ("k_lfac (one dense system alone: the left-looking schedule of csrc/lfac.hip — every launch = the pivot chain's workgroup for one 64-pivot panel + 255 workers on "
              "slices of S = Lxx + eps*I + omega*gx'gx + hx'(Omega hx) and on the deferred trailing updates, v_mfma_f64_16x16x4_f64; flops = Schur complement + nx^3/3)")
    K_SCH = "k_schur (S = Lxx + eps*I + omega*gx'gx + hx'(Omega hx), v_mfma_f64_16x16x4_f64)"
    cands = []
    if ph is not None:
        step_ms = float(np.mean(ph["total"]))
        kt_single = ph["kt"]
        if kt_single[6] > 0:      # the Schur complement's products are slices of the panel launches: ONE kernel carries both (calipso_hip_kernel_times [6])
            cands = [entry(K_LFAC, flops_schur + flops_ldl, max(1, int(kt_single[1])), float(np.mean(ph["chain"])), 1, "single", "calipso::k_lfac", step_ms)]
            cands[0]["flops_schur"] = flops_schur; cands[0]["flops_ldl"] = flops_ldl
            cands[0]["schedule_buffers_bytes"] = kt_single[7]
        else:
            cands = [entry(K_LDL, flops_ldl, n_ldl_launch, float(np.mean(ph["chain"])), 1, "single", "calipso::k_ldl_step", step_ms),
                     entry(K_SCH, flops_schur, 1, float(np.mean(ph["schur"])), 1, "single", "calipso::k_schur", step_ms)]
    grp = None
    if alone:
        al = np.mean(np.asarray(alone), axis=0)
        grp = [entry(K_LDL, flops_ldl, n_ldl_launch, float(np.mean(alone_chain)), G, "group", "calipso::k_ldl_step", float(al[6])),
               entry(K_SCH, flops_schur, 1, float(al[7]), G, "group", "calipso::k_schur", float(al[6]))]
        grp[1]["avg_launch_ms_with_all_units_in_flight"] = float(np.mean(sch_conc)) if sch_conc else None
        if not cands:
            cands = grp
    if staged is not None:                                    # stage-structured: the LDL^T is the multifrontal path, k_ldl_step does not run
        cands = [c for c in cands if c["ms_per_step"] > 0] or cands
        for c in cands + (grp or []):
            if c["kernel"] == K_LDL:                          # (the dense nx^3 / 3 does not price a factorisation over the stage tree: time only)
                c.update(kernel="multifrontal LDL^T of S over the nested-dissection tree of the stages (sparse.hip), timed as a whole", achieved=None, frac=None,
                         flops_per_launch=None, launches_per_step=None, avg_launch_ms=None)
    cands.sort(key=lambda c: -c["ms_per_step"])
    roof = dict(cands[0])
    roof["secondary"] = cands[1:]
    if ph is not None and np.mean(ph["matvec"]) > 0:
        # the HBM-bound side of the step: the mat-vec kernel of a refinement residual ([gx; hx]' times two vectors + Lxx times one, one pass over both), ONE
        # launch timed live per step with HIP events on the handle's stream (calipso_hip_kernel_times [4]); algorithmic bytes = the two blocks read once
        mv_ms = float(np.mean(ph["matvec"]))
        mv_bytes = 8.0 * (m * nx + nx * nx)
        roof["secondary"].append({"kernel": "k_gemv_t2_and_n (refinement residual: [gx; hx]'(v_yz, Omega b_m) and Lxx v_x in one pass)", "bound": "hbm", "peak": 8000.0, "unit": "GB/s",
                                  "achieved": mv_bytes / (mv_ms * 1e-3) * 1e-9, "frac": mv_bytes / (mv_ms * 1e-3) * 1e-9 / 8000.0, "bytes_per_launch": mv_bytes, "avg_launch_ms": mv_ms,
                                  "launches_per_step": 1 + int(single_infos[-1]["refinement_rounds"]), "traffic": pmc_traffic("single", "calipso::k_gemv_t2_and_n")})
    roof["dominance"] = ("the kernel with the largest share of the headline step among all kernels (profiles/r05_kernel_stats_single.csv lists every kernel); "
                         "HIP-event durations of this run")
    roof["peak_measured"] = peak_measured
    roof["peak_note"] = ("peak = datasheet fp64 matrix rate (not tabulated in MI355X_MICROARCH.md); peak_measured = calipso_hip_mfma_f64_peak in this run; "
                         "traffic = (2 FETCH_SIZE + WRITE_SIZE) KiB of %s (separate rocprofv3 --pmc passes of the same command), null when no counter file covers the kernel" % pmc.get("_file", "profiles/"))
    if grp:
        roof["group_launch"] = {g["kernel"].split(" ")[0]: g for g in grp}

    # per-phase rooflines: SURVEY.md 8(d)'s algorithmic figures AND the bytes / flops the constraint-first path really executes
    n_cond = nx + m
    n_r = int(info["refinement_rounds"])

    def phases(al, inst):
        t_factor = float(al[1] + al[7] + al[3])                       # cone pivots + Schur complement + LDL^T of S
        t_solve = float(al[2]) - t_factor                              # condensed solves + recovery + refinement residuals
        f_survey = inst * n_cond ** 3 / 3.0                            # dense n^3/3 of 8(d)
        f_exec = inst * (flops_schur + flops_ldl)                      # what the constraint-first order executes
        b_survey = inst * (1 + n_r) * (2 * 8 * n_cond * (n_cond + 1) / 2 + 8.0 * (nx * nx + ne * nx + nc * nx))
        # executed: every solve reads the factor of S twice (L forward, L' backward: NP^2 / 2 doubles each); [gx; hx] is passed over
        # 2 (first solve) + n_r (correction solves) + n_r + 1 (refinement residuals) times, Lxx n_r + 1 times
        b_exec = inst * 8.0 * ((1 + n_r) * NP * NP + (2 * n_r + 3) * m * nx + (n_r + 1) * nx * nx)
        return {"instances": inst, "whole_step_ms": float(al[6]),
                "factor": {"ms": t_factor, "schur_ms": float(al[7]), "ldl_ms": float(al[3]), "bound": "mfma", "flops_survey_n3_over_3": f_survey,
                           "flops_executed": f_exec, "achieved_TFLOPs_survey": f_survey / t_factor * 1e-9,
                           "achieved_TFLOPs_executed": f_exec / t_factor * 1e-9, "frac_executed": f_exec / t_factor * 1e-9 / FP64_MFMA_PEAK_TFLOPS},
                "solve_and_refine": {"ms": t_solve, "bound": "hbm", "solves": 1 + n_r, "bytes_executed": b_exec, "achieved_GBs_executed": b_exec / t_solve * 1e-6,
                                     "frac_executed": b_exec / t_solve * 1e-6 / 8000.0, "bytes_survey": b_survey,
                                     "note": "priced with the bytes the executed path moves (factor of the NP x NP Schur complement, [gx; hx], Lxx); bytes_survey = the "
                                             "n = nx + ne + nc triangular-solve figure of SURVEY 8(d), kept for reference only"}}
    cfg_phases = {}
    if ph is not None:
        al1 = np.zeros(9); al1[7] = np.mean(ph["schur"]); al1[3] = np.mean(ph["ldl"]); al1[2] = np.mean(ph["sd"]); al1[6] = np.mean(ph["total"])
        al1[1] = np.mean(ph["cone"])      # (read per step inside the timed region: afterwards the handle has been the base of a group and holds the group's figure)
        cfg_phases["single_system"] = phases(al1, 1)
        cfg_phases["single_system"]["launches_per_step"] = wl.single.kernel_times()[1]
    if alone:
        cfg_phases["one_group_alone"] = phases(np.mean(np.asarray(alone), axis=0), G)

    batched = None
    if batched_elapsed is not None:
        brate = world * B * P / batched_elapsed
        batched = {"newton_steps_per_s": brate, "problems_per_s_of_10_steps": brate / 10.0, "instances_per_gpu": B, "instances_per_group": G,
                   "groups_in_flight": wl.batch.lanes, "lane_streams": wl.batch.stream_report, "passes": P, "lanes_synchronised_per_pass": bool(args.lockstep_passes), "ms_per_pass": 1e3 * batched_elapsed / P, "one_group_alone_steps_per_s": unit_rate,
                   "scaling": "weak (instances sharded block-contiguously over ranks, no data-path collective)", "post_round_exchange": exchange.path}
    workload = ("%s synthetic " + wl.kind(args.dense_structure) + "conic QP: %s; %s; 1 LDL^T factorisation, %d refinement round(s) per step") % (
        args.config, wl.describe(), "ONE system per GPU stepped sequentially (replicas at N > 1)" if single_elapsed is not None
        else "%d independent instances per GPU (batched rate, --no-single)" % B, info["refinement_rounds"])
    refinement_rounds, factorizations = info["refinement_rounds"], info["factorizations"]
    wl.close()
    del wl

    def structured_roofline(w, n_r, seconds_per_instance_step, what):
        """Executed flops and bytes of ONE Newton step of a structured handle (calipso_hip_structure_work: Schur complement by segment pairs + the multifrontal LDL^T
        of S; bytes: the packed blocks of [gx; hx] and Lxx once per pass over them — Schur complement, 2 n_r + 3 mat-vec passes over [gx; hx], n_r + 1 over Lxx — and
        the fronts of L once for the factorisation (written) and twice per solve) against both ceilings.  Such a step is neither: its kernels are small and the step is
        bound by launch latency, which is what the fractions say."""
        sw = w.single.structure_work()
        flops = sw["schur_flops"] + sw["factor_flops"]
        passes = 1 + (2 * n_r + 3) + (n_r + 1)
        byts = 8.0 * (sw["packed_doubles"] / 2.0 * passes + sw["factor_nnz"] * (1 + 2 * (1 + n_r)))      # (packed_doubles counts both orientations; a pass reads one)
        tf = flops / seconds_per_instance_step * 1e-12
        gb = byts / seconds_per_instance_step * 1e-9
        out = {"what": what, "flops_executed_per_step": flops, "schur_flops": sw["schur_flops"], "factor_flops": sw["factor_flops"], "bytes_executed_per_step": byts,
               "us_per_instance_step": 1e6 * seconds_per_instance_step, "achieved_TFLOPs": tf, "frac_mfma": tf / FP64_MFMA_PEAK_TFLOPS, "achieved_GBs": gb, "frac_hbm": gb / 8000.0,
               "bound": "hbm" if gb / 8000.0 >= tf / FP64_MFMA_PEAK_TFLOPS else "mfma",
               "limited_by": "launch latency: ~110 dependent launches of a few microseconds of work per step (frac_mfma and frac_hbm are both far from 1)"}
        # the multifrontal factorisation of the LAST group factorisation that went through this handle's tree (events around its launches: k_gather_dense, one k_mf_factor
        # per level, k_count_signs; the other lane's kernels run beside it)
        ms = float(w.single.kernel_times()[0])
        if ms > 0.0 and sw["factor_flops"] > 0:
            tfk = sw["factor_flops"] * w.G / ms * 1e-9
            out["kernels"] = {"k_mf_factor": {"bound": "mfma", "ms_per_group_factorisation": ms, "members": w.G, "flops_executed": sw["factor_flops"] * w.G,
                                              "achieved": tfk, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tfk / FP64_MFMA_PEAK_TFLOPS,
                                              "note": "one launch per level of the stage tree (6-7), one workgroup per front and member; the levels near the root hold 1-5 fronts per member"}}
        return out

    # =================================================================== config.c4: BASELINE config 4 ===========================
    # 256 quadruped-gait-sized problems sharded over 8 GPUs = 32 instances per GPU, here in groups of 16, two groups in flight: C4 (the dense
    # treatment of that size) and C4T (the same size with the stage structure of a 41-stage trajectory problem: band-limited Schur complement,
    # stage-parallel multifrontal LDL^T).  Same timed-region rules as above; no CPU baseline.
    c4 = None
    c4_configs = args.c4_configs if args.c4_configs is not None else ("C4,C4T" if args.config == "C3" else "")
    if c4_configs and not args.no_c4 and args.c4_batch > 0:
        c4 = {"instances_per_gpu": args.c4_batch, "instances_per_group": args.c4_group, "problems_total": world * args.c4_batch,
              "problem": "10 Newton steps of one instance (SURVEY 8(d))"}
        # (CALIPSO_BENCH_C4_SHIFT=k: k small handles created first — shifts which hardware queues the lanes' streams get: the experiment behind BatchSolver.spread_streams,
        # profiles/r06_ab_closing.txt)
        _shift = [make_instance(pkg, pr, 900 + q, (16, 4, 4, 0, 3), local_rank)[4] for q in range(int(os.environ.get("CALIPSO_BENCH_C4_SHIFT", "0")))]
        for cname in [c for c in c4_configs.split(",") if c]:
            w4 = Workload(pkg, pr, cname, rank, world, local_rank, args.c4_batch, args.c4_group, 2)
            # (CALIPSO_BENCH_PROBE_PRINT=1: the raw numbers of calipso_hip_streams_concurrent for the first two lanes' leaders on stderr, four times — profiles/r06_ab_closing.txt)
            if os.environ.get("CALIPSO_BENCH_PROBE_PRINT") and w4.batch is not None and len(w4.units) > 1:
                la, lb = w4.batch._leader(w4.units[0]), w4.batch._leader(w4.units[1])
                for _ in range(4):
                    sys.stderr.write("PROBE %s %s\n" % (cname, ["%.1f" % v if not isinstance(v, bool) else str(v) for v in la.streams_concurrent(lb)]))
            for _ in range(max(1, min(args.warmup, 2))):
                w4.single.newton_step(advance=False)
            K4 = max(3, min(K, 10))
            e1, i1 = run_single(w4, K4)
            if w4.staged is None and w4.G > 1:
                for h4 in w4.solvers:
                    h4.set_option("solve_block", 512)
                    h4.set_option("solve_wform", 0)
            for _ in range(max(1, min(args.warmup, 2))):
                w4.batched_pass()
            P4 = max(1, min(P, 10))
            if w4.staged is not None and w4.structured:
                P4 *= 8             # (a pass of 32 structured instances is ~1.2 ms: ten passes are a 12 ms region in which ONE host hiccup moves the rate by 25 % — 19 to 26 k steps/s in six runs)
            e2, i2, _ = run_batched(w4, P4)
            r2 = world * w4.B * P4 / e2
            c4[cname] = {"workload": cname + " " + w4.kind(False) + w4.describe(), "single_system_steps_per_s": world * K4 / e1, "single_ms_per_step": 1e3 * e1 / K4,
                         "batched_newton_steps_per_s": r2, "batched_problems_per_s_of_10_steps": r2 / 10.0, "ms_per_pass": 1e3 * e2 / P4, "passes": P4,
                         "refinement_rounds": int(i2[0]["refinement_rounds"]), "stage_parallel": w4.stage_parallel, "stage_blocks": w4.stage_blocks,
                         "device_bytes_per_instance": w4.single.device_bytes(), "lane_streams": w4.batch.stream_report if w4.batch is not None else None}
            if w4.staged is not None and w4.structured:
                c4[cname]["roofline"] = structured_roofline(w4, int(i2[0]["refinement_rounds"]), 1e-3 * c4[cname]["ms_per_pass"] / w4.B, "%d instances in groups of %d" % (w4.B, w4.G))
            if rank == 0 and world == 1 and not args.no_cpu_baseline:
                try:
                    cb4 = cpu_step_baseline(w4.shape, w4.staged, samples=1)
                    cb4["gpu_single_over_cpu"] = c4[cname]["single_system_steps_per_s"] / cb4["value"]
                    cb4["gpu_batched_over_cpu"] = r2 / cb4["value"]
                    c4[cname]["cpu_baseline"] = cb4
                except Exception as e:
                    c4[cname]["cpu_baseline"] = {"error": repr(e)}
            w4.close()
            del w4
            if cname == "C4T" and args.c4_all > 0:
                # the WHOLE batch of config 4 on this GPU (what one GPU does with all 256 problems): groups of up to 128 members, two in flight
                ga = min(128, args.c4_all)
                wa = Workload(pkg, pr, cname, rank, world, local_rank, args.c4_all, ga, 2)
                for _ in range(2):
                    wa.batched_pass()
                ea, ia, _ = run_batched(wa, P4)
                ra = world * wa.B * P4 / ea
                c4[cname]["all_%d_on_one_gpu" % args.c4_all] = {"instances_per_gpu": wa.B, "instances_per_group": ga, "groups_in_flight": 2, "batched_newton_steps_per_s": ra,
                                                               "batched_problems_per_s_of_10_steps": ra / 10.0, "ms_per_pass": 1e3 * ea / P4,
                                                               "roofline": structured_roofline(wa, int(ia[0]["refinement_rounds"]), ea / P4 / wa.B, "%d instances in groups of %d" % (wa.B, ga))}
                wa.close()
                del wa
                # config 4 is 256 problems on 8 GPUs: what N GPUs would make of it from the per-GPU rates measured HERE at 256 / N instances (no collective on the data path:
                # a projection from one GPU's measurements, the driver's SCALE run is the measurement)
                if args.c4_all == 256 and world == 1:
                    per_gpu = {256: ra / world, args.c4_batch: r2 / world}
                    for bb in (128, 64):
                        if bb in per_gpu:
                            continue
                        wb = Workload(pkg, pr, cname, rank, world, local_rank, bb, min(128, bb), 2)
                        for _ in range(2):
                            wb.batched_pass()
                        eb, _, _ = run_batched(wb, P4)
                        per_gpu[bb] = wb.B * P4 / eb
                        wb.close()
                        del wb
                    if all(k in per_gpu for k in (256, 128, 64, 32)):
                        c4[cname]["strong_scaling_projection"] = {
                            "what": "256 problems over N GPUs = 256 / N instances per GPU, each rate measured on ONE GPU; speedup = N x rate(256 / N) / rate(256)",
                            "steps_per_s_per_gpu": {str(k): per_gpu[k] for k in (256, 128, 64, 32)},
                            "speedup": {str(n): n * per_gpu[256 // n] / per_gpu[256] for n in (1, 2, 4, 8)}}

    c2 = c5 = None
    if rank == 0 and world == 1 and not args.no_c2_c5 and args.config == "C3":
        try:
            c2 = config_c2(pkg, pr, local_rank if args.force_device < 0 else args.force_device, cpu=not args.no_cpu_baseline)
            c5 = config_c5(pkg, pr, local_rank if args.force_device < 0 else args.force_device, cpu=not args.no_cpu_baseline)
        except Exception as e:      # (never lose the headline line to a side figure)
            c2 = c2 or {"error": repr(e)}
            c5 = c5 or {"error": repr(e)}
    c3_solve = small_newton = None
    if rank == 0 and world == 1 and not args.no_c2_c5 and args.config == "C3":
        try:
            c3_solve = config_c3_solve(pkg, pr, local_rank if args.force_device < 0 else args.force_device, shape)
        except Exception as e:
            c3_solve = {"error": repr(e)}
        try:
            small_newton = config_small_newton(pkg, pr, local_rank if args.force_device < 0 else args.force_device, cpu=not args.no_cpu_baseline)
        except Exception as e:
            small_newton = {"error": repr(e)}
    out = {
        "metric": "Newton steps/sec (n~5k KKT)", "value": value, "unit": "Newton steps/s", "n_gpus": world, "steps": steps_timed,
        "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / steps_timed, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": workload,
                   "parallelism": "one system per GPU: replicas only; batched: independent problems per GPU (no data-path collective)",
                   "refinement_rounds": refinement_rounds, "factorizations_per_step": factorizations,
                   "cpu_baseline_rows": "cpu_baseline.kind = 'port': the repo's single-thread C++ restatement of the reference's CPU path (oracle/), NOT the Julia reference (no julia on the box: "
                                        "B2 says so); value = B0(i), one factorisation per step (favourable to the reference); B0_ii (the reference's re-factorisation before every solve) is "
                                        + ("MEASURED in this run (--cpu-baseline-full)" if args.cpu_baseline_full else "NOT in this line (value: null; --cpu-baseline-full measures it, ~100 s more)")
                                        + "; B1 = LAPACK on all cores, not the reference",
                   "batched": batched, "c4": c4, "c2": c2, "c5": c5, "c3_solve": c3_solve, "small_newton": small_newton, "roofline_phases": cfg_phases,
                   "rccl_ranks": exchange.ranks_reported, "exchange_path": exchange.path},
        "roofline": roof,
    }
    exchange.close()
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(shape, args.config, staged, samples=args.cpu_samples, full=args.cpu_baseline_full)
    else:
        out["cpu_baseline"] = None
    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
