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


def make_instance(pkg, pr, pid, shape, device, staged=None, analyze=True):
    nx, ne, n_nn, n_soc, dim = shape
    if staged is not None:
        prob, pt, lam = pr.staged_conic_qp(pkg.splitmix_uniform, pid, *staged)
    else:
        prob, pt, lam = pr.synthetic_conic_qp(pkg.splitmix_uniform, pid, nx, ne, n_nn, n_soc, dim)
    s = pkg.Solver(prob, prob.nx, 0, prob.ne, prob.nc, nonnegative_indices=prob.nonnegative_indices,
                   second_order_indices=prob.second_order_indices, device=device)
    w = np.concatenate([pt[k] for k in "xrsyzt"])
    s.set("solution", w)
    s.set("dual", lam)
    for name, v in (("central_path", 0.17), ("penalty", 52.0), ("fraction_to_boundary", 0.99)):
        s.set(name, [v])
    s.qp_attach(prob.P, prob.q, prob.A, prob.b, prob.G, prob.h, 0.5)
    if staged is not None and analyze:
        s.analyze_structure()
    fl = pkg.FLAGS
    s.qp_evaluate(fl["objective"] | fl["equality_constraint"] | fl["cone_constraint"], 0)
    s.cone(product=True, target=True)
    s.synchronize()
    return prob, pt, lam, w, s


def staged_shape(st):
    T, nv, nd, nn, nsoc, dim = st
    return (T * nv, (T - 1) * nd, T * nn, T * nsoc, dim)


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
    b0ii = dict(value=1.0 / (t_i + extra * t_fact), unit="Newton steps/s", cores=1, factorizations_per_step=1 + extra, measured=False,
                how="B0(i) median + %d x one separately timed factorisation (%.2f s)" % (extra, t_fact))
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
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
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-full", action="store_true", help="measure B0(ii) (re-factorisation before every solve) instead of deriving it (~1 min more)")
    ap.add_argument("--cpu-samples", type=int, default=3)
    ap.add_argument("--no-single", action="store_true", help="profiling runs: skip the single-system region (every launch in the trace then carries a\n"
                    "whole group); the headline then falls back to the batched rate")
    ap.add_argument("--dist-backend", default="nccl", help="testing only: gloo lets two ranks share one GPU")
    ap.add_argument("--force-device", type=int, default=-1, help="testing only: every rank uses this device ordinal")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    dist = None
    if args.force_device >= 0:
        local_rank = args.force_device
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
    staged = STAGED.get(args.config)
    shape = staged_shape(staged) if staged else CONFIGS[args.config]
    B = max(0, args.batch)
    from calipso_jl_amd.batch import BatchSolver, gather_results, shard_range
    G = max(1, args.group)
    assert B % G == 0, "--batch must be a multiple of --group"
    nb = max(B, 1)
    ids = list(shard_range(world * nb, rank, world))          # block-contiguous problem ids of this rank
    # creation order: the first member of every unit first, so that the streams that carry the launches get distinct priority
    # classes (handles take class = creation index mod 3, calipso_hip_create)
    order = [k for k in range(nb) if k % G == 0] + [k for k in range(nb) if k % G != 0]
    made = {}
    for k in order:
        inst = make_instance(pkg, pr, ids[k], shape, local_rank, staged, not args.dense_structure)
        # the dense host copies of the problem data (~100 MB per C3 instance) are only needed until they are on the device
        made[k] = inst if k == 0 else (None, None, None, None, inst[4])
        if k != 0:
            inst[4].problem = None
            inst[4].methods = None
    solvers = [made[k][4] for k in range(nb)]
    stage_parallel = None
    if staged is not None and not args.dense_structure and not args.no_stage_parallel:
        # the Schur complement through the multifrontal sparse LDL^T over a nested dissection of its pattern (calipso_hip_set_stage_parallel):
        # on the handle that leads each unit (its storage covers the unit's G members)
        try:
            for k in range(0, nb, max(G, 1)):
                stage_parallel = solvers[k].set_stage_parallel(True, batch=max(G, 1))
        except pkg.CalipsoHipError as e:                      # a front exceeds one CU's LDS: the blocked factorisation stays
            stage_parallel = dict(refused=str(e))
    single = solvers[0]                                       # the headline system of this rank (problem id = first of its shard)
    units = ([pkg.Group(solvers[k:k + G]) for k in range(0, B, G)] if G > 1 else solvers[:B]) if B else []
    batch = BatchSolver(units, lanes=args.lanes) if units else None

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        for s in solvers:
            s.synchronize()

    def max_over_ranks(t):
        if dist is None:
            return t
        tt = torch.tensor([t], dtype=torch.float64, device="cuda" if args.dist_backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return float(tt.item())

    def batched_pass():
        out = batch.newton_step(advance=False)               # units run concurrently, one HIP stream each
        return [i for u in out for i in u] if G > 1 else out

    # ---- warm-up: W steps of the single system (captures its launch graphs) and of the batched pass ----------------------------
    for _ in range(args.warmup):
        if not args.no_single:
            single.newton_step(advance=False)
        if batch is not None:
            batched_pass()
    peak_measured = pkg.mfma_f64_peak(local_rank) if rank == 0 else None

    # ---- timed region 1 (headline): K sequential Newton steps of ONE system per GPU ------------------------------------------------
    K = args.steps
    single_elapsed, single_infos, sch_single, ldl_single, sd_single, tot_single = None, [], [], [], [], []
    if not args.no_single:
        barrier()
        t0 = time.perf_counter()
        for _ in range(K):
            single_infos.append(single.newton_step(advance=False))
            pt_ = single.phase_times()
            sch_single.append(pt_[7]); ldl_single.append(pt_[3]); sd_single.append(pt_[2]); tot_single.append(pt_[6])
        barrier()
        single_elapsed = max_over_ranks(time.perf_counter() - t0)
        assert all(i["status"] >= 0 for i in single_infos), "a Newton step of the timed region failed"

    # ---- unit 0 alone (one group of G instances): its launches have the device to themselves => clean per-launch figures ------------
    alone, unit_rate = [], None
    if batch is not None:
        barrier()
        n_alone = max(3, min(10, K))
        ts = time.perf_counter()
        for _ in range(n_alone):
            units[0].newton_step(advance=False)
            alone.append(units[0].phase_times())
        units[0].synchronize()
        unit_rate = G * n_alone / (time.perf_counter() - ts)

    # ---- timed region 2 (batched): P passes over all B instances of the rank ------------------------------------------------------
    P = max(1, args.batched_passes)
    batched_elapsed, infos, sch_conc = None, None, []
    if batch is not None:
        barrier()
        t0 = time.perf_counter()
        if args.lockstep_passes:
            for _ in range(P):
                infos = batched_pass()
                sch_conc.append(solvers[0].phase_times()[7])
        else:                                                 # lanes run free: independent problems need no pass-level synchronisation
            out_ = batch.newton_steps(P, advance=False)
            infos = [i for u in out_ for i in u] if G > 1 else out_
            sch_conc.append(solvers[0].phase_times()[7])
        barrier()
        batched_elapsed = max_over_ranks(time.perf_counter() - t0)
        assert all(i["status"] >= 0 for i in infos), "a Newton step of the batched region failed"
        # post-round exchange (outside the data path): per-problem status rows all-gathered, step counters all-reduced
        status = [[int(i["status"] >= 0), P, i["refinement_rounds"], i["factorizations"]] for i in infos]
        all_status, counters = gather_results(status, [float(len(infos) * P)])
        assert all_status.shape[0] == world * B and int(counters[0]) == world * B * P
        assert all_status[:, 0].all(), "failed Newton steps must not count towards the reported rate"

    info = single_infos[-1] if single_infos else infos[0]
    nx, ne, n_nn, n_soc, dim = shape
    nc = n_nn + n_soc * dim
    m = ne + nc
    if single_elapsed is not None:
        value, elapsed, steps_timed = world * K / single_elapsed, single_elapsed, K
    else:                                                     # --no-single (profiling): the batched rate stands in
        value, elapsed, steps_timed = world * B * P / batched_elapsed, batched_elapsed, P
    # dominant kernel (by flops): the Schur-complement update S = Lxx + ep*I + Z' Omega Z on the fp64 matrix cores (k_schur).
    # algorithmic flops per instance = multiply-adds of the lower triangle incl. diagonal: per constraint row with w non-zero columns
    # w (w + 1); dense rows: nx (nx + 1).  Launch duration: HIP events on the stream the kernel runs on (phase_times[7]), averaged over
    # the launches of the TIMED region (one instance per launch there); the group launch (G instances) is reported beside it.
    prob0 = made[0][0]
    wrow = np.concatenate([np.count_nonzero(prob0.A, axis=1), np.count_nonzero(prob0.G, axis=1)]).astype(np.float64)
    flops1 = float(np.sum(wrow * (wrow + 1.0)))
    roof = {"kernel": "k_schur (S = Lxx + eps*I + omega*gx'gx + hx'(Omega hx), v_mfma_f64_16x16x4_f64)", "bound": "mfma",
            "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "peak_measured": peak_measured,
            "peak_note": "peak = datasheet fp64 matrix rate (not tabulated in MI355X_MICROARCH.md); peak_measured = calipso_hip_mfma_f64_peak in this run"}
    pmc = {}
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", "r02_pmc_summary.json")))
    except Exception:
        pmc = {}
    if sch_single:
        ms1 = float(np.mean(sch_single))
        roof.update(achieved=flops1 / (ms1 * 1e-3) * 1e-12, flops_per_launch=flops1, instances_per_launch=1, avg_launch_ms=ms1)
        e = pmc.get("single", {}).get("calipso::k_schur") if args.config == "C3" else None
        roof["traffic"] = e["hbm_bytes_per_launch"] if e else None
    if alone:
        msg = float(np.mean([a[7] for a in alone]))
        grp = dict(achieved=G * flops1 / (msg * 1e-3) * 1e-12, flops_per_launch=G * flops1, instances_per_launch=G, avg_launch_ms=msg,
                   avg_launch_ms_with_all_units_in_flight=float(np.mean(sch_conc)) if sch_conc else None)
        e = pmc.get("group", {}).get("calipso::k_schur") if args.config == "C3" else None
        grp["traffic"] = e["hbm_bytes_per_launch"] * G / float(e.get("instances_per_launch", G)) if e else None
        grp["frac"] = grp["achieved"] / FP64_MFMA_PEAK_TFLOPS
        if "achieved" not in roof:
            roof.update({k: v for k, v in grp.items()})
        roof["group_launch"] = grp
    roof["frac"] = roof["achieved"] / FP64_MFMA_PEAK_TFLOPS
    roof.setdefault("traffic", None)

    # per-phase rooflines from SURVEY.md 8(d)'s algorithmic figures (HIP-event phase times of the handle)
    n_cond = nx + m
    n_r = int(info["refinement_rounds"])

    def phases(al, inst):
        t_factor = float(al[1] + al[7] + al[3])                       # cone pivots + Schur complement + LDL^T of S
        t_solve = float(al[2]) - t_factor                              # condensed solves + recovery + refinement residuals
        f_survey = inst * n_cond ** 3 / 3.0                            # dense n^3/3 of 8(d)
        f_exec = inst * (flops1 + nx ** 3 / 3.0)                       # what the constraint-first order executes
        b_solves = inst * (1 + n_r) * 2 * 8 * n_cond * (n_cond + 1) / 2
        b_resid = inst * (1 + n_r) * 8.0 * (nx * nx + ne * nx + nc * nx)
        return {"instances": inst, "whole_step_ms": float(al[6]),
                "factor": {"ms": t_factor, "schur_ms": float(al[7]), "ldl_ms": float(al[3]), "bound": "mfma", "flops_survey_n3_over_3": f_survey,
                           "flops_executed": f_exec, "achieved_TFLOPs_survey": f_survey / t_factor * 1e-9,
                           "achieved_TFLOPs_executed": f_exec / t_factor * 1e-9, "frac_executed": f_exec / t_factor * 1e-9 / FP64_MFMA_PEAK_TFLOPS},
                "solve_and_refine": {"ms": t_solve, "bound": "hbm", "bytes_survey": b_solves + b_resid, "solves": 1 + n_r,
                                     "achieved_GBs_survey": (b_solves + b_resid) / t_solve * 1e-6,
                                     "frac_survey": (b_solves + b_resid) / t_solve * 1e-6 / 8000.0}}
    cfg_phases = {}
    if tot_single:
        al1 = np.zeros(9); al1[7] = np.mean(sch_single); al1[3] = np.mean(ldl_single); al1[2] = np.mean(sd_single); al1[6] = np.mean(tot_single)
        al1[1] = single.phase_times()[1]
        cfg_phases["single_system"] = phases(al1, 1)
    if alone:
        cfg_phases["one_group_alone"] = phases(np.mean(np.asarray(alone), axis=0), G)

    kind = ("stage-structured (%d stages, %s treatment) " % (staged[0], "dense" if args.dense_structure else ("banded, stage-parallel multifrontal LDL^T of S"
            if stage_parallel and "levels" in stage_parallel else "banded"))) if staged else "dense "
    batched = None
    if batched_elapsed is not None:
        brate = world * B * P / batched_elapsed
        batched = {"newton_steps_per_s": brate, "problems_per_s_of_10_steps": brate / 10.0, "instances_per_gpu": B, "instances_per_group": G,
                   "groups_in_flight": batch.lanes, "passes": P, "lanes_synchronised_per_pass": bool(args.lockstep_passes), "ms_per_pass": 1e3 * batched_elapsed / P, "one_group_alone_steps_per_s": unit_rate,
                   "scaling": "weak (instances sharded block-contiguously over ranks, no data-path collective)"}
    out = {
        "metric": "Newton steps/sec (n~5k KKT)", "value": value, "unit": "Newton steps/s", "n_gpus": world, "steps": steps_timed,
        "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / steps_timed, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": ("%s synthetic " + kind + "conic QP: nx=%d ne=%d nc=%d (%d R+ + %d x SOC%d), n=%d condensed, N=%d unreduced; "
                                "%s; 1 LDL^T factorisation, %d refinement round(s) per step") % (
                                   args.config, nx, ne, nc, n_nn, n_soc, dim, nx + m, nx + 2 * ne + 3 * nc,
                                   "ONE system per GPU stepped sequentially (replicas at N > 1)" if single_elapsed is not None
                                   else "%d independent instances per GPU (batched rate, --no-single)" % B, info["refinement_rounds"]),
                   "parallelism": "one system per GPU: replicas only; batched: independent problems per GPU (no data-path collective)",
                   "refinement_rounds": info["refinement_rounds"], "factorizations_per_step": info["factorizations"],
                   "batched": batched, "roofline_phases": cfg_phases},
        "roofline": roof,
    }
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
