#!/usr/bin/env python3
"""bench.py — Newton steps/s of the CALIPSO KKT hot path on MI355X (BASELINE.json metric).

A "step" = one inner Newton iteration of solve! (src/solver/solve.jl:98-353) on each of the B independent problem
instances this rank holds: evaluate (QP mat-vecs on the device) -> cone! -> residual! -> inertia-corrected LDL^T of the
condensed KKT matrix -> condensed solve + step recovery -> >= 1 refinement round against the unreduced system -> cone
fraction-to-boundary search -> candidate merit / violation -> filter line-search decision.  Inputs are resident in HBM
before the timed region.  The B instances of a rank are bound into groups of G (default 36 = 3 x 12): a group steps its members
in lockstep through the same kernel launches (the problem instance is a grid dimension of every kernel), three groups are in
flight on three HIP streams.  ms_per_step is the time of one such pass over all B instances.  Workload = BASELINE config C3 (synthetic dense conic QP, nx=2500, ne=1500, nc=400 R+ + 200 x SOC3
=> n = 5000 condensed, N = 8500 unreduced), problem ids  rank*B .. rank*B+B-1  (SplitMix64 streams).

  python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run, one rank per GPU)

Multi-GPU: problems are independent => ranks share nothing on the data path (weak scaling); torch.distributed (RCCL) is
used only for the barrier and the max-over-ranks time.  Prints ONE JSON line on rank 0.
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
FP64_MFMA_PEAK_TFLOPS = 78.6   # MI355X fp64 matrix peak (datasheet; bench/mfma_f64_peak.hip measures the achievable ceiling)


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


def cpu_baseline(shape, name="C3", staged=None):
    """The oracle (faithful single-thread restatement of the reference's CPU path) on ONE Newton step of the same C3
    problem (problem id 0): search_direction! = assemble + sparse up-looking LDL^T (QDLDL order of operations, constraint-first
    permutation) + solve + refinement, with ONE factorisation per step (the reference re-factorises before every solve —
    linear_solver.jl:53 — so this is favourable to it)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle
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
    t0 = time.perf_counter()
    prob.evaluate(pr.ALL_VARIABLE_FLAGS, pt["x"], pt["y"], pt["z"], np.zeros(0), o.buf)
    o.cone(product=True, jacobian=True, target=True, barrier=True, barrier_gradient=True)
    o.residual()
    rc = o.search_direction()
    dt = time.perf_counter() - t0
    st = o.stats()
    note = "" if staged is None else (" (the port assembles and factors the blocks densely: it does not exploit the stage structure, which the "
                                      "reference's sparse LDL^T would)")
    return dict(value=1.0 / dt, unit="Newton steps/s", cores=1, kind="port",
                sample=note.strip() + (" " if note else "") + "1 Newton step (evaluate + cone + residual + search_direction: 1 LDL^T factorisation, %d solves) of %s problem 0, %.1f s; "
                       "with the reference's re-factorisation before every solve it would be %dx the factorisation time" % (
                           1 + st["last_refinement_rounds"], name, dt, 1 + st["last_refinement_rounds"]),
                status=int(rc))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=36, help="independent problem instances per GPU (B), arranged in groups of --group members;\n"
                    "--lanes groups are in flight at a time, see calipso.jl_amd/batch.py")
    ap.add_argument("--lanes", type=int, default=3, help="host threads / HIP streams driving the units (groups or single instances) concurrently")
    ap.add_argument("--group", type=int, default=12, help="instances per group: the members of a group are stepped in lockstep through the same\n"
                    "kernel launches (calipso_hip_group_*); --batch must be a multiple of it")
    ap.add_argument("--config", default="C3", choices=list(CONFIGS) + list(STAGED))
    ap.add_argument("--dense-structure", action="store_true", help="stage-structured configs: keep the dense treatment (no calipso_hip_analyze_structure)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-single", action="store_true", help="skip the informational single-instance loop (profiling runs: every launch\n"
                    "in the trace then carries a whole group)")
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
    B = args.batch
    from calipso_jl_amd.batch import BatchSolver, gather_results, shard_range
    G = max(1, args.group)
    assert B % G == 0, "--batch must be a multiple of --group"
    ids = list(shard_range(world * B, rank, world))          # block-contiguous problem ids of this rank
    # creation order: the first member of every unit first, so that the streams that carry the launches get distinct priority
    # classes (handles take class = creation index mod 3, calipso_hip_create)
    order = [k for k in range(B) if k % G == 0] + [k for k in range(B) if k % G != 0]
    made = {}
    for k in order:
        inst = make_instance(pkg, pr, ids[k], shape, local_rank, staged, not args.dense_structure)
        # the dense host copies of the problem data (~100 MB per C3 instance) are only needed until they are on the device
        made[k] = inst if k == 0 else (None, None, None, None, inst[4])
        if k != 0:
            inst[4].problem = None
            inst[4].methods = None
    solvers = [made[k][4] for k in range(B)]
    units = [pkg.Group(solvers[k:k + G]) for k in range(0, B, G)] if G > 1 else solvers
    batch = BatchSolver(units, lanes=args.lanes)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        for s in solvers:
            s.synchronize()

    def one_step():
        out = batch.newton_step(advance=False)              # units run concurrently, one HIP stream each
        return [i for u in out for i in u] if G > 1 else out

    for _ in range(args.warmup):
        one_step()
    # single-instance latency rate (informational): instance 0 alone, same step
    barrier()
    n_single = max(3, min(10, args.steps))
    single_rate = None
    if not args.no_single:
        for _ in range(2):
            solvers[0].newton_step(advance=False)            # (first single-handle call captures its launch graphs)
        solvers[0].synchronize()
        ts = time.perf_counter()
        for _ in range(n_single):
            solvers[0].newton_step(advance=False)
        solvers[0].synchronize()
        single_rate = n_single / (time.perf_counter() - ts)
    # unit 0 alone (one group of G instances, or one instance): its launches have the device to themselves, so the HIP-event
    # durations of its kernels are clean per-launch figures (the roofline below uses them)
    barrier()
    ts = time.perf_counter()
    sch_alone, alone = [], []
    for _ in range(n_single):
        units[0].newton_step(advance=False)
        alone.append(units[0].phase_times())
        sch_alone.append(alone[-1][7])
    units[0].synchronize()
    unit_rate = G * n_single / (time.perf_counter() - ts)
    barrier()
    t0 = time.perf_counter()
    sch, ldl, tot, sd = [], [], [], []
    for _ in range(args.steps):
        infos = one_step()
        pt = solvers[0].phase_times()
        sch.append(pt[7]); ldl.append(pt[3]); tot.append(pt[6]); sd.append(pt[2])
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if args.dist_backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    # post-round exchange (outside the data path): per-problem status rows all-gathered, step counters all-reduced
    status = [[int(i["status"] >= 0), args.steps, i["refinement_rounds"], i["factorizations"]] for i in infos]
    all_status, counters = gather_results(status, [float(len(solvers) * args.steps)])
    assert all_status.shape[0] == world * B and int(counters[0]) == world * B * args.steps
    info = infos[0]
    nx, ne, n_nn, n_soc, dim = shape
    nc = n_nn + n_soc * dim
    m = ne + nc
    value = world * B * args.steps / elapsed
    # dominant kernel: the Schur-complement update S = Lxx + ep*I + Z' Omega Z on the fp64 matrix cores (k_schur); one launch
    # covers the G instances of a group.  algorithmic flops per launch = G x multiply-adds of the lower triangle incl. diagonal,
    # nx (nx+1) m each.  Its launch duration is taken from HIP events on the stream while ONE unit is in flight (the kernel has
    # the device to itself; with several units in flight concurrent launches share the CUs and a per-launch time is ill-defined)
    sch_ms = float(np.mean(sch_alone))
    sch_ms_concurrent = float(np.mean(sch))
    # (per constraint row with w non-zero columns: w (w + 1) multiply-add flops of the lower triangle; dense rows: nx (nx + 1) each)
    prob0 = made[0][0]
    wrow = np.concatenate([np.count_nonzero(prob0.A, axis=1), np.count_nonzero(prob0.G, axis=1)]).astype(np.float64)
    flops1 = float(np.sum(wrow * (wrow + 1.0)))
    flops = G * flops1
    achieved = flops / (sch_ms * 1e-3) * 1e-12
    # per-phase rooflines from SURVEY.md 8(d)'s algorithmic figures, one instance in flight (HIP-event phase times of the handle)
    al = np.mean(np.asarray(alone), axis=0)
    n_cond = nx + m
    n_r = int(info["refinement_rounds"])
    t_factor = float(al[1] + al[7] + al[3])                       # cone pivots + Schur complement + LDL^T of S (G instances)
    t_solve = float(al[2]) - t_factor                              # condensed solves + recovery + refinement residuals
    f_survey = G * n_cond ** 3 / 3.0                               # dense n^3/3 of 8(d), per instance
    f_exec = G * (flops1 + nx ** 3 / 3.0)                          # what the constraint-first order executes
    b_solves = G * (1 + n_r) * 2 * 8 * n_cond * (n_cond + 1) / 2   # 8(d): each solve reads the factor twice
    b_resid = G * (1 + n_r) * 8.0 * (nx * nx + ne * nx + nc * nx)  # 8(d): matrix-free R - H*step per refinement residual
    phases = {
        "factor": {"ms": t_factor, "bound": "mfma", "flops_survey_n3_over_3": f_survey, "flops_executed": f_exec,
                   "achieved_TFLOPs_survey": f_survey / t_factor * 1e-9, "achieved_TFLOPs_executed": f_exec / t_factor * 1e-9,
                   "frac_survey": f_survey / t_factor * 1e-9 / FP64_MFMA_PEAK_TFLOPS, "frac_executed": f_exec / t_factor * 1e-9 / FP64_MFMA_PEAK_TFLOPS},
        "solve_and_refine": {"ms": t_solve, "bound": "hbm", "bytes_survey": b_solves + b_resid, "solves": 1 + n_r,
                             "achieved_GBs_survey": (b_solves + b_resid) / t_solve * 1e-6, "frac_survey": (b_solves + b_resid) / t_solve * 1e-6 / 8000.0},
        "whole_step_ms": float(al[6]),
    }
    traffic = None
    try:   # HBM bytes per launch of k_schur from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, C3 shape only)
        if args.config == "C3":
            e = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_summary.json")))["calipso::k_schur"]
            traffic = e["hbm_bytes_per_launch"] * G / float(e.get("instances_per_launch", 1))
    except Exception:
        traffic = None
    kind = ("stage-structured (%d stages, %s treatment) " % (staged[0], "dense" if args.dense_structure else "banded")) if staged else "dense "
    out = {
        "metric": "Newton steps/sec (n~5k KKT)", "value": value, "unit": "Newton steps/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": ("%s synthetic " + kind + "conic QP: nx=%d ne=%d nc=%d (%d R+ + %d x SOC%d), n=%d condensed, N=%d unreduced; "
                                "%d independent instance(s) per GPU; 1 LDL^T factorisation, %d refinement round(s) per step") % (
                                   args.config, nx, ne, nc, n_nn, n_soc, dim, nx + m, nx + 2 * ne + 3 * nc, B, info["refinement_rounds"]),
                   "instances_per_gpu": B, "instances_per_group": G, "instances_in_flight": batch.lanes * G, "parallelism": "independent problems per GPU (no data-path collective)",
                   "refinement_rounds": info["refinement_rounds"], "factorizations_per_step": info["factorizations"],
                   "single_instance_steps_per_s": single_rate, "one_unit_alone_steps_per_s": unit_rate,
                   "problems_per_s_of_10_steps": value / 10.0, "roofline_phases_one_unit_alone": phases,
                   "phase_ms": {"whole_step_gpu": float(np.mean(tot)), "search_direction": float(np.mean(sd)), "schur_mfma": sch_ms,
                                "ldl_of_schur_complement": float(np.mean(ldl))}},
        "roofline": {"kernel": "k_schur (S = Lxx + eps*I + omega*gx'gx + hx'(Omega hx), v_mfma_f64_16x16x4_f64)", "bound": "mfma",
                     "achieved": achieved, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": achieved / FP64_MFMA_PEAK_TFLOPS,
                     "traffic": traffic, "flops_per_launch": flops, "instances_per_launch": G, "avg_launch_ms": sch_ms,
                     "avg_launch_ms_with_%d_units_in_flight" % batch.lanes: sch_ms_concurrent},
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(shape, args.config, staged)
    else:
        out["cpu_baseline"] = None
    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
