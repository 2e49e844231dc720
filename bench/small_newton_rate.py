#!/usr/bin/env python3
"""Rate of the batched small-problem path (csrc/smallnewton.hip): python bench/small_newton_rate.py [nx ne nc [batch [steps]]] — whole solve!s of `batch` C5-shaped random
QPs (nx = 49, ne = 40: the shape of the reference's cart-pole MPC problem, examples/autotuning/cartpole.jl:85-146) in one launch, and `steps` non-advancing Newton
steps per instance in one launch; the same problems through the oracle on one host core for comparison (a sample of them)."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
from __graft_entry__ import load_package
import problems as pr

def main():
    a = sys.argv[1:]
    nx, ne, nc = (int(a[0]), int(a[1]), int(a[2])) if len(a) >= 3 else (49, 40, 0)
    B = int(a[3]) if len(a) > 3 else 4096
    K = int(a[4]) if len(a) > 4 else 20
    pkg = load_package()
    nprob = min(B, 64)                                 # distinct problems (the batch cycles through them with perturbed starting points)
    probs = [pr.random_qp(nx, ne, nc, seed=1000 + k, nonnegative_indices=list(range(1, nc + 1))) for k in range(nprob)]
    idx = np.arange(B) % nprob
    st = lambda name: np.stack([np.asarray(getattr(probs[i], name), dtype=np.float64) for i in idx])
    sn = pkg.SmallNewtonBatch(nx, ne, nc, B)
    if os.environ.get("SN_THREADS"): sn.set_option("threads", int(os.environ["SN_THREADS"]))      # threads per instance (0 / unset: chosen by the LDS footprint)
    sn.set_qp(st("P"), st("q"), st("A"), st("b"), st("G"), st("h"), objective_scale=probs[0].c, shared=False)
    rng = np.random.default_rng(0)
    x0 = np.stack([probs[i].x0 for i in idx]) + 0.01 * rng.standard_normal((B, nx))
    import ctypes as C
    from calipso_jl_amd._lib import lib
    dsc = np.zeros(4); f = lib().calipso_hip_debug_smallnewton_describe; f.argtypes = [C.c_void_p, C.POINTER(C.c_double)]; f(sn._h, dsc.ctypes.data_as(C.POINTER(C.c_double)))
    out = {"shape": [nx, ne, nc], "n": nx + ne + nc, "batch": B, "threads": os.environ.get("SN_THREADS", "auto"),
           "kernel": {"threads_per_instance": int(dsc[0]), "lds_bytes_per_instance": int(dsc[1]), "instances_per_compute_unit": int(dsc[2]), "compute_units": int(dsc[3])}}
    ms_all = []
    for rep in range(3):
        sn.initialize(x0)
        res, ms = sn.solve()
        ms_all.append(ms)
    stt = sn.get_state()
    its = stt["counters"]["total_iterations"]; steps = stt["counters"]["newton_steps"]
    ms = min(ms_all)
    out["solve"] = {"launch_ms": ms, "launch_ms_all": ms_all, "converged": int((res == 1).sum()), "solves_per_s": B / (ms * 1e-3), "newton_steps_total": int(steps.sum()),
                    "newton_steps_per_s": float(steps.sum()) / (ms * 1e-3), "mean_iterations": float(its.mean()), "max_iterations": int(its.max()),
                    "mean_factorizations": float(stt["counters"]["factorizations"].mean()), "max_refinement_rounds": int(stt["counters"]["max_refinement_rounds"].max())}
    # non-advancing steps from an interior state (the benchmark step of the headline, for the batch)
    w = stt["solution"].copy()
    if nc:
        w[:, nx + ne:nx + ne + nc] += 0.5; w[:, -nc:] += 0.5
    w[:, :nx] += 0.05 * rng.standard_normal((B, nx))
    sn.set_state(w=w, scalars=np.tile([0.17, 0.99, 52.0], (B, 1)))
    sn.steps(2, advance=False)
    t = [sn.steps(K, advance=False) for _ in range(3)]
    msk = min(x[2] for x in t)
    info, stat = t[0][0], t[0][1]
    out["steps"] = {"count_per_instance": K, "launch_ms": msk, "newton_steps_per_s": B * K / (msk * 1e-3), "ok": int((stat == 0).sum()), "stepped": int((info[:, 6] == 0).sum()),
                    "refinement_rounds_mean": float(info[:, 2].mean()), "us_per_step_of_a_resident_instance": msk * 1e3 / K / max(1.0, B / max(1.0, dsc[2] * dsc[3]))}
    # bytes an instance moves once per launch (problem data + state): the HBM side of the roofline is irrelevant here — say so with the number
    bytes_inst = 8.0 * (nx * nx + (ne + nc) * nx + nx + (ne + nc) + 2 * (nx + 2 * ne + 3 * nc))
    out["steps"]["hbm_fraction"] = B * bytes_inst / (msk * 1e-3) / 8e12
    # the oracle on the same problems (one host core)
    try:
        import oracle
        from test_oracle_solve import run as run_oracle
        ts = []
        for k in range(min(8, nprob)):
            t0 = time.perf_counter(); o, status = run_oracle(oracle, probs[k]); ts.append(time.perf_counter() - t0)
        out["cpu_baseline"] = {"kind": "port", "cores": 1, "solve_ms_median": 1e3 * float(np.median(ts)), "solves_per_s": 1.0 / float(np.median(ts)),
                               "sample": "%d solve!s by the oracle (evaluation through Python callbacks)" % len(ts)}
        out["solve"]["gpu_over_cpu"] = out["solve"]["solves_per_s"] / out["cpu_baseline"]["solves_per_s"]
    except Exception as e:
        out["cpu_baseline"] = {"error": repr(e)}
    sn.close()
    print(json.dumps(out))

if __name__ == "__main__":
    main()
