#!/usr/bin/env python3
"""Rate of the batched LDS-resident small-system path (csrc/small.hip) on the shapes of BASELINE configs C5 (cart-pole MPC sensitivities:
n = 89, 102 right-hand sides) and C2 (pendulum: n = 56): instances per second of ONE launch, inputs resident in HBM, next to the general
device path (calipso_hip_ldl_*: one instance at a time) and to LAPACK on the host cores.   python bench/small_batch_rate.py > gpurun_out/..."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
from __graft_entry__ import load_package  # noqa: E402

pkg = load_package()
import scipy.sparse as sp  # noqa: E402
from scipy.linalg import lapack  # noqa: E402

out = {}
rng = np.random.default_rng(0)
for name, n, nx, nrhs in (("C5_cartpole_n89_p102", 89, 49, 102), ("C2_pendulum_n56_p1", 56, 32, 1)):
    Q = rng.standard_normal((n, n))
    K = Q @ Q.T + n * np.eye(n)
    K[nx:, nx:] = -K[nx:, nx:]; K[:nx, nx:] *= 0.1; K[nx:, :nx] = K[:nx, nx:].T
    rows = {}
    for batch in (256, 1024, 4096):
        B = rng.standard_normal((batch, n, nrhs))
        sb = pkg.SmallBatch(n, nrhs, batch)
        sb.set(np.repeat(K[None], batch, axis=0), B)
        sb.solve()
        ms = min(sb.solve() for _ in range(5))
        X, inr, bad = sb.get()
        assert bad == 0 and np.abs(K @ X[batch - 1] - B[batch - 1]).max() < 1e-8
        rows["batch_%d" % batch] = dict(launch_ms=ms, instances_per_s=batch / (ms * 1e-3), solves_per_s=batch * nrhs / (ms * 1e-3))
        sb.close()
    # the general device path, one instance per call (factor + nrhs solves)
    ls = pkg.LDLSolver(n)
    Ks = sp.csc_matrix(K)
    b1 = rng.standard_normal((n, nrhs))
    ls.factorize(Ks); ls.linear_solve(b1)
    t0 = time.perf_counter()
    for _ in range(5):
        ls.factorize(Ks); ls.linear_solve(b1)
    rows["general_path_one_instance_per_call"] = dict(instances_per_s=5 / (time.perf_counter() - t0))
    # LAPACK dsytrf/dsytrs on the host, one instance after the other (single call each)
    Kf = np.asfortranarray(K)
    t0 = time.perf_counter()
    reps = 200
    for _ in range(reps):
        ldu, ipiv, info = lapack.dsytrf(Kf, lower=0)
        x, info = lapack.dsytrs(ldu, ipiv, b1, lower=0)
    rows["host_lapack_dsytrf_dsytrs"] = dict(instances_per_s=reps / (time.perf_counter() - t0), note="one host thread's call sequence, BLAS threads as configured")
    out[name] = rows
print(json.dumps(out, indent=1))
