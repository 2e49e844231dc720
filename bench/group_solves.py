"""full solves (solve!, solve.jl:8-377) of B synthetic conic QPs of one shape: one after the other vs one lockstep group.
python bench/group_solves.py [B] [nx ne n_nn n_soc dim]"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0] = [R, os.path.join(R, "tests")]
import numpy as np
from helpers import load_pkg
import problems as pr
pkg = load_pkg()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
shape = tuple(int(v) for v in sys.argv[2:7]) if len(sys.argv) > 6 else (600, 300, 100, 50, 3)


def build(pid):
    nx, ne, n_nn, n_soc, dim = shape
    prob, pt, lam = pr.synthetic_conic_qp(pkg.splitmix_uniform, pid, nx, ne, n_nn, n_soc, dim)
    s = pkg.Solver(prob, prob.nx, 0, prob.ne, prob.nc, nonnegative_indices=prob.nonnegative_indices, second_order_indices=prob.second_order_indices)
    s.qp_attach(prob.P, prob.q, prob.A, prob.b, prob.G, prob.h, 0.5)
    x0 = np.concatenate([pt["x"], np.zeros(s.N - nx)])
    return s, x0


inst = [build(100 + k) for k in range(B)]
g = pkg.Group([s for s, _ in inst])
for rep in range(2):                       # second pass is the timed one (first pass warms code objects / graphs)
    for s, x0 in inst: s.set("solution", x0)
    t0 = time.perf_counter(); ok1 = [pkg.solve_b(s) for s, _ in inst]; t_single = time.perf_counter() - t0
    it = [s.stats()["total_iterations"] for s, _ in inst]
    for s, x0 in inst: s.set("solution", x0)
    t0 = time.perf_counter(); ok2 = g.solve(); t_group = time.perf_counter() - t0
print("shape nx=%d ne=%d nc=%d: %d solves, iterations %d..%d | one after the other %.1f ms (%.1f solves/s) | lockstep group %.1f ms (%.1f solves/s)" % (
    shape[0], shape[1], shape[2] + shape[3] * shape[4], B, min(it), max(it), 1e3 * t_single, B / t_single, 1e3 * t_group, B / t_group),
    "all converged" if all(ok1) and all(ok2) else "NOT all converged")
