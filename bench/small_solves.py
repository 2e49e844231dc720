import sys, time
import os; R=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0]=[R, os.path.join(R,'tests')]
import numpy as np
from helpers import load_pkg
import problems as pr
pkg=load_pkg()
for name, prob in (("wachter C1", pr.wachter()), ("pendulum C2", pr.pendulum(action_guess=np.zeros(10)))):
    s = pkg.Solver(prob, prob.nx, prob.np, prob.ne, prob.nc)
    pkg.initialize_b(s, prob.x0); pkg.solve_b(s)      # warm
    t0=time.perf_counter(); pkg.initialize_b(s, prob.x0); ok=pkg.solve_b(s); dt=time.perf_counter()-t0
    st=s.stats()
    # time spent in python callbacks
    print(name, "ok", ok, "iters", st["total_iterations"], "factorizations", st["factorizations"], "wall ms %.2f" % (dt*1e3), "per iteration ms %.3f" % (dt*1e3/st["total_iterations"]))

# differentiate! with p right-hand sides (BASELINE config C5: cart-pole auto-tuning, p = 102)
prob = pr.cartpole_mpc()
s = pkg.Solver(prob, prob.nx, prob.np, prob.ne, prob.nc, parameters=prob.parameters, options=dict(differentiate=0))
pkg.initialize_b(s, prob.x0); ok = pkg.solve_b(s)
s.differentiate()
t0 = time.perf_counter(); s.differentiate(); dt = time.perf_counter() - t0
print("cartpole C5 solve ok", ok, "differentiate (nx=%d ne=%d, p=%d columns) wall ms %.2f" % (prob.nx, prob.ne, prob.np, dt * 1e3))
prob = pr.parametric_conic_qp(1500, 400, 100, 50, 3, seed=1)
from helpers import interior_point
pt, lam = interior_point(prob, 3)
s = pkg.Solver(prob, prob.nx, prob.np, prob.ne, prob.nc, parameters=prob.parameters, nonnegative_indices=prob.nonnegative_indices, second_order_indices=prob.second_order_indices)
s.set("solution", np.concatenate([pt[k] for k in "xrsyzt"])); s.set("dual", lam)
for name, v in (("central_path", 0.17), ("penalty", 52.0), ("primal_regularization", 1e-5), ("dual_regularization", 1e-5)):
    s.set(name, [v])
s.differentiate()
t0 = time.perf_counter(); s.differentiate(); dt = time.perf_counter() - t0
print("parametric conic QP nx=1500 ne=400 nc=250: differentiate p=%d columns wall ms %.2f (includes the python evaluation callback)" % (prob.np, dt * 1e3))

# a batch of C2 problems: one after the other vs one lockstep group (evaluation through the python callbacks either way)
B = 16
def mk(k):
    p = pr.pendulum(action_guess=0.05 * k * np.ones(10))
    s = pkg.Solver(p, p.nx, p.np, p.ne, p.nc)
    return p, s
inst = [mk(k) for k in range(B)]
grp = pkg.Group([s for _, s in inst])
for rep in range(2):
    for p, s in inst: pkg.initialize_b(s, p.x0)
    t0 = time.perf_counter(); ok1 = [pkg.solve_b(s) for _, s in inst]; t1 = time.perf_counter() - t0
    for p, s in inst: pkg.initialize_b(s, p.x0)
    t0 = time.perf_counter(); ok2 = grp.solve(); t2 = time.perf_counter() - t0
print("%d pendulum (C2) solves: one after the other %.1f ms, lockstep group %.1f ms (python evaluation callbacks in both)" % (B, t1 * 1e3, t2 * 1e3), all(ok1), all(ok2))
