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
