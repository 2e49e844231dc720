# wall-clock time of repeated Newton steps of ONE C3 system (no phase queries in the loop): python bench/_steptime.py
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
from __graft_entry__ import load_package
pkg = load_package()
import problems as pr
import bench
prob, pt, lam, w, s = bench.make_instance(pkg, pr, 0, bench.CONFIGS["C3"], 0)
for _ in range(4):
    s.newton_step(advance=False)
s.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(20):
        s.newton_step(advance=False)
    s.synchronize()
    print("%.3f ms per step" % ((time.perf_counter() - t0) / 20 * 1e3))
