#!/usr/bin/env python3
"""What the items of the left-looking factorisation (csrc/lfac.hip) cost on the real kernel: synthetic item lists on a C3 handle (GPU box, repo root): python bench/lfac_items.py"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import bench as B
from helpers import load_pkg
import problems as pr
pkg = load_pkg()
prob, pt, lam, w, s = B.make_instance(pkg, pr, 0, B.CONFIGS["C3"], 0)
info = s.newton_step(advance=False)
s.synchronize()
L = s._L
f = L.calipso_hip_debug_lfac_items
f.argtypes = [ctypes.c_void_p] + [ctypes.c_int32] * 5
f.restype = ctypes.c_double
def run(kind, n, P=1, per=1, skew=0):
    return f(s._h, kind, n, P, per, skew)
print("SCHUR, one item per worker, all workers in step:")
for n in (79, 40, 20, 10, 5):
    t = run(0, n); print("  %2d stages: %7.1f us  (%.2f us/stage)" % (n, t, t / n))
print("SCHUR, every worker at a stage range of its own:")
for n in (40, 20, 10):
    t = run(0, n, skew=1); print("  %2d stages: %7.1f us  (%.2f us/stage)" % (n, t, t / n))
print("SCHUR split over P workgroups:")
for P in (2, 4):
    for n in (79, 40, 20):
        t = run(0, n, P=P); print("  P %d %2d stages: %7.1f us  (%.2f us/stage)" % (P, n, t, t / n))
print("SCHUR, two items of 10 stages per worker: %.1f us" % run(0, 10, per=2))
print("FAR:")
for n in (1, 2, 5, 10):
    t = run(1, n); print("  %2d panels: %7.1f us  (%.2f us/panel)" % (n, t, t / n))
print("ROW:")
for n in (0, 1, 3):
    t = run(2, n); print("  %d pending: %7.1f us" % (n, t))
