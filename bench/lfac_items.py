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
f.argtypes = [ctypes.c_void_p] + [ctypes.c_int32] * 6
f.restype = ctypes.c_double
def run(kind, n, P=1, per=1, skew=0, active=0):
    return f(s._h, kind, n, P, per, skew, active)
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
for n in (0, 1):
    print("  half items, %d pending: first tile %.1f us, second tile %.1f us" % (n, run(3, n), run(4, n)))
print("SCHUR 40 stages, FAR 10 panels on only some of the 255 workers (the matrix cores' rate depends on how many compute units are busy):")
for act in (255, 192, 128, 64, 16):
    t = run(0, 40, active=act); u = run(1, 10, active=act); r = run(2, 1, active=act)
    print("  %3d workers: SCHUR %6.1f us (%.2f us/stage)   FAR %6.1f us (%.2f us/panel)   ROW %5.1f us" % (act, t, t / 40, u, u / 10, r))
