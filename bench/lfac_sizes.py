"""Left-looking (csrc/lfac.hip) against right-looking factorisation of ONE dense system at other sizes than C3: ms per Newton step and per factorisation, each in a
process of its own (the switch is read once).  python bench/lfac_sizes.py ["nx,ne,nn,nsoc,dim;..."]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, os, time
sys.path[:0] = [%(root)r, os.path.join(%(root)r, "tests")]
from helpers import load_pkg
from test_gpu_group import build
pkg = load_pkg()
shape = tuple(int(v) for v in os.environ["SHAPE"].split(","))
s = build(pkg, 77, shape)
t0 = time.perf_counter(); s.newton_step(advance=False); first = time.perf_counter() - t0
for _ in range(2): s.newton_step(advance=False)
t0 = time.perf_counter()
for _ in range(10): s.newton_step(advance=False)
dt = (time.perf_counter() - t0) / 10
kt = s.kernel_times()
import ctypes, numpy as np
from calipso_jl_amd._lib import lib
d8 = np.zeros(8); f = lib().calipso_hip_debug_lfac_describe; f.argtypes = [ctypes.c_void_p, ctypes.c_void_p]; f(s._h, d8.ctypes.data)
print("RESULT lfac %%d (-1: wanted, no plan: %%s)  first step %%.3f s (plan scan %%.1f ms)  step %%.3f ms  pivot chain / panel launches %%.3f ms (%%d launches)" %% (int(kt[6]), lib().calipso_hip_last_error(s._h).decode()[:90] if int(kt[6]) < 0 else "-", first, d8[6], 1e3 * dt, kt[0], int(kt[1])))
'''
shapes = (sys.argv[1] if len(sys.argv) > 1 else "1000,300,60,30,3;1500,300,60,30,3;2500,1500,500,250,2;4000,2000,600,200,3;6000,3000,1000,300,3").split(";")
for sh in shapes:
    for lf in ("1", "0"):
        e = dict(os.environ, SHAPE=sh, CALIPSO_HIP_LFAC=lf)
        r = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT}], env=e, capture_output=True, text=True, timeout=900)
        line = [l for l in r.stdout.splitlines() if l.startswith("RESULT")]
        print("%-28s LFAC=%s  %s" % (sh, lf, line[0][7:] if line else "FAILED: " + r.stderr[-300:]))
