#!/usr/bin/env python3
"""bench/mf_trace.py for a GROUP of C4T instances: the phases of workgroup (0, 0) of every k_mf_factor launch while the other fronts and members of the level
run beside it (trace build: `make trace` in calipso.jl_amd/csrc).  python bench/mf_trace_group.py [members]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
from __graft_entry__ import load_package   # noqa: E402

pkg = load_package()
import calipso_jl_amd._lib as _lib   # noqa: E402
_lib.LIB_PATH = os.path.join(ROOT, "calipso.jl_amd", "libcalipso_hip_trace.so")
import problems as pr   # noqa: E402
import bench   # noqa: E402

G = int(sys.argv[1]) if len(sys.argv) > 1 else 64
wl = bench.Workload(pkg, pr, "C4T", 0, 1, 0, G, G, 1)
for _ in range(2):
    wl.batched_pass()
L = _lib.lib()
buf = (C.c_longlong * (64 * 12))()
L.calipso_hip_debug_mf_trace.restype = C.c_int32
L.calipso_hip_debug_mf_trace(buf, 1)
wl.batched_pass()
n = L.calipso_hip_debug_mf_trace(buf, 0)
t = np.array(buf[:], dtype=np.int64).reshape(64, 12)
print("group of %d: %d k_mf_factor launches in one pass" % (G, n))
print("%6s %4s %4s %7s %7s %7s %7s %7s %7s %8s %9s" % ("launch", "c", "m", "zero", "own", "extend", "panels", "updates", "write", "total", "core MHz"))
for k in range(min(n, 64)):
    r = t[k]
    us = lambda a, b: (r[a] - r[b]) / 100.0
    print("%6d %4d %4d %7.2f %7.2f %7.2f %7.2f %7.2f %7.2f %8.2f %9.0f" % (k, r[10], r[11], us(1, 0), us(2, 1), us(3, 2), r[8] / 100.0, r[9] / 100.0, us(5, 4), us(5, 0),
                                                                        (r[7] - r[6]) / max(us(5, 0), 1e-9)))
wl.close()
