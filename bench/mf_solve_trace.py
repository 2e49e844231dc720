#!/usr/bin/env python3
"""The phases of workgroup (0, 0) of every k_mf_forward / k_mf_backward launch of one pass over a GROUP of C4T instances (trace build: `make trace` in
calipso.jl_amd/csrc).  python bench/mf_solve_trace.py [members]"""
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

G = int(sys.argv[1]) if len(sys.argv) > 1 else 16
wl = bench.Workload(pkg, pr, "C4T", 0, 1, 0, G, G, 1)
for _ in range(2):
    wl.batched_pass()
L = _lib.lib()
buf = (C.c_longlong * (128 * 8))()
L.calipso_hip_debug_mfs_trace.restype = C.c_int32
L.calipso_hip_debug_mfs_trace(buf, 1)
wl.batched_pass()
n = L.calipso_hip_debug_mfs_trace(buf, 0)
t = np.array(buf[:], dtype=np.int64).reshape(128, 8)
print("group of %d: %d sweep launches in one pass (the first 24 shown); forward: load+fill | children | triangular | product+store; backward: load+fill | product | triangular" % (G, n))
print("%6s %4s %4s %4s %8s %8s %8s %8s %8s %10s" % ("launch", "kind", "c", "m", "fill", "ph2", "ph3", "ph4", "total", "since prev"))
prev = None
for k in range(min(n, 24)):
    r = t[k]
    us = lambda a, b: (r[a] - r[b]) / 100.0
    print("%6d %4s %4d %4d %8.2f %8.2f %8.2f %8.2f %8.2f %10.2f" % (k, "bwd" if r[6] else "fwd", r[7] & 0xffff, r[7] >> 16, us(1, 0), us(2, 1), us(3, 2), us(4, 3), us(4, 0),
                                                                    (r[0] - prev) / 100.0 if prev is not None else 0.0))
    prev = r[4]
wl.close()
