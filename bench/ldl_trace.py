#!/usr/bin/env python3
"""Timeline of the pivot chain of the LDL^T of S (one C3 instance): per panel step, 100 MHz wall-clock stamps of the workgroup that carries
tile 0 + the diagonal block (csrc/ldl.hip built with -DCALIPSO_LDL_TRACE into libcalipso_hip_trace.so: `make trace` in calipso.jl_amd/csrc).
  gap     end of the previous launch's chain workgroup -> entry of this launch's   (kernel boundary)
  formZ   entry -> Z = A(k+1,k) M_k in LDS (one global round trip + 16 MFMAs)
  tile    -> tile (k+1,k+1) updated and handed to the diagonal block
  ldl     -> the 64 pivots (four rounds of 16 columns; most of X = L^-1 is assembled meanwhile)
  inv     -> the last blocks of X = L^-1 (two short phases)
  M       -> M = X' D^-1 X on the matrix cores
  stores  -> D, inertia counts, L, X, M issued to global memory"""
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

name = sys.argv[1] if len(sys.argv) > 1 else "C3"
prob, pt, lam, w, s = bench.make_instance(pkg, pr, 0, bench.CONFIGS[name], 0)
if os.environ.get("LDL_TRACE_SOLVE_BLOCK"):
    s.set_option("solve_block", int(os.environ["LDL_TRACE_SOLVE_BLOCK"]))
for _ in range(4):
    s.newton_step(advance=False)
s.synchronize()
L = _lib.lib()
buf = (C.c_longlong * (64 * 16))()
L.calipso_hip_debug_ldl_trace.restype = C.c_int32
assert L.calipso_hip_debug_ldl_trace(buf) == 0
t = np.array(buf[:], dtype=np.int64).reshape(64, 16)
nb = s.padded_nx() // 64
us = lambda a, b: (a - b) / 100.0
print("%4s %7s %7s %7s %7s %7s %7s %7s %8s" % ("blk", "gap", "formZ", "tile", "ldl", "inv", "M", "stores", "launch"))
rows = []
for k in range(1, nb):
    r = [us(t[k, 0], t[k - 1, 5]), us(t[k, 1], t[k, 0]), us(t[k, 2], t[k, 1]), us(t[k, 3], t[k, 2]), us(t[k, 4], t[k, 3]), us(t[k, 6], t[k, 4]), us(t[k, 5], t[k, 6])]
    rows.append(r + [us(t[k, 5], t[k, 0])])
    if k <= 6 or k >= nb - 3 or os.environ.get("LDL_TRACE_ALL"):
        print("%4d " % k + " ".join("%7.2f" % v for v in rows[-1]))
m = np.mean(np.array(rows), axis=0)
print("mean " + " ".join("%7.2f" % v for v in m))
print("chain: block 0 start .. last block end = %.1f us over %d panel steps; sum of means per step %.2f us" % (us(t[nb - 1, 5], t[0, 2]), nb - 1, float(np.sum(m[:7]))))
