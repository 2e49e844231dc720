#!/usr/bin/env python3
"""Where a worker workgroup of the group's two-panel pass k_ldl_step<2> spends a tile (first pass of a group of 12 C3 instances; trace build:
`make trace` in calipso.jl_amd/csrc).  Core-clock stamps (s_memtime) of thread 0, cycles per phase:
  formZ    Z = A(i, panels) M (only when the tile row changes)
  stage    this tile's column panels written to LDS
  barrier  all 16 wavefronts arrived
  mfma     next tile's loads issued, 2 x 16 MFMAs, accumulators subtracted
  store    the tile stored, the next tile's entries awaited
  barrier2 operand reads done"""
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

G = int(sys.argv[1]) if len(sys.argv) > 1 else 12
wl = bench.Workload(pkg, pr, "C3", 0, 1, 0, G, G, 1)
for s_ in wl.solvers:
    s_.set_solve_block(512) if hasattr(s_, "set_solve_block") else None
for _ in range(3):
    wl.batched_pass()
wl.sync()
L = _lib.lib()
buf = (C.c_longlong * (32 * 8 + 32 * 16 * 2))()
L.calipso_hip_debug_ldl_bulk_trace.restype = C.c_int32
assert L.calipso_hip_debug_ldl_bulk_trace(buf) == 0
t = np.array(buf[:32 * 8], dtype=np.int64).reshape(32, 8)
wv = np.array(buf[32 * 8:], dtype=np.int64).reshape(32, 16, 2)
print("%4s %7s %7s %7s %7s %7s %7s %8s   (core clocks; 2.4 GHz)" % ("tile", "formZ", "stage", "barrier", "mfma", "store", "barr2", "total"))
rows = []
for k in range(32):
    if t[k, 6] == 0:
        break
    r = [t[k, 1] - t[k, 0], t[k, 2] - t[k, 1], t[k, 3] - t[k, 2], t[k, 4] - t[k, 3], t[k, 5] - t[k, 4], t[k, 6] - t[k, 5]]
    rows.append(r + [t[k, 6] - t[k, 0]])
    print("%4d " % k + " ".join("%7d" % v for v in rows[-1]))
if rows:
    m = np.mean(np.array(rows, dtype=float), axis=0)
    print("mean " + " ".join("%7.0f" % v for v in m))
    print("matrix-core floor per tile: 4 wavefronts per SIMD x 32 MFMAs x 64 cycles = 8192")
print("MFMA phase per wavefront of tiles 8..10 (start, end relative to thread 0's barrier exit); SIMD = wavefront % 4")
for k in (8, 9, 10):
    print("tile %d: " % k + " ".join("w%d[%d..%d]" % (w, wv[k, w, 0] - t[k, 3], wv[k, w, 1] - t[k, 3]) for w in range(16)))
