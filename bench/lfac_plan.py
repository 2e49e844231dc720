#!/usr/bin/env python3
"""The launch plan of the left-looking factorisation (csrc/lfac.hip) for a shape, without a GPU: python bench/lfac_plan.py [nx ne nc [budget_us [head_us]]]"""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
L = ctypes.CDLL(os.path.join(ROOT, "calipso.jl_amd", "libcalipso_hip.so"))
f = L.calipso_hip_debug_lfac_plan
f.argtypes = [ctypes.c_int32] * 4 + [ctypes.c_double, ctypes.c_double, ctypes.c_void_p, ctypes.c_int32]
f.restype = ctypes.c_int32
a = sys.argv[1:]
nx, ne, nc = (int(a[0]), int(a[1]), int(a[2])) if len(a) >= 3 else (2500, 1500, 1000)
budget = float(a[3]) if len(a) > 3 else 19.0
head = float(a[4]) if len(a) > 4 else 0.0
NP = (nx + 511) // 512 * 512 if nx > 512 else nx
out = np.zeros(6 * 256)
n = f(NP // 64, nx, ne, nc, budget, head, out.ctypes.data, 256)
print("nx %d ne %d nc %d NP %d budget %.1f us (0: the library's scan; chosen budget %.1f margin %d): %d launches" % (nx, ne, nc, NP, budget, out[6 * 255], int(out[6 * 255 + 1]), n))
o = out[:6 * n].reshape(n, 6)
for l in range(n):
    print("launch %3d  longest worker %6.1f us  items %4d (schur %4d far %4d row %3d)  mean worker %5.1f" % (l - 2, o[l, 0], o[l, 1], o[l, 2], o[l, 3], o[l, 4], o[l, 5]))
if n:
    print("sum of max(longest worker, 19.4): %.0f us (head %.0f)" % (sum(max(v, 19.4) for v in o[1:, 0]) + o[0, 0], o[0, 0]))
