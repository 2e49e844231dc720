"""Fronts beyond the LDS through many workgroups (csrc/sparse_wide.hpp) against the one-workgroup kernel: factorisation times of
   (a) one dense block of N (N / 64 chained fronts of up to N rows), (b) two meshes joined through a separator of N vertices.
   python bench/wide_fronts.py [N] [repeat]      (under rocprofv3 --kernel-trace --stats for the per-kernel split)"""
import ctypes as C
import os
import sys

import numpy as np
import scipy.sparse as sp

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from __graft_entry__ import load_package   # noqa: E402

pkg = load_package()

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
R = int(sys.argv[2]) if len(sys.argv) > 2 else 5
modes = [int(v) for v in sys.argv[3].split(",")] if len(sys.argv) > 3 else [1, 0]
fn = pkg._lib.lib().calipso_hip_debug_wide_fronts
fn.restype = C.c_int32
rng = np.random.default_rng(0)
M = rng.standard_normal((N, N)); Kd = M @ M.T + N * np.eye(N)
Ad = sp.csc_matrix(np.triu(Kd))
gm = 60
Tm = sp.diags([-1.0, 2.5, -1.0], [-1, 0, 1], shape=(gm, gm), format="csc")
Km = (sp.kron(sp.identity(gm), Tm) + sp.kron(Tm, sp.identity(gm))).tocsc()
nm = gm * gm
Ssep = sp.diags([-0.5, 6.0, -0.5], [-1, 0, 1], shape=(N, N), format="lil")
for off in range(2, 40):
    Ssep.setdiag(-0.01, off); Ssep.setdiag(-0.01, -off)
C1 = sp.lil_matrix((N, nm)); C2 = sp.lil_matrix((N, nm))
for i in range(N):
    C1[i, (i * 7) % nm] = -0.3; C2[i, (i * 11) % nm] = -0.3
Kw = sp.bmat([[Km, None, C1.T], [None, Km, C2.T], [C1, C2, Ssep]], format="csc"); Kw.sort_indices()
Aw = sp.triu(Kw).tocsc()
for name, A, K in (("dense block of %d" % N, Ad, Kd), ("two 60 x 60 meshes + separator of %d" % N, Aw, Kw)):
    b = rng.standard_normal(A.shape[0])
    for on in modes:
        fn(C.c_int32(on))
        S = pkg.SparseLDL(A, method="nested_dissection")
        ts, tv = [], []
        for _ in range(R):
            assert S.factorize(A) == 0
            x = S.solve(b)
            t = S.timing(); ts.append(t[0]); tv.append(t[1])
        err = np.abs(K @ x - b).max() / max(1.0, np.abs(x).max())
        print("%-44s %-26s factor %8.3f ms   solve %7.3f ms   launches %4d   residual %.1e" % (
            name, "many workgroups per front" if on else "one workgroup per front", min(ts), min(tv), S.info["launches"], err))
        S.close()
fn(C.c_int32(1))
