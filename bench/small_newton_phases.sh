#!/bin/bash
# phase clocks of one instance of the batched small-problem kernel (GPU box): rebuilds csrc/smallnewton.hip with -DSN_TRACE in place, runs, restores the plain build
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R/calipso.jl_amd/csrc
FL="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-result"
hipcc $FL -DSN_TRACE ${SN_EXTRA:-} -c smallnewton.hip -o smallnewton.o && hipcc --offload-arch=gfx950 -shared -fPIC -o ../libcalipso_hip.so *.o -ldl
cd $R
python - "$@" <<'PY'
import ctypes as C, os, sys, numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.getcwd())
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
from __graft_entry__ import load_package
import problems as pr
pkg = load_package()
from calipso_jl_amd._lib import lib
a = sys.argv[1:]
nx, ne, nc = (int(a[0]), int(a[1]), int(a[2])) if len(a) >= 3 else (49, 40, 0)
B = int(a[3]) if len(a) > 3 else 4096
K = 10
prob = pr.random_qp(nx, ne, nc, seed=1000, nonnegative_indices=list(range(1, nc + 1)))
sn = pkg.SmallNewtonBatch(nx, ne, nc, B)
if os.environ.get("SN_THREADS"): sn.set_option("threads", int(os.environ["SN_THREADS"]))
sn.set_qp(prob.P, prob.q, prob.A, prob.b, prob.G, prob.h, objective_scale=prob.c)
sn.initialize(np.tile(prob.x0, (B, 1)))
res, ms = sn.solve()
w = sn.get_state()["solution"].copy()
if nc: w[:, nx + ne:nx + ne + nc] += 0.5; w[:, -nc:] += 0.5
w[:, :nx] += 0.05
sn.set_state(w=w, scalars=np.tile([0.17, 0.99, 52.0], (B, 1)))
for Bn in (B,):
    info, st, msk = sn.steps(K, advance=False)
    out = np.zeros(12)
    f = lib().calipso_hip_debug_smallnewton_profile; f.argtypes = [C.c_void_p, C.POINTER(C.c_double)]; f(sn._h, out.ctypes.data_as(C.POINTER(C.c_double)))
    names = ["eval+residual+norms", "inertia logic", "weights + S assembly", "LDL^T trailing updates", "first solve", "refinement", "cone search + candidate", "merit + line search", "accept", "LDL^T panels (1 wave)", "-", "between steps"]
    dsc = np.zeros(4); g = lib().calipso_hip_debug_smallnewton_describe; g.argtypes = [C.c_void_p, C.POINTER(C.c_double)]; g(sn._h, dsc.ctypes.data_as(C.POINTER(C.c_double)))
    print("shape (%d, %d, %d), batch %d, %d steps, %d threads and %d B of LDS per instance, %d instances per compute unit: launch %.3f ms = %.1f us per step of a resident instance; instance 0, us per step:" % (nx, ne, nc, B, K, dsc[0], dsc[1], dsc[2], msk, msk * 1e3 / K / max(1, B / (dsc[2] * dsc[3]))))
    for n_, v in zip(names, out):
        if n_ != "-": print("   %-26s %8.2f" % (n_, v / K))
    print("   sum %.2f   rounds %.1f" % (out.sum() / K, info[0, 2]))
PY
cd $R/calipso.jl_amd/csrc && hipcc $FL -c smallnewton.hip -o smallnewton.o && hipcc --offload-arch=gfx950 -shared -fPIC -o ../libcalipso_hip.so *.o -ldl
