"""CPU-side checks of the drop-in boundary (no GPU needed): the shared library loads, exports every symbol that
include/calipso_hip.h declares, the ctypes table covers exactly those symbols, and the library refuses to work without a
HIP device instead of falling back to a CPU path."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from helpers import ROOT, load_pkg

HEADER = os.path.join(ROOT, "include", "calipso_hip.h")


def declared_symbols():
    txt = open(HEADER).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(calipso_hip_[a-z_0-9]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    pkg = load_pkg()
    from calipso_jl_amd._lib import LIB_PATH, SYMBOLS, lib
    assert os.path.exists(LIB_PATH), "libcalipso_hip.so must be built in-tree (python __graft_entry__.py build)"
    L = lib()
    decl = declared_symbols()
    assert len(decl) >= 35
    for name in decl:
        assert hasattr(L, name), "missing export: " + name
    assert sorted(SYMBOLS) == decl            # the ctypes binding table is exactly the header


def test_host_side_functions_work_without_gpu():
    pkg = load_pkg()
    from calipso_jl_amd._lib import lib
    L = lib()
    assert L.calipso_hip_version().startswith(b"calipso-hip")
    u = pkg.splitmix_uniform(7, 3, -1.0, 1.0, 5)
    # SplitMix64 reference values (seed 0xCA11B50000000000 + 4096*7 + 3), independent python implementation
    state = (0xCA11B50000000000 + 4096 * 7 + 3) & (2**64 - 1)
    ref = []
    for _ in range(5):
        state = (state + 0x9E3779B97F4A7C15) & (2**64 - 1)
        z = state
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & (2**64 - 1)
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & (2**64 - 1)
        z = z ^ (z >> 31)
        ref.append(-1.0 + 2.0 * ((z >> 11) * 2.0**-53))
    assert np.array_equal(u, np.array(ref))


def test_no_cpu_fallback():
    """without a HIP device the product must fail loudly (there is no CPU path behind the C ABI)"""
    pkg = load_pkg()
    from calipso_jl_amd._lib import lib
    if lib().calipso_hip_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(pkg.CalipsoHipError) as e:
        pkg.Solver(None, 3, 0, 2, 2)
    assert "no HIP device" in str(e.value)
    # every other handle type of the ABI as well: the LinearSolver seam (dense and sparse), the batched small systems
    import scipy.sparse as sp
    for make in (lambda: pkg.LDLSolver(4), lambda: pkg.SparseLDL(sp.identity(4, format="csc")), lambda: pkg.SmallBatch(4, 1, 2), lambda: pkg.SmallNewtonBatch(4, 2, 2, 3)):
        with pytest.raises(pkg.CalipsoHipError) as e:
            make()
        assert "no HIP device" in str(e.value)


def test_product_does_not_reference_oracle():
    """oracle/ is test infrastructure: nothing under calipso.jl_amd/ may import, link or mention it"""
    pk = os.path.join(ROOT, "calipso.jl_amd")
    for dp, _, files in os.walk(pk):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h", ".jl", "Makefile")):
                txt = open(os.path.join(dp, f), errors="ignore").read()
                assert "liboracle" not in txt and "import oracle" not in txt and "calipso_oracle" not in txt, os.path.join(dp, f)


def test_left_looking_plan_does_not_depend_on_the_scan_threads():
    """the plan scan of csrc/lfac.hip deals its candidates over host threads (CALIPSO_HIP_LFAC_PLAN_THREADS, default: up to 64): the winner must be the first of the cheapest in
    candidate order whatever the thread count — the launch plan of a shape (and so the schedule every handle of that shape runs) is a function of the shape alone"""
    import subprocess
    import sys
    child = r'''
import ctypes, os, sys, hashlib
import numpy as np
L = ctypes.CDLL(os.path.join(%r, "calipso.jl_amd", "libcalipso_hip.so"))
f = L.calipso_hip_debug_lfac_plan
f.argtypes = [ctypes.c_int32] * 4 + [ctypes.c_double, ctypes.c_double, ctypes.c_void_p, ctypes.c_int32]
f.restype = ctypes.c_int32
h = hashlib.sha256()
for shape in ((16, 1000, 300, 150), (24, 1500, 300, 150), (40, 2500, 1500, 1000)):
    out = np.zeros(6 * 256)
    n = f(*shape, 0.0, 0.0, out.ctypes.data, 256)
    assert n == shape[0] + 1, (shape, n)
    h.update(out.tobytes())
print("PLAN " + h.hexdigest())
''' % ROOT
    digests = []
    for threads in ("1", "3", "8"):
        r = subprocess.run([sys.executable, "-c", child], env=dict(os.environ, CALIPSO_HIP_LFAC_PLAN_THREADS=threads), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        digests.append([l for l in r.stdout.splitlines() if l.startswith("PLAN ")][0])
    assert len(set(digests)) == 1, digests
