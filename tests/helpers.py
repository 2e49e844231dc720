"""shared test helpers: package loader (the package directory is named `calipso.jl_amd`, not importable by name) and
construction of an (oracle, HIP) solver pair in identical states."""
import importlib.util
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_pkg():
    if "calipso_jl_amd" in sys.modules:
        return sys.modules["calipso_jl_amd"]
    path = os.path.join(ROOT, "calipso.jl_amd")
    spec = importlib.util.spec_from_file_location("calipso_jl_amd", os.path.join(path, "__init__.py"), submodule_search_locations=[path])
    m = importlib.util.module_from_spec(spec)
    sys.modules["calipso_jl_amd"] = m
    spec.loader.exec_module(m)
    return m


def interior_point(prob, seed, tail=0.3):
    """tail: scale of the second-order cone tails (head = 1 + |tail|): wide cones need a smaller one to stay as far inside the cone, relatively,
    as the small ones (the upper-triangle symmetrisation of the reference, quirk B-3, is the coarser the closer to the boundary)"""
    rng = np.random.default_rng(seed)
    pt = dict(x=rng.standard_normal(prob.nx), r=rng.random(prob.ne), s=0.5 + rng.random(prob.nc), y=rng.standard_normal(prob.ne),
              z=rng.standard_normal(prob.nc), t=0.5 + rng.random(prob.nc))
    for c in prob.second_order_indices:
        if c:
            i = np.array(c) - 1
            pt["s"][i[1:]] = tail * rng.standard_normal(len(i) - 1)
            pt["t"][i[1:]] = tail * rng.standard_normal(len(i) - 1)
            pt["s"][i[0]] = 1.0 + np.linalg.norm(pt["s"][i[1:]])
            pt["t"][i[0]] = 1.0 + np.linalg.norm(pt["t"][i[1:]])
    lam = rng.standard_normal(prob.ne)
    return pt, lam


def make_pair(oracle_mod, prob, pt, lam, kappa=0.17, rho=52.0, ep=0.12, ed=0.21, tau=0.99):
    """oracle solver and HIP solver holding the same problem data, iterate and scalars"""
    import problems as pr
    pkg = load_pkg()
    o = oracle_mod.OracleSolver(prob.nx, prob.np, prob.ne, prob.nc, prob.nonnegative_indices, prob.second_order_indices)
    g = pkg.Solver(prob, prob.nx, prob.np, prob.ne, prob.nc, parameters=prob.parameters,
                   nonnegative_indices=prob.nonnegative_indices, second_order_indices=prob.second_order_indices)
    op = o.point()
    for k in "xrsyzt":
        op[k][:] = pt[k]
    w = op["all"].copy()
    o.buf("dual")[:] = lam
    for name, v in (("central_path", kappa), ("penalty", rho), ("primal_regularization", ep), ("dual_regularization", ed), ("fraction_to_boundary", tau)):
        o.buf(name)[0] = v
        g.set(name, [v])
    g.set("solution", w)
    if prob.ne:
        g.set("dual", lam)
    prob.evaluate(pr.ALL_VARIABLE_FLAGS, op["x"], op["y"], op["z"], prob.parameters, o.buf)
    g.evaluate(pr.ALL_VARIABLE_FLAGS, 0)
    return o, g


def near_boundary_point(prob, seed):
    """interior point whose second-order cone slacks / duals sit close to the cone boundary with unrelated directions: the
    upper-triangle symmetrisation of the condensed SOC blocks is then so coarse that iterative refinement diverges and
    search_direction! takes its `H \\ residual` fallback (search_direction.jl:22)"""
    rng = np.random.default_rng(seed)
    pt, lam = interior_point(prob, seed)
    for c in prob.second_order_indices:
        if c:
            i = np.array(c) - 1
            u = rng.standard_normal(len(i) - 1); v = rng.standard_normal(len(i) - 1)
            pt["s"][i[1:]] = u; pt["s"][i[0]] = (1 + 10 ** rng.uniform(-6, -2)) * np.linalg.norm(u)
            pt["t"][i[1:]] = v; pt["t"][i[0]] = (1 + 10 ** rng.uniform(-6, -2)) * np.linalg.norm(v)
    return pt, lam
