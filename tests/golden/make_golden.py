#!/usr/bin/env python3
"""Generates the committed golden fixtures (tests/golden/*.npz).

The reference (CALIPSO.jl) is pure Julia and cannot run in the build container, and it ships no golden-vector files, so
these fixtures are produced by the repo's own CPU oracle (oracle/), whose arithmetic is pinned against the closed-form
identities and known answers of the reference's tests (tests/test_oracle_*.py).  They freeze the oracle's outputs so that
(a) a later change of the oracle is detected and (b) the HIP path is compared with stored vectors on the GPU box.
    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import oracle  # noqa: E402
import problems as pr  # noqa: E402
from helpers import interior_point  # noqa: E402


def kat(name, prob, seed):
    """one Newton step at the constants of test/solver/problem.jl:56-65 (kappa=.17, rho=52, eps_p=.12, eps_d=.21)"""
    pt, lam = interior_point(prob, seed)
    o = oracle.OracleSolver(prob.nx, 0, prob.ne, prob.nc, prob.nonnegative_indices, prob.second_order_indices)
    op = o.point()
    for k in "xrsyzt":
        op[k][:] = pt[k]
    o.buf("dual")[:] = lam
    for nm, v in (("central_path", 0.17), ("penalty", 52.0), ("primal_regularization", 0.12), ("dual_regularization", 0.21), ("fraction_to_boundary", 0.99)):
        o.buf(nm)[0] = v
    prob.evaluate(pr.ALL_VARIABLE_FLAGS, op["x"], op["y"], op["z"], prob.parameters, o.buf)
    o.cone(barrier=True, barrier_gradient=True, product=True, jacobian=True, target=True)
    o.residual_jacobian_variables(); o.residual_jacobian_variables_symmetric(); o.residual(); o.residual_symmetric(0)
    out = dict(P=prob.P, q=prob.q, A=prob.A, b=prob.b, G=prob.G, h=prob.h, w=op["all"].copy(), lam=lam,
               nonneg=np.array(prob.nonnegative_indices, dtype=np.int64),
               soc=np.array([i for c in prob.second_order_indices for i in c], dtype=np.int64),
               soc_ptr=np.cumsum([0] + [len(c) for c in prob.second_order_indices]).astype(np.int64),
               cone_product=o.buf("cone_product").copy(), cone_target=o.buf("cone_target").copy(), barrier=o.buf("barrier").copy(),
               barrier_gradient=o.buf("barrier_gradient")[:prob.nc].copy(), residual=o.buf("residual").copy(),
               H=o.H_dense(), K=o.K_dense().copy(), residual_symmetric=o.buf("residual_symmetric").copy())
    o.factorize(update=False)
    out["inertia"] = np.array(o.compute_inertia(), dtype=np.int64)
    o.search_direction_symmetric(0, fact=False)
    out["step_first"] = o.buf("step").copy()
    assert o.iterative_refinement()
    out["step"] = o.buf("step").copy()
    s, t = op["s"], op["t"]
    st = out["step"]
    alphas = []
    for vec, dv in ((s, st[o.index("cone_slack") - 1]), (t, st[o.index("cone_slack_dual") - 1])):
        a = 1.0
        while prob.nc and o.cone_violation(vec - a * dv, vec, 0.99):
            a *= 0.5
        alphas.append(a)
    out["alpha"] = np.array(alphas)
    out["merit"] = np.array([o.merit(o.buf("objective")[0], op["r"], o.buf("barrier")[0])])
    out["theta"] = np.array([o.constraint_violation(o.buf("equality_constraint"), op["r"], o.buf("cone_constraint"), op["s"])])
    o.merit_gradient()
    out["merit_gradient"] = o.buf("merit_gradient").copy()
    out["optimality_error"] = np.array([o.optimality_error()])
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    # the same inputs as plain text for bench/ref_fixtures.jl (Julia, the reference itself): "name rows cols" + column-major values
    with open(os.path.join(HERE, name + "_inputs.txt"), "w") as fh:
        def rec(key, arr):
            a = np.atleast_2d(np.asarray(arr, dtype=np.float64))
            if a.shape[0] == 1 and np.asarray(arr).ndim <= 1:
                a = a.T
            fh.write("%s %d %d\n" % (key, a.shape[0], a.shape[1]))
            for v in a.T.reshape(-1):
                fh.write(repr(float(v)) + "\n")
        for key in ("P", "q", "A", "b", "G", "h", "w"):
            rec(key, out[key])
        rec("dual", lam)
        rec("objective_scale", [prob.c])
        rec("nonnegative_indices", prob.nonnegative_indices)
        rec("second_order_ptr", out["soc_ptr"])
        rec("second_order_indices", out["soc"])
        for key, v in (("central_path", 0.17), ("penalty", 52.0), ("primal_regularization", 0.12), ("dual_regularization", 0.21), ("fraction_to_boundary", 0.99)):
            rec(key, [v])
    print(name, "inertia", out["inertia"], "alpha", out["alpha"])


def write_inputs(name, prob, w, lam, scalars):
    """the inputs of a known-answer case as plain text for bench/ref_fixtures.jl"""
    with open(os.path.join(HERE, name + "_inputs.txt"), "w") as fh:
        def rec(key, arr):
            a = np.atleast_2d(np.asarray(arr, dtype=np.float64))
            if a.shape[0] == 1 and np.asarray(arr).ndim <= 1:
                a = a.T
            fh.write("%s %d %d\n" % (key, a.shape[0], a.shape[1]))
            for v in a.T.reshape(-1):
                fh.write(repr(float(v)) + "\n")
        for key, arr in (("P", prob.P), ("q", prob.q), ("A", prob.A), ("b", prob.b), ("G", prob.G), ("h", prob.h), ("w", w)):
            rec(key, arr)
        rec("dual", lam)
        rec("objective_scale", [prob.c])
        rec("nonnegative_indices", prob.nonnegative_indices)
        rec("second_order_ptr", np.cumsum([0] + [len(c) for c in prob.second_order_indices]))
        rec("second_order_indices", [i for c in prob.second_order_indices for i in c])
        for key, v in scalars:
            rec(key, [v])


def kat_search_direction(name, prob, seed, kappa, rho):
    """search_direction! as a whole (search_direction.jl:1-23) from the DEFAULT regularisation start (options.jl): inertia_correction! walks its sequence IC-1 .. IC-6
    (inertia.jl:30-80, quirk B-1 included), then the condensed solve and the refinement.  What the closed forms of test/solver/problem.jl do not hold: the sequence, the
    final regularisation, the condensed second-order blocks at work."""
    pt, lam = interior_point(prob, seed)
    o = oracle.OracleSolver(prob.nx, 0, prob.ne, prob.nc, prob.nonnegative_indices, prob.second_order_indices)
    op = o.point()
    for k in "xrsyzt":
        op[k][:] = pt[k]
    o.buf("dual")[:] = lam
    scalars = (("central_path", kappa), ("penalty", rho), ("primal_regularization", 0.0), ("dual_regularization", 0.0), ("fraction_to_boundary", 0.99))
    for nm, v in scalars:
        o.buf(nm)[0] = v
    prob.evaluate(pr.ALL_VARIABLE_FLAGS, op["x"], op["y"], op["z"], prob.parameters, o.buf)
    o.cone(product=True, jacobian=True, target=True)
    o.residual()
    rc = o.search_direction()
    assert rc in (0, 2), rc
    st = o.stats()
    out = dict(P=prob.P, q=prob.q, A=prob.A, b=prob.b, G=prob.G, h=prob.h, w=op["all"].copy(), lam=lam,
               soc_ptr=np.cumsum([0] + [len(c) for c in prob.second_order_indices]).astype(np.int64),
               status=np.array([rc]), factorizations=np.array([st["factorizations"]]), inertia=np.array(o.compute_inertia(), dtype=np.int64),
               primal_regularization=o.buf("primal_regularization").copy(), primal_regularization_last=o.buf("primal_regularization_last").copy(),
               dual_regularization=o.buf("dual_regularization").copy(), residual=o.buf("residual").copy(), step=o.buf("step").copy())
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    write_inputs(name, prob, out["w"], lam, scalars)
    print(name, "status", rc, "factorisations", st["factorizations"], "eps_p", out["primal_regularization"][0], "inertia", out["inertia"])


def trace(name, prob, **opts):
    """full solve! iterate trace (BASELINE configs C1 / C2)"""
    o = oracle.OracleSolver(prob.nx, prob.np, prob.ne, prob.nc, prob.nonnegative_indices, prob.second_order_indices)
    o.point()["x"][:] = prob.x0
    status = o.solve(prob)
    tr = o.trace()
    st = o.stats()
    np.savez_compressed(os.path.join(HERE, name + ".npz"), x0=prob.x0, trace=tr, status=np.array([status]),
                        total_iterations=np.array([st["total_iterations"]]), solution=o.point()["all"].copy(),
                        residual=o.buf("residual").copy())
    print(name, "status", status, st["total_iterations"], "iterations, trace", tr.shape)


if __name__ == "__main__":
    kat("kat_qp_10_5_5", pr.random_qp(10, 5, 5, seed=3), 1)
    kat("kat_soc_6_3_9", pr.random_qp(6, 3, 9, seed=10, nonnegative_indices=[1, 2], second_order_indices=[[3, 4, 5], [6, 7, 8, 9]]), 1)
    nonconvex = pr.random_qp(12, 3, 4, seed=4)
    nonconvex.P = -nonconvex.P
    nonconvex.Psym = nonconvex.c * (nonconvex.P + nonconvex.P.T)
    kat_search_direction("kat_sd_nonconvex_12_3_4", nonconvex, 2, 1.0, 1.0)        # IC-1 fails: the regularisation sequence of inertia.jl:30-80
    kat_search_direction("kat_sd_portfolio_soc12", pr.portfolio(seed=0, p=10), 1, 0.17, 52.0)   # one second-order cone of dimension 12 (test/solver/portfolio.jl:33-62)
    trace("c1_wachter_trace", pr.wachter())
    trace("c2_pendulum_trace", pr.pendulum(action_guess=np.zeros(10)))
