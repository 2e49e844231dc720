"""Reference-generated fixtures (tests/golden/ref_kat_*.txt, produced by bench/ref_fixtures.jl with the Julia reference itself).

The build container has no Julia, so these files may be absent: then only the transport format is tested (the text inputs the
Julia script reads hold exactly the bits of the .npz fixtures) and the parity status stays "unpinned".  Once a maintainer has run
the script where Julia exists and committed its outputs, the same tests pin the oracle — H, K, R, b bit-for-bit up to 1e-12,
inertia exactly, steps to 1e-8 (the elimination order of AMD.jl differs) — and through it the HIP path."""
import os

import numpy as np
import pytest

import problems as pr
from helpers import interior_point

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
NAMES = ["kat_qp_10_5_5", "kat_soc_6_3_9"]
SD_NAMES = ["kat_sd_nonconvex_12_3_4", "kat_sd_portfolio_soc12"]      # search_direction! as a whole: the IC sequence on a non-convex Hessian, a second-order cone of dimension 12


def read_records(path):
    toks = open(path).read().split()
    out, i = {}, 0
    while i < len(toks):
        name, r, c = toks[i], int(toks[i + 1]), int(toks[i + 2])
        vals = np.array([float(v) for v in toks[i + 3: i + 3 + r * c]])
        out[name] = vals.reshape(c, r).T
        i += 3 + r * c
    return out


@pytest.mark.parametrize("name", NAMES + SD_NAMES)
def test_text_inputs_hold_the_bits_of_the_npz_fixture(name):
    d = np.load(os.path.join(HERE, name + ".npz"))
    t = read_records(os.path.join(HERE, name + "_inputs.txt"))
    for key in ("P", "A", "G"):
        assert np.array_equal(t[key], d[key]), key
    for key in ("q", "b", "h", "w"):
        assert np.array_equal(t[key][:, 0], d[key]), key
    assert np.array_equal(t["dual"][:, 0], d["lam"])
    assert np.array_equal(t["second_order_ptr"][:, 0].astype(np.int64), d["soc_ptr"])
    if name in NAMES:
        assert t["central_path"][0, 0] == 0.17 and t["penalty"][0, 0] == 52.0


def rel(a, b):
    return np.abs(np.asarray(a) - np.asarray(b)).max() / max(1.0, np.abs(np.asarray(b)).max())


@pytest.mark.parametrize("name", NAMES)
def test_oracle_against_reference_generated_fixture(name):
    path = os.path.join(HERE, "ref_" + name + ".txt")
    if not os.path.exists(path):
        pytest.skip("no reference-generated fixture (bench/ref_fixtures.jl needs Julia + CALIPSO.jl): parity unpinned")
    ref = read_records(path)
    d = np.load(os.path.join(HERE, name + ".npz"))
    # tests/golden/*.npz is what the oracle produces (test_golden.py asserts that); compare it with the reference's output
    assert rel(d["H"], ref["H"]) <= 1e-12
    assert rel(d["K"], ref["K"]) <= 1e-12
    assert rel(d["residual"], ref["residual"][:, 0]) <= 1e-12
    assert rel(d["residual_symmetric"], ref["residual_symmetric"][:, 0]) <= 1e-12
    assert np.array_equal(d["inertia"], ref["inertia"][:, 0].astype(np.int64))
    assert rel(d["step_first"], ref["step_first"][:, 0]) <= 1e-8
    assert rel(d["step"], ref["step"][:, 0]) <= 1e-8
    assert np.array_equal(d["alpha"], ref["alpha"][:, 0])
    assert abs(d["merit"][0] - ref["merit"][0, 0]) <= 1e-12 * max(1.0, abs(ref["merit"][0, 0]))
    assert abs(d["theta"][0] - ref["theta"][0, 0]) <= 1e-12 * max(1.0, abs(ref["theta"][0, 0]))
    assert rel(d["merit_gradient"], ref["merit_gradient"][:, 0]) <= 1e-12
    assert abs(d["optimality_error"][0] - ref["optimality_error"][0, 0]) <= 1e-12 * max(1.0, ref["optimality_error"][0, 0])


@pytest.mark.parametrize("name", SD_NAMES)
def test_oracle_search_direction_against_reference_generated_fixture(name):
    """what no closed form of the reference's tests holds: the regularisation sequence of inertia_correction! and a Newton step through a wide second-order cone"""
    path = os.path.join(HERE, "ref_" + name + ".txt")
    if not os.path.exists(path):
        pytest.skip("no reference-generated fixture (bench/ref_fixtures.jl needs Julia + CALIPSO.jl): parity unpinned")
    ref = read_records(path)
    d = np.load(os.path.join(HERE, name + ".npz"))
    assert rel(d["residual"], ref["residual"][:, 0]) <= 1e-12
    assert np.array_equal(d["inertia"], ref["inertia"][:, 0].astype(np.int64))
    for key in ("primal_regularization", "primal_regularization_last", "dual_regularization"):
        assert d[key][0] == ref[key][0, 0], key                       # the same walk through IC-1 .. IC-6: the same floating-point products
    assert rel(d["step"], ref["step"][:, 0]) <= 1e-8


@pytest.mark.parametrize("name", SD_NAMES)
def test_oracle_reproduces_the_search_direction_fixture(name, oracle_mod):
    """the .npz fixture is what the oracle produces today (a later change of the oracle is detected); GPU: tests/test_golden.py"""
    d, o, rc = sd_case(name, oracle_mod)
    assert rc == d["status"][0] and np.array_equal(np.array(o.compute_inertia()), d["inertia"])
    assert o.buf("primal_regularization")[0] == d["primal_regularization"][0] and o.buf("dual_regularization")[0] == d["dual_regularization"][0]
    assert rel(o.buf("step"), d["step"]) <= 1e-12


def sd_problem(d):
    soc_flat = read_records(os.path.join(HERE, d + "_inputs.txt"))
    ptr = soc_flat["second_order_ptr"][:, 0].astype(int); flat = soc_flat["second_order_indices"][:, 0].astype(int) if soc_flat["second_order_indices"].size else np.zeros(0, int)
    soc = [list(flat[ptr[k]:ptr[k + 1]]) for k in range(len(ptr) - 1)]
    nn = [int(v) for v in soc_flat["nonnegative_indices"][:, 0]] if soc_flat["nonnegative_indices"].size else []
    t = soc_flat
    prob = pr.ConicQP(t["P"], t["q"][:, 0], t["A"], t["b"][:, 0] if t["b"].size else np.zeros(0), t["G"], t["h"][:, 0] if t["h"].size else np.zeros(0),
                      nonnegative_indices=nn, second_order_indices=soc, objective_scale=float(t["objective_scale"][0, 0]))
    return prob, t


def sd_case(name, oracle_mod):
    d = np.load(os.path.join(HERE, name + ".npz"))
    prob, t = sd_problem(name)
    o = oracle_mod.OracleSolver(prob.nx, 0, prob.ne, prob.nc, prob.nonnegative_indices, prob.second_order_indices)
    op = o.point()
    op["all"][:] = d["w"]
    o.buf("dual")[:] = d["lam"]
    for nm in ("central_path", "penalty", "primal_regularization", "dual_regularization", "fraction_to_boundary"):
        o.buf(nm)[0] = t[nm][0, 0]
    prob.evaluate(pr.ALL_VARIABLE_FLAGS, op["x"], op["y"], op["z"], prob.parameters, o.buf)
    o.cone(product=True, jacobian=True, target=True)
    o.residual()
    return d, o, o.search_direction()
