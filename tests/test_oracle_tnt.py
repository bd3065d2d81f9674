"""oracle/tnt.py against the REFERENCE's own TNT / STPCG.

tests/golden/tnt_ref.jsonl is the output of oracle/_ref/tnt_ref, the
reference's header-only solver (C++/Optimization/include/Optimization/
Riemannian/TNT.h, LinearAlgebra/IterativeSolvers.h) compiled by
oracle/ref_tnt/Makefile around oracle/ref_tnt/harness.cpp.  The first cases
are the known-answer fixtures of the reference's unit tests
(IterativeSolvers_unit_test.cpp:79-247, TNT_unit_test.cpp:63-187).
"""
import json
import math
import os

import numpy as np
import pytest

from oracle import tnt as T

STATUS = ["Gradient", "PreconditionedGradient", "RelativeDecrease", "Stepsize",
          "TrustRegion", "IterationLimit", "ElapsedTime", "UserFunction"]


def _cases(golden_dir, kind):
    with open(os.path.join(golden_dir, "tnt_ref.jsonl")) as fh:
        return [c for c in map(json.loads, fh) if c["kind"] == kind]


def test_stpcg_matches_reference(golden_dir):
    cases = _cases(golden_dir, "stpcg")
    assert len(cases) >= 11
    for c in cases:
        g, Hd = np.array(c["g"]), np.array(c["Hdiag"])
        P = None
        if "Mdiag" in c:
            Mi = 1.0 / np.array(c["Mdiag"])
            P = lambda v, Mi=Mi: Mi * v
        s, nrm, nit = T.stpcg(g, lambda v: Hd * v, lambda a, b: float(a @ b), c["Delta"],
                              c["max_it"], c["kappa"], c["theta"], P)
        assert nit == c["num_iterations"], c["case"]
        np.testing.assert_allclose(s, c["s"], rtol=1e-12, atol=1e-13, err_msg=c["case"])
        assert math.isclose(nrm, c["step_norm"], rel_tol=1e-12), c["case"]


def test_stpcg_unit_test_known_answers(golden_dir):
    """The assertions of IterativeSolvers_unit_test.cpp:138-247 themselves."""
    g, Pd, Md = np.array([21, -.4, 19]), np.array([1000., 100., 1.]), np.array([100., 10., 1.])
    ip = lambda a, b: float(a @ b)
    big = np.finfo(np.float64).max
    s, nrm, _ = T.stpcg(g, lambda v: Pd * v, ip, big, 3, 1e-8, .999)
    assert np.linalg.norm(s + g / Pd) < 1e-6 and abs(nrm - np.linalg.norm(s)) / np.linalg.norm(s) < 1e-6
    s, nrm, _ = T.stpcg(g, lambda v: -Pd * v, ip, 1000, 3, 1e-8, .999)
    assert np.linalg.norm(s + 1000 / np.linalg.norm(g) * g) < 1e-6
    s, nrm, _ = T.stpcg(g, lambda v: Pd * v, ip, big, 3, 1e-8, .999, lambda v: v / Md)
    assert np.linalg.norm(s + g / Pd) < 1e-6
    assert abs(nrm - math.sqrt(s @ (Md * s))) / nrm < 1e-6
    s, nrm, _ = T.stpcg(g, lambda v: -Pd * v, ip, 1000, 3, 1e-8, .999, lambda v: v / Md)
    p = -g / Md
    assert np.linalg.norm(s - 1000 / math.sqrt(p @ (Md * p)) * p) < 1e-6


def _sphere_problem(use_precon):
    Pt = np.array([0., 0., 1.])
    proj = lambda X, V: V - (X @ V) * X
    gradF = lambda X: proj(X, 2 * (X - Pt))
    f = lambda X: float((X - Pt) @ (X - Pt))

    def QM(X):
        return gradF(X), (lambda Xc, Xd: proj(Xc, 2 * Xd) - (Xc @ gradF(Xc)) * Xd)
    metric = lambda X, a, b: float(a @ b)
    retract = lambda X, V: (X + V) / np.linalg.norm(X + V)
    precon = (lambda X, V: np.array([1., 2., 3.]) * V) if use_precon else None
    return f, QM, metric, retract, precon, gradF


def test_tnt_matches_reference(golden_dir):
    cases = _cases(golden_dir, "tnt")
    assert len(cases) >= 5
    for c in cases:
        f, QM, metric, retract, precon, gradF = _sphere_problem(bool(c["precon"]))
        p = T.TNTParams()
        p.max_iterations, p.max_iterations_accepted = c["max_it"], c["max_acc"]
        p.gradient_tolerance, p.preconditioned_gradient_tolerance = c["gtol"], c["pgtol"]
        p.relative_decrease_tolerance, p.stepsize_tolerance = c["reltol"], c["steptol"]
        p.kappa_fgr, p.theta, p.max_TPCG_iterations = c["kappa"], c["theta"], 10000
        log = []
        r = T.tnt(f, QM, metric, retract, np.array(c["x0"]), precon, p, log)
        assert r["status"] == STATUS[c["status"]], c["case"]
        np.testing.assert_allclose(r["x"], c["x"], rtol=1e-9, atol=1e-12, err_msg=c["case"])
        assert math.isclose(r["f"], c["f"], rel_tol=1e-9, abs_tol=1e-20), c["case"]
        assert [l["inner"] for l in log] == [int(v) for v in c["inner_iterations"]], c["case"]
        np.testing.assert_allclose([l["Delta"] for l in log], c["trust_region_radius"][:len(log)],
                                   rtol=1e-9, err_msg=c["case"])
        np.testing.assert_allclose([l["rho"] for l in log], c["gain_ratios"], rtol=1e-6, err_msg=c["case"])


def test_tnt_unit_test_known_answers():
    """The assertions of TNT_unit_test.cpp:126-187."""
    for use_precon in (False, True):
        f, QM, metric, retract, precon, gradF = _sphere_problem(use_precon)
        p = T.TNTParams()
        p.relative_decrease_tolerance = p.stepsize_tolerance = 0
        p.preconditioned_gradient_tolerance = 0
        p.gradient_tolerance = 1e-8
        X0 = np.array([-0.5, -0.5, -0.707107])
        r = T.tnt(f, QM, metric, retract, X0, precon, p)
        assert r["status"] == "Gradient"
        assert np.linalg.norm(gradF(r["x"])) < 1e-8
        assert f(r["x"]) < f(X0)


def _pgo_cases(golden_dir):
    with open(os.path.join(golden_dir, "tnt_pgo_ref.jsonl")) as fh:
        return [json.loads(l) for l in fh if l.strip()]


def _capture(recipe):
    import importlib.util
    here = os.path.dirname(os.path.abspath(__file__))
    spec = importlib.util.spec_from_file_location("make_tnt_pgo_golden", os.path.join(here, "golden", "make_tnt_pgo_golden.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.capture(recipe["dataset"], recipe["num_nodes"], recipe["loss"], recipe["accelerated"], recipe["node"],
                       recipe["call"], recipe["overrides"])


def test_tnt_on_pgo_problems_matches_reference(golden_dir):
    """the reference's TNT (TNT.h:242-693, compiled where it lies) driven by DPGO-shaped operators on one
    node's surrogate of smallGrid3D / tinyGrid3D (tests/golden/tnt_pgo_ref.jsonl, made by make_tnt_pgo_golden.py with
    oracle/ref_tnt/harness_pgo.cpp's own dense restatement of the operators) against the oracle's TNT call of
    DPGOHash (oracle/hash.py::_tnt with the operators of oracle/problem.py): same objective values, radii, gain ratios,
    CG iteration counts, status and final point.  This pins the a11 pieces (reduced gradient, Hessian-vector product,
    preconditioner, retraction) and their wiring into TNT, not only the solver on toy problems."""
    cases = _pgo_cases(golden_dir)
    assert len(cases) >= 7
    for c in cases:
        _, nd, got = _capture(c["recipe"])
        res, log = got["res"], got["log"]
        assert got["outer"] == c["recipe"]["outer_iteration"], c["case"]
        assert res["status"] == STATUS[c["status"]], c["case"]
        assert math.isclose(res["f"], c["f"], rel_tol=1e-11), c["case"]
        assert [l["inner"] for l in log] == [int(v) for v in c["inner_iterations"]], c["case"]
        np.testing.assert_allclose([l["Delta"] for l in log], c["trust_region_radius"][:len(log)], rtol=1e-9, err_msg=c["case"])
        # (a gain ratio is a quotient of differences of nearly equal objective values: the last iterations of the tight
        # case have df / f ~ 1e-10, so it carries about five digits)
        np.testing.assert_allclose([l["rho"] for l in log], c["gain_ratios"], rtol=1e-4, atol=1e-9, err_msg=c["case"])
        np.testing.assert_allclose(res["x"], np.asarray(c["x"]).reshape(res["x"].shape), atol=1e-10, err_msg=c["case"])
