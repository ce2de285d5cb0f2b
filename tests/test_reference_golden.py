"""Reference-generated parity: golden vectors written by NonlinearSolve.jl itself (tests/golden/make_reference_golden.jl, to be
run wherever Julia and the reference are installed — not in the build container) against the oracle (CPU) and the device (GPU).
Skipped while tests/golden/reference/ holds no files; the moment they are committed, these tests pin BOTH the checker and the
product on the reference's own output: iterates, residual histories, NLStats, return codes, the precs call protocol.

Tolerances: final u within 1e-8·max(1, ‖u‖∞) where the linear solves are exact (direct linsolve), 1e-6 where
both sides stop Krylov solves at forwarded / forcing tolerances; ‖f‖∞ histories rtol 1e-6 (direct) / 1e-2 (Krylov, plain
tolerances); Newton step counts equal (direct, plain Krylov) or within ±1 (Eisenstat–Walker); return codes equal."""
import glob
import json
import os

import numpy as np
import pytest

from oracle import reference_restatement as R

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference")
FILES = sorted(f for f in glob.glob(os.path.join(GOLD, "*.json")) if not f.endswith(".FAILED.json"))
needs_golden = pytest.mark.skipif(not FILES, reason="no reference-generated golden files: run tests/golden/make_reference_golden.jl "
                                  "where Julia + NonlinearSolve.jl are installed and commit tests/golden/reference/*.json")


def _load(name):
    path = os.path.join(GOLD, name + ".json")
    if not os.path.exists(path):
        pytest.skip(f"{name}.json not generated")
    d = json.load(open(path))
    for k in ("u", "resid", "fnorm_inf", "fnorm_2"):
        if k in d:
            d[k] = np.array([float(x) for x in d[k]], dtype=np.float64)
    return d


# name → (problem factory, algorithm factory(mod), solve kwargs, kind): one table for the oracle (mod = R) and the device (mod = nls)
def _cases(mod, device=False):
    def bratu(n):
        return (lambda: mod.Bratu2D(n, 6.0))

    def krylov(**kw):
        # KrylovJL_GMRES(gmres_restart = 30): restarted GMRES(30), itmax = length(b) [EXT LinearSolve → Krylov.jl]
        return mod.KrylovJL_GMRES(gmres_restart=30, maxiters=64 * 64, **({"ortho": "mgs"} if not device else {}), **kw)
    c = {
        "c1_quadratic1000_newton": (lambda: mod.Quadratic(1000, 2.0), lambda: mod.NewtonRaphson(), {}, "direct"),
        "c1_quadratic1000_trustregion": (lambda: mod.Quadratic(1000, 2.0), lambda: mod.TrustRegion(), {}, "direct"),
        "c2_bratu256_newton_direct": (bratu(256), lambda: mod.NewtonRaphson(), dict(abstol=1e-8, maxiters=50), "direct"),
        "bratu64_newton_direct": (bratu(64), lambda: mod.NewtonRaphson(), dict(abstol=1e-8, maxiters=50), "direct"),
        "bratu64_trustregion_direct": (bratu(64), lambda: mod.TrustRegion(), dict(abstol=1e-8, maxiters=50), "direct"),
        "bratu64_newton_gmres30_matfree": (bratu(64), lambda: mod.NewtonRaphson(linsolve=krylov()), dict(abstol=1e-8, maxiters=50), "krylov"),
        "bratu64_newton_gmres30_concrete": (bratu(64), lambda: mod.NewtonRaphson(linsolve=krylov(), concrete_jac=True),
                                            dict(abstol=1e-8, maxiters=50), "krylov"),
        "bratu64_newton_gmres30_ew_matfree": (bratu(64), lambda: mod.NewtonRaphson(linsolve=krylov(), forcing=mod.EisenstatWalkerForcing2()),
                                              dict(abstol=1e-8, maxiters=50), "ew"),
        "bratu64_newton_gmres30_ew_concrete": (bratu(64), lambda: mod.NewtonRaphson(linsolve=krylov(), forcing=mod.EisenstatWalkerForcing2(),
                                                                                    concrete_jac=True), dict(abstol=1e-8, maxiters=50), "ew"),
        "bratu64_trustregion_gmres30_matfree": (bratu(64), lambda: mod.TrustRegion(linsolve=krylov()), dict(abstol=1e-8, maxiters=50), "krylov"),
        "brusselator32_newton_dense_ad": (lambda: mod.Brusselator2D(32), lambda: mod.NewtonRaphson(), dict(abstol=1e-8), "direct"),
        "brusselator32_newton_sparse_ad": (lambda: mod.Brusselator2D(32), lambda: mod.NewtonRaphson(), dict(abstol=1e-8), "direct"),
        "brusselator32_trustregion_dense_ad": (lambda: mod.Brusselator2D(32), lambda: mod.TrustRegion(), dict(abstol=1e-8), "direct"),
    }
    return c


def _compare(gold, u, retcode, nsteps, fnorm_trace, kind):
    utol = {"direct": 1e-8, "krylov": 1e-6, "ew": 1e-6}[kind]
    assert retcode == gold["retcode"], (retcode, gold["retcode"])
    assert np.max(np.abs(u - gold["u"])) <= utol * max(1.0, np.max(np.abs(gold["u"])))
    if kind == "ew":
        assert abs(nsteps - gold["nsteps"]) <= 1
    else:
        assert nsteps == gold["nsteps"]
        ftol = 1e-6 if kind == "direct" else 1e-2
        g = gold["fnorm_inf"][1:]
        m = min(len(g), len(fnorm_trace))
        big = g[:m] > 1e3 * (gold.get("abstol", 1e-8))      # (the last entries sit at the tolerance: rounding decides their digits)
        assert np.allclose(np.asarray(fnorm_trace[:m])[big], g[:m][big], rtol=ftol)


@needs_golden
@pytest.mark.parametrize("name", sorted(_cases(R)))
def test_oracle_reproduces_the_reference(name):
    gold = _load(name)
    mk_prob, mk_alg, kw, kind = _cases(R)[name]
    sol = R.solve(mk_prob(), mk_alg(), **kw)
    _compare(gold, sol.u, R.RETCODE_NAMES[sol.retcode], sol.stats.nsteps, [t["fnorm_inf"] for t in sol.trace], kind)
    if kind == "direct":
        assert (sol.stats.nf, sol.stats.njacs, sol.stats.nsolve) == (gold["nf"], gold["njacs"], gold["nsolve"])


@needs_golden
@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(_cases(R)))
def test_device_reproduces_the_reference(nls, name):
    gold = _load(name)
    mk_prob, mk_alg, kw, kind = _cases(nls, device=True)[name]
    sol = nls.solve(nls.NonlinearProblem(mk_prob()), mk_alg(), store_trace=True, **kw)
    _compare(gold, np.asarray(sol.u.cpu()), sol.retcode, sol.stats.nsteps, [t["fnorm_inf"] for t in sol.trace], kind)
    if kind == "direct":
        assert (sol.stats.nf, sol.stats.njacs, sol.stats.nsolve) == (gold["nf"], gold["njacs"], gold["nsolve"])


@needs_golden
def test_tridiagonal_known_answer_from_the_reference():
    xref = _load("tridiagonal40_xref")["u"]
    for name in ("tridiagonal40_matrixoperator_gmres", "tridiagonal40_sparse_gmres"):
        if os.path.exists(os.path.join(GOLD, name + ".json")):
            g = _load(name)
            assert g["retcode"] == "Success" and np.allclose(g["u"], xref)
    import scipy.sparse as sp
    W = sp.diags([-np.ones(39), 4.0 * np.ones(40), -np.ones(39)], [-1, 0, 1]).tocsr()
    assert np.allclose(np.linalg.solve(W.toarray(), np.arange(1.0, 41.0)), xref)


@needs_golden
def test_precs_call_protocol_matches_the_reference():
    """the counts the reference itself produced for test/Core/core_tests__item21.jl's sequence against the oracle's restatement"""
    g = _load("precs_protocol_counts")

    class Cubic:
        n, p = 2, 0
        def u0(self): return np.zeros(2)
        def f(self, u): return -(u - 0.1) ** 3
        def jvp(self, v, u): return -3.0 * (u - 0.1) ** 2 * v
        def vjp(self, v, u): return -3.0 * (u - 0.1) ** 2 * v
    calls = []
    def precs(W, p=None):
        calls.append(float(p.p))
        return None, None
    it = R.init(Cubic(), R.NewtonRaphson(linsolve=R.KrylovJL_GMRES(precs=precs)))
    n0 = len(calls); s1 = it.solve(); n1 = len(calls)
    it.reinit(np.zeros(2), p=1); n2 = len(calls); it.solve(); n3 = len(calls)
    it.reinit(p=2); n4 = len(calls); it.solve(); n5 = len(calls)
    assert (n1 - n0, n2 - n1, n3 - n2, n4 - n3, n5 - n4) == (g["calls_first_solve"], g["calls_by_reinit_u0"], g["calls_second_solve"],
                                                             g["calls_by_reinit_p"], g["calls_third_solve"])
    assert n0 == g["calls_at_init"] and s1.stats.nsteps == g["nsteps_first"]
    assert calls == [float(x) for x in g["p_seen"]]


def test_the_generator_is_committed_and_covers_the_path():
    """(always runs) the Julia generator exists, names every configuration this file consumes, and asks for nothing outside the
    reference's own API — so that the only missing ingredient for reference-generated parity is an environment with Julia."""
    src = open(os.path.join(os.path.dirname(GOLD), "make_reference_golden.jl")).read()
    for name in list(_cases(R)) + ["tridiagonal40_xref", "precs_protocol_counts"]:
        assert name.replace("_matfree", "").replace("_concrete", "").split("_ew")[0] in src or name in src, name
    for api in ("NewtonRaphson(", "TrustRegion(", "KrylovJL_GMRES(", "EisenstatWalkerForcing2()", "init(", "step!(", "solve!(", "reinit!("):
        assert api in src
