"""GPU parity tests of GMRES / NewtonRaphson / TrustRegion against the oracle (through the C ABI).

Stated tolerances:
  GMRES vs oracle GMRES at equal (m, rtol): ‖x−x_ref‖₂ ≤ 10·rtol·‖x_ref‖₂
  Newton/TR final u vs oracle:              ‖u−u_ref‖∞ ≤ 1e-8·max(1, ‖u_ref‖∞), both at ‖F‖∞ ≤ abstol
  Newton step counts equal ±1 under the identical protocol."""
import os

import numpy as np
import pytest

from oracle import reference_restatement as R

pytestmark = pytest.mark.gpu


def uerr(u, uref):
    return float(np.max(np.abs(np.asarray(u) - uref)) / max(1.0, np.max(np.abs(uref))))


@pytest.mark.parametrize("ortho", ["mgs", "cgs2", "cgs", "dcgs2"])
@pytest.mark.parametrize("ns,rtol", [(16, 1e-10), (48, 1e-6)])
def test_gmres_csr_vs_oracle(nls, ns, rtol, ortho):
    p = R.Bratu2D(ns)
    u = 0.2 * np.random.default_rng(0).standard_normal(p.n)
    J = p.jac(u)
    b = np.random.default_rng(1).standard_normal(p.n)
    xref, iref = R.gmres(lambda z: J @ z, b, rtol=rtol, restart=30, itmax=5000)
    G = nls.GMRES(p.n, restart=30, ortho=ortho).set_operator(nls.CSRMatrix.from_scipy(J))
    x, info = G.solve(b, abstol=0.0, reltol=rtol, maxiters=5000)
    assert info["converged"] and not info["failed"]
    assert np.linalg.norm(x - xref) <= 10 * rtol * np.linalg.norm(xref)
    assert np.linalg.norm(J @ x - b) <= 1.01 * rtol * np.linalg.norm(b) + 1e-12
    assert abs(info["iters"] - iref.iters) <= max(3, iref.iters // 20)
    assert abs(info["rnorm0"] - np.linalg.norm(b)) <= 1e-12 * np.linalg.norm(b)


def test_gmres_first_cycle_matches_oracle_mgs(nls):
    """With MGS the device Arnoldi process follows the oracle's arithmetic: recurrence residual after a
    fixed number of steps agrees to rounding."""
    p = R.Bratu2D(24)
    J = p.jac(np.zeros(p.n))
    b = np.random.default_rng(2).standard_normal(p.n)
    for k in (1, 5, 30, 45):
        xref, iref = R.gmres(lambda z: J @ z, b, restart=30, fixed_iters=k)
        G = nls.GMRES(p.n, restart=30, ortho="mgs").set_operator(nls.CSRMatrix.from_scipy(J))
        x, info = G.solve(b, fixed_iters=k)
        assert info["iters"] == k == iref.iters
        assert abs(info["rnorm"] - iref.rnorm) <= 1e-10 * iref.rnorm0
        assert np.linalg.norm(x - xref) <= 1e-9 * np.linalg.norm(xref)


def test_gmres_dcgs2_follows_the_oracle_restatement(nls):
    """CGS2 with delayed re-orthogonalisation (2 sweeps over the basis per step): recurrence residuals and iterates agree
    with the oracle's restatement of the same scheme — and hence with CGS2 — to rounding, across restarts, with the
    Chebyshev right preconditioner, and at early termination inside a cycle."""
    p = R.Bratu2D(24)
    J = p.jac(0.1 * np.random.default_rng(5).standard_normal(p.n))
    b = np.random.default_rng(2).standard_normal(p.n)
    A = nls.CSRMatrix.from_scipy(J)
    for k in (1, 2, 5, 30, 31, 45, 90):
        xref, iref = R.gmres(lambda z: J @ z, b, restart=30, fixed_iters=k, ortho="dcgs2")
        xc, ic = R.gmres(lambda z: J @ z, b, restart=30, fixed_iters=k, ortho="cgs2")
        x, info = nls.GMRES(p.n, restart=30, ortho="dcgs2").set_operator(A).solve(b, fixed_iters=k)
        assert info["iters"] == k == iref.iters
        assert abs(info["rnorm"] - iref.rnorm) <= 1e-10 * iref.rnorm0 and abs(iref.rnorm - ic.rnorm) <= 1e-10 * ic.rnorm0
        assert np.linalg.norm(x - xref) <= 1e-9 * np.linalg.norm(xref)
    G = nls.GMRES(p.n, restart=30, ortho="dcgs2").set_operator(A)
    G.set_chebyshev_preconditioner(8, ratio=30.0)
    x, info = G.solve(b, abstol=0.0, reltol=1e-10, maxiters=2000)
    G2 = nls.GMRES(p.n, restart=30, ortho="cgs2").set_operator(A)
    G2.set_chebyshev_preconditioner(8, ratio=30.0)
    x2, info2 = G2.solve(b, abstol=0.0, reltol=1e-10, maxiters=2000)
    assert info["converged"] and info["iters"] == info2["iters"]
    assert np.linalg.norm(J @ x - b) <= 1.01e-10 * np.linalg.norm(b) and np.linalg.norm(x - x2) <= 1e-9 * np.linalg.norm(x2)


def test_gmres_dcgs2_one_reduction_follows_the_oracle(nls):
    """DCGS2 with a single reduction per Arnoldi step (`ortho="dcgs2_1r"`, what several ranks use): iterates, recurrence
    residuals and iteration counts equal the oracle's restatement of the scheme and plain CGS2 to rounding — fixed
    numbers of steps across restarts, tolerance stops inside a cycle, matrix-free operator, Chebyshev preconditioner."""
    p = R.Bratu2D(24)
    u = 0.1 * np.random.default_rng(5).standard_normal(p.n)
    J = p.jac(u)
    b = np.random.default_rng(2).standard_normal(p.n)
    A = nls.CSRMatrix.from_scipy(J)
    for k in (1, 2, 5, 30, 31, 45, 90):
        xref, iref = R.gmres_dcgs2_1r(lambda z: J @ z, b, restart=30, fixed_iters=k)
        x, info = nls.GMRES(p.n, restart=30, ortho="dcgs2_1r").set_operator(A).solve(b, fixed_iters=k)
        assert info["iters"] == k == iref.iters
        assert abs(info["rnorm"] - iref.rnorm) <= 1e-10 * iref.rnorm0
        assert np.linalg.norm(x - xref) <= 1e-9 * np.linalg.norm(xref)
    for rtol in (1e-4, 1e-10):
        xref, iref = R.gmres_dcgs2_1r(lambda z: J @ z, b, rtol=rtol, restart=30, itmax=3000)
        xc, ic = R.gmres(lambda z: J @ z, b, rtol=rtol, restart=30, itmax=3000, ortho="cgs2")
        x, info = nls.GMRES(p.n, restart=30, ortho="dcgs2_1r").set_operator(A).solve(b, abstol=0.0, reltol=rtol, maxiters=3000)
        assert info["converged"] and info["iters"] == iref.iters == ic.iters
        assert np.linalg.norm(x - xc) <= 1e-9 * np.linalg.norm(xc)
    prob = nls.NonlinearProblem(nls.Bratu2D(24))
    op = nls.StatefulJacobianOperator(nls.JacobianOperator(prob), u)
    G = nls.GMRES(p.n, restart=30, ortho="dcgs2_1r").set_operator(op)
    G.set_chebyshev_preconditioner(8, ratio=30.0)
    x, info = G.solve(b, abstol=0.0, reltol=1e-10, maxiters=2000)
    G2 = nls.GMRES(p.n, restart=30, ortho="cgs2").set_operator(op)
    G2.set_chebyshev_preconditioner(8, ratio=30.0)
    x2, info2 = G2.solve(b, abstol=0.0, reltol=1e-10, maxiters=2000)
    assert info["converged"] and info["iters"] == info2["iters"]
    assert np.linalg.norm(J @ x - b) <= 1.01e-10 * np.linalg.norm(b) and np.linalg.norm(x - x2) <= 1e-9 * np.linalg.norm(x2)


def test_gmres_matrix_free_and_callable_operator(nls, dev):
    import torch
    ns = 32
    p = R.Bratu2D(ns)
    u = 0.1 * np.random.default_rng(3).standard_normal(p.n)
    J = p.jac(u)
    b = np.random.default_rng(4).standard_normal(p.n)
    xref, _ = R.gmres(lambda z: J @ z, b, rtol=1e-9, itmax=3000)
    prob = nls.NonlinearProblem(nls.Bratu2D(ns))
    op = nls.StatefulJacobianOperator(nls.JacobianOperator(prob), u)
    G = nls.GMRES(p.n).set_operator(op)
    x, info = G.solve(b, reltol=1e-9, maxiters=3000)
    assert info["converged"] and np.linalg.norm(x - xref) <= 1e-7 * np.linalg.norm(xref)
    # generic AbstractSciMLOperator-style callable on device tensors
    Jd = torch.sparse_csr_tensor(torch.tensor(J.indptr, dtype=torch.int64), torch.tensor(J.indices, dtype=torch.int64),
                                 torch.tensor(J.data), size=J.shape, device=dev)
    G2 = nls.GMRES(p.n).set_operator(lambda z: Jd @ z)
    x2, info2 = G2.solve(torch.tensor(b, device=dev), reltol=1e-9, maxiters=3000)
    assert info2["converged"] and np.linalg.norm(x2.cpu().numpy() - xref) <= 1e-7 * np.linalg.norm(xref)


def test_gmres_zero_rhs_and_warm_start(nls):
    p = R.Bratu2D(8)
    J = p.jac(np.zeros(p.n))
    G = nls.GMRES(p.n).set_operator(nls.CSRMatrix.from_scipy(J))
    x, info = G.solve(np.zeros(p.n))
    assert info["converged"] and info["iters"] == 0 and np.all(x == 0)
    b = np.ones(p.n)
    xs, _ = G.solve(b, reltol=1e-12, maxiters=1000)
    # warm start: r0 = b − A x0 is already below the absolute tolerance ⇒ no Arnoldi step
    x2, info2 = G.solve(b, x0=xs, abstol=1e-6 * np.linalg.norm(b), reltol=0.0)
    assert info2["iters"] == 0 and info2["converged"] and np.allclose(x2, xs)
    assert info2["rnorm0"] <= 1e-10 * np.linalg.norm(b)


def test_gmres_nan_is_failure(nls):
    p = R.Bratu2D(8)
    J = p.jac(np.zeros(p.n))
    G = nls.GMRES(p.n).set_operator(nls.CSRMatrix.from_scipy(J))
    b = np.ones(p.n)
    b[3] = np.nan
    _, info = G.solve(b)
    assert info["failed"]


# ------------------------------------------------------------------ Newton–Raphson
def test_quadratic_newton_gmres(nls):
    """common/common_rootfind_testing.jl: quadratic_f, u0 = ones, p = 2 ⇒ sqrt(2), err < 1e-9."""
    prob = nls.NonlinearProblem(nls.Quadratic(1000, 2.0))
    sol = nls.solve(prob, nls.NewtonRaphson(linsolve=nls.KrylovJL_GMRES()), abstol=1e-9)
    assert sol.successful_retcode
    assert np.max(np.abs(sol.u - np.sqrt(2.0))) < 1e-9
    assert np.max(np.abs(sol.resid)) < 1e-9
    ref = R.solve(R.Quadratic(1000), R.NewtonRaphson(linsolve=R.KrylovJL_GMRES()), abstol=1e-9)
    assert sol.stats.nsteps == ref.stats.nsteps and sol.stats.nf == ref.stats.nf


@pytest.mark.parametrize("concrete", [False, True])
@pytest.mark.parametrize("ortho", ["mgs", "cgs2", "dcgs2"])
def test_bratu_newton_ew_vs_oracle(nls, concrete, ortho):
    ns = 48
    prob = nls.NonlinearProblem(nls.Bratu2D(ns, 6.0))
    alg = nls.NewtonRaphson(linsolve=nls.KrylovJL_GMRES(ortho=ortho), forcing=nls.EisenstatWalkerForcing2(),
                            concrete_jac=concrete)
    sol = nls.solve(prob, alg, abstol=1e-8, maxiters=50, store_trace=True)
    ref = R.solve(R.Bratu2D(ns), R.NewtonRaphson(linsolve=R.KrylovJL_GMRES(), forcing=R.EisenstatWalkerForcing2(),
                                                 concrete_jac=concrete), abstol=1e-8, maxiters=50)
    assert sol.retcode == "Success" == R.RETCODE_NAMES[ref.retcode]
    assert np.max(np.abs(sol.resid)) <= 1e-8
    assert uerr(sol.u, ref.u) <= 1e-8 * 50  # both are only converged to ‖F‖∞ ≤ 1e-8 with loose inner solves
    assert abs(sol.stats.nsteps - ref.stats.nsteps) <= 1
    # forcing sequence (with the reference's one-step lag) matches the oracle's at the first steps
    assert sol.trace[0]["eta"] == 0.5 and sol.trace[1]["eta"] == 0.9
    assert abs(sol.trace[2]["eta"] - ref.trace[2]["eta"]) <= 1e-6


def test_bratu_newton_tight_inner_matches_direct(nls):
    """With near-exact linear solves Newton–Krylov reproduces Newton + sparse direct solve to 1e-8."""
    ns = 32
    prob = nls.NonlinearProblem(nls.Bratu2D(ns, 6.0))
    alg = nls.NewtonRaphson(linsolve=nls.KrylovJL_GMRES(maxiters=5000, reltol=1e-12, abstol=0.0))
    sol = nls.solve(prob, alg, abstol=1e-10, maxiters=50)
    ref = R.solve(R.Bratu2D(ns), R.NewtonRaphson(), abstol=1e-10, maxiters=50)
    assert sol.retcode == "Success"
    assert uerr(sol.u, ref.u) <= 1e-8
    assert sol.stats.nsteps == ref.stats.nsteps


def test_iterator_interface_and_reinit(nls):
    """nlprob_iterator_interface (common_rootfind_testing.jl:47-57): reinit!(cache, u; p) + solve! ≈ sqrt.(p)."""
    prob = nls.NonlinearProblem(nls.Quadratic(4, 1.0), u0=np.full(4, 0.5))
    cache = nls.init(prob, nls.NewtonRaphson(linsolve=nls.KrylovJL_GMRES()), maxiters=100, abstol=1e-10)
    for p in np.linspace(1.0, 10.0, 12):
        nls.reinit_(cache, cache.u, p=float(p))
        sol = nls.solve_(cache)
        assert sol.retcode == "Success"
        assert np.allclose(sol.u, np.sqrt(p), atol=1e-9)
    cache.close()


def test_step_by_step(nls):
    prob = nls.NonlinearProblem(nls.Quadratic(10, 2.0))
    cache = nls.init(prob, nls.NewtonRaphson(linsolve=nls.KrylovJL_GMRES()), abstol=1e-9)
    oc = R.init(R.Quadratic(10), R.NewtonRaphson(linsolve=R.KrylovJL_GMRES()), abstol=1e-9)
    for _ in range(3):
        nls.step_(cache)
        oc.step()
        assert np.allclose(cache.u, oc.u, rtol=1e-12)
        assert cache.nsteps == oc.nsteps
    cache.close()


def test_maxiters_retcode(nls):
    prob = nls.NonlinearProblem(nls.Bratu2D(32))
    sol = nls.solve(prob, nls.NewtonRaphson(linsolve=nls.KrylovJL_GMRES(fixed_iters=2)), abstol=1e-12, maxiters=3)
    assert sol.retcode == "MaxIters" and sol.stats.nsteps == 3 and sol.stats.gmres_iters == 6


def test_user_function_custom_jvp(nls, dev):
    """rootfind_tests__item20.jl: F(u) = u + 0.1 u .* (Δ u) − p with an analytic JVP, N = 100, GMRES."""
    import torch
    N = 100
    rng = np.random.default_rng(20)
    u0 = rng.random(N)
    D = (torch.diag(2 * torch.ones(N)) - torch.diag(torch.ones(N - 1), 1) - torch.diag(torch.ones(N - 1), -1)).to(
        dev, torch.float64)
    pvec = torch.tensor(u0, device=dev)

    def F(du, u, p):
        du.copy_(u + 0.1 * u * (D @ u) - p)

    def JVP(out, v, u, p):
        out.copy_(v + 0.1 * (u * (D @ v) + v * (D @ u)))

    def VJP(out, v, u, p):
        out.copy_(v + 0.1 * (D @ (u * v) + v * (D @ u)))

    prob = nls.NonlinearProblem(nls.NonlinearFunction(F, jvp=JVP, vjp=VJP), torch.tensor(u0, device=dev), pvec)
    sol = nls.solve(prob, nls.NewtonRaphson(linsolve=nls.KrylovJL_GMRES()), abstol=1e-13)
    assert float(sol.resid.abs().max()) < 1e-6
    sol = nls.solve(prob, nls.TrustRegion(linsolve=nls.KrylovJL_GMRES()), abstol=1e-13)
    assert float(sol.resid.abs().max()) < 1e-6


# ------------------------------------------------------------------ TrustRegion
@pytest.mark.parametrize("scheme", list(range(7)))
def test_trust_region_quadratic_all_schemes(nls, scheme):
    """rootfind_tests__item8.jl: TrustRegion × radius update schemes × GMRES on quadratic_f (abstol 1e-6…1e-9)."""
    prob = nls.NonlinearProblem(nls.Quadratic(12, 2.0))
    sol = nls.solve(prob, nls.TrustRegion(linsolve=nls.KrylovJL_GMRES(), radius_update_scheme=scheme), abstol=1e-9)
    assert sol.retcode == "Success"
    assert np.max(np.abs(sol.resid)) < 1e-9
    ref = R.solve(R.Quadratic(12), R.TrustRegion(linsolve=R.KrylovJL_GMRES(), radius_update_scheme=scheme), abstol=1e-9)
    assert sol.stats.nsteps == ref.stats.nsteps
    assert uerr(sol.u, ref.u) <= 1e-9


@pytest.mark.parametrize("N,restart,concrete", [(16, 30, False), (16, 30, True), (32, 60, False)])
def test_trust_region_brusselator_vs_oracle(nls, N, restart, concrete):
    """Brusselator (sparsity_tests__item1.jl kernel; N = 32 is the reference's size), TrustRegion + GMRES,
    ‖resid‖∞ < 1e-8. Unpreconditioned GMRES(30) stagnates at N = 32 on the oracle as well (α = 10·31²), so the
    N = 32 case uses GMRES(60); linear solves are run to 1e-10 so that device and oracle take the same steps."""
    lin = dict(gmres_restart=restart, maxiters=8000, reltol=1e-10, abstol=0.0)
    prob = nls.NonlinearProblem(nls.Brusselator2D(N))
    sol = nls.solve(prob, nls.TrustRegion(linsolve=nls.KrylovJL_GMRES(**lin), concrete_jac=concrete), abstol=1e-8,
                    maxiters=25, store_trace=True)
    rb = R.Brusselator2D(N)
    oc = R.init(rb, R.TrustRegion(linsolve=R.KrylovJL_GMRES(gmres_restart=restart, maxiters=8000),
                                  concrete_jac=concrete), abstol=1e-8, maxiters=25)
    oc.lin_reltol, oc.lin_abstol = 1e-10, 0.0
    ref = oc.solve()
    assert sol.retcode == "Success" == R.RETCODE_NAMES[ref.retcode]
    assert np.max(np.abs(sol.resid)) < 1e-8
    assert uerr(sol.u, ref.u) <= 1e-8
    if N == 32:
        # The third linear system of this run is at the edge of what restarted GMRES(60) can do within 8000 iterations
        # (from a cold start the oracle's MGS stalls at 1e-6, its CGS2 at 1e-4, its DCGS2 converges after 6541): which side of
        # the edge a run lands on depends on rounding, so only the first two (well-conditioned) steps are compared one to one.
        assert abs(sol.stats.nsteps - ref.stats.nsteps) <= 1
        pairs = list(zip(sol.trace, ref.trace))[:2]
    else:
        assert sol.stats.nsteps == ref.stats.nsteps
        pairs = list(zip(sol.trace, ref.trace))
    for a, b in pairs:
        assert a["accepted"] == b["accepted"]
        assert abs(a["trust_region"] - b["trust_region"]) <= 1e-6 * b["trust_region"]


def test_trust_region_maxiters_sweep_matches_oracle(nls):
    """rootfind_tests__item12.jl: TR iterates agree at maxiters ∈ {2,3,4,5} (here: device vs oracle)."""
    for mi in (2, 3, 4, 5):
        prob = nls.NonlinearProblem(nls.Quadratic(6, 2.0), u0=np.full(6, 3.0))
        sol = nls.solve(prob, nls.TrustRegion(linsolve=nls.KrylovJL_GMRES()), maxiters=mi, abstol=1e-14)
        ref = R.solve(R.Quadratic(6), R.TrustRegion(linsolve=R.KrylovJL_GMRES()), maxiters=mi, abstol=1e-14,
                      u0=np.full(6, 3.0))
        assert np.allclose(sol.u, ref.u, rtol=1e-10)


def test_jacobian_operators(nls):
    """core_tests__item2.jl: sop*v ≈ J v, sop'*v ≈ Jᵀ v, (sop'*sop)*v ≈ JᵀJ v (atol 1e-5) — Brusselator J."""
    N = 8
    b = R.Brusselator2D(N)
    prob = nls.NonlinearProblem(nls.Brusselator2D(N))
    jac_op = nls.JacobianOperator(prob)
    rng = np.random.default_rng(0)
    for _ in range(4):
        u, v = b.u0() + rng.random(b.n), rng.random(b.n)
        J = b.jac(u).toarray()
        sop = nls.StatefulJacobianOperator(jac_op, u)
        assert np.allclose(sop @ v, J @ v, atol=1e-5)
        assert np.allclose(sop.T @ v, J.T @ v, atol=1e-5)
        assert np.allclose((sop.T @ sop) @ v, J.T @ (J @ v), atol=1e-5, rtol=1e-10)


# ------------------------------------------------------------------ direct linsolve (linsolve = nothing → banded LU)
def test_quadratic_default_linsolve(nls):
    """Config C1 / rootfind_tests: NewtonRaphson() with the default (direct) linear solver on quadratic_f."""
    prob = nls.NonlinearProblem(nls.Quadratic(1000, 2.0))
    sol = nls.solve(prob, nls.NewtonRaphson())
    ref = R.solve(R.Quadratic(1000), R.NewtonRaphson())
    assert sol.retcode == "Success" and np.max(np.abs(sol.resid)) <= 3e-13
    assert np.max(np.abs(sol.u - np.sqrt(2.0))) < 1e-12
    assert (sol.stats.nsteps, sol.stats.nf, sol.stats.njacs, sol.stats.nfactors, sol.stats.nsolve) == \
        (ref.stats.nsteps, ref.stats.nf, ref.stats.njacs, ref.stats.nfactors, ref.stats.nsolve)


@pytest.mark.parametrize("ns", [5, 33, 64])
def test_bratu_direct_newton_vs_oracle(nls, ns):
    """Config C2 protocol at small sizes: NewtonRaphson + concrete sparse J + direct solve (SuperLU on the oracle)."""
    prob = nls.NonlinearProblem(nls.Bratu2D(ns, 6.0))
    sol = nls.solve(prob, nls.NewtonRaphson(), abstol=1e-10, maxiters=50, store_trace=True)
    ref = R.solve(R.Bratu2D(ns), R.NewtonRaphson(), abstol=1e-10, maxiters=50)
    assert sol.retcode == "Success" == R.RETCODE_NAMES[ref.retcode]
    assert uerr(sol.u, ref.u) <= 1e-10
    assert sol.stats.nsteps == ref.stats.nsteps and sol.stats.nfactors == ref.stats.nfactors
    for a, b in zip(sol.trace, ref.trace):
        assert abs(a["fnorm_inf"] - b["fnorm_inf"]) <= 1e-9 * max(b["fnorm_inf"], 1e-9) + 1e-14


def test_c2_bratu_256_direct(nls):
    """Config C2 at full size (n = 256², N = 65 536, band 513 × 65 536): Newton + direct solve; solution checked
    against the oracle's sparse-direct Newton and through the residual."""
    ns = 256
    prob = nls.NonlinearProblem(nls.Bratu2D(ns, 6.0))
    sol = nls.solve(prob, nls.NewtonRaphson(), abstol=1e-8, maxiters=50)
    ref = R.solve(R.Bratu2D(ns), R.NewtonRaphson(), abstol=1e-8, maxiters=50)
    assert sol.retcode == "Success" and sol.stats.nsteps == ref.stats.nsteps <= 6
    assert np.max(np.abs(sol.resid)) <= 1e-8
    assert uerr(sol.u, ref.u) <= 1e-9
    assert 0.79 < float(np.max(sol.u)) < 0.80


def test_trust_region_direct_reuses_factorisation(nls):
    """reuse_A_if_factorization = !new_jacobian: a rejected trust-region step must not refactorise."""
    ns = 24
    prob = nls.NonlinearProblem(nls.Bratu2D(ns, 6.0), u0=np.full(ns * ns, 3.0))  # far start ⇒ some rejections
    sol = nls.solve(prob, nls.TrustRegion(), abstol=1e-9, maxiters=12, store_trace=True)
    ref = R.solve(R.Bratu2D(ns), R.TrustRegion(), abstol=1e-9, maxiters=12, u0=np.full(ns * ns, 3.0))
    assert not all(t["accepted"] for t in ref.trace)  # the scenario really contains rejected steps
    assert sol.retcode == R.RETCODE_NAMES[ref.retcode]
    assert sol.stats.nsteps == ref.stats.nsteps
    assert sol.stats.nfactors == ref.stats.nfactors and sol.stats.njacs == ref.stats.njacs
    assert [t["accepted"] for t in sol.trace] == [t["accepted"] for t in ref.trace]
    if sol.retcode == "Success":
        assert uerr(sol.u, ref.u) <= 1e-8


def test_banded_lu_seam_vs_scipy(nls, dev):
    """linear_solver_routing.jl:44-61 — res.u ≈ A \\ b for the factorisation seam; refactor after A changes."""
    import scipy.sparse as sp
    import scipy.sparse.linalg as spla
    import torch
    rng = np.random.default_rng(0)
    n, bw = 700, 9
    diags = [rng.standard_normal(n - abs(k)) for k in range(-bw, bw + 1)]
    A = sp.diags(diags, list(range(-bw, bw + 1)), format="csr") + sp.identity(n) * (4.0 * bw)  # diagonally dominant
    A = sp.csr_matrix(A)
    M = nls.CSRMatrix.from_scipy(A)
    F = nls.BandedLU(M)
    assert F.info()["kl"] == bw and F.info()["ku"] == bw
    b = rng.standard_normal(n)
    x = F.solve(b)
    assert np.allclose(x, spla.spsolve(A.tocsc(), b), rtol=1e-11, atol=1e-12)
    xd = F.solve(torch.tensor(b, device=dev))
    assert np.allclose(xd.cpu().numpy(), x, rtol=1e-14)
    A2 = sp.csr_matrix(A + sp.identity(n))
    M.set_values(A2.data)
    F.factor()
    assert np.allclose(F.solve(b), spla.spsolve(A2.tocsc(), b), rtol=1e-11, atol=1e-12)


# ------------------------------------------------------------------ more reference fixtures / edge cases
@pytest.mark.parametrize("lin", ["gmres", "gmres_concrete", "direct"])
def test_operator_jacobian_tridiagonal(nls, dev, lin):
    """operator_jacobian.jl:11-29 — linear residual W z − b with `jac_prototype`, N = 40: sol.u ≈ W \\ b for
    GMRES (matrix-free), GMRES on the concrete J, and the factorisation."""
    import scipy.sparse as sp
    import scipy.sparse.linalg as spla
    import torch
    N = 40
    W = sp.csr_matrix(sp.diags([-np.ones(N - 1), 4.0 * np.ones(N), -np.ones(N - 1)], [-1, 0, 1]))
    bvec = np.arange(1.0, N + 1)
    xref = spla.spsolve(W.tocsc(), bvec)
    Wd = nls.CSRMatrix.from_scipy(W)             # used inside the residual (device SpMV)
    proto = nls.CSRMatrix.from_scipy(W)          # jac_prototype
    bd = torch.tensor(bvec, device=dev)
    wvals = torch.tensor(W.data, device=dev)

    def resid(F, z, p):
        Wd.matvec(z, out=F)
        F.sub_(bd)

    def jvp(Jv, v, z, p):
        Wd.matvec(v, out=Jv)

    def jac(nzval, z, p):
        nzval.copy_(wvals)

    f = nls.NonlinearFunction(resid, jvp=jvp, jac=jac, jac_prototype=proto)
    prob = nls.NonlinearProblem(f, torch.zeros(N, dtype=torch.float64, device=dev))
    alg = {"gmres": nls.NewtonRaphson(linsolve=nls.KrylovJL_GMRES()),
           "gmres_concrete": nls.NewtonRaphson(linsolve=nls.KrylovJL_GMRES(), concrete_jac=True),
           "direct": nls.NewtonRaphson()}[lin]
    sol = nls.solve(prob, alg)
    assert sol.successful_retcode
    assert np.allclose(sol.u.cpu().numpy(), xref, rtol=1e-8)


@pytest.mark.parametrize("n", [1, 2, 3])
def test_tiny_systems(nls, n):
    for alg in (nls.NewtonRaphson(linsolve=nls.KrylovJL_GMRES()), nls.NewtonRaphson(),
                nls.TrustRegion(linsolve=nls.KrylovJL_GMRES()), nls.TrustRegion()):
        sol = nls.solve(nls.NonlinearProblem(nls.Quadratic(n, 2.0)), alg, abstol=1e-9)
        assert sol.retcode == "Success" and np.allclose(sol.u, np.sqrt(2.0), atol=1e-9)


def test_unstable_retcode_on_nan_residual(nls, dev):
    """termination_conditions.jl:260-264 — a non-finite objective ends the solve with ReturnCode.Unstable."""
    import torch

    def f(du, u, p):
        du.copy_(torch.where(u > 5.0, torch.full_like(u, float("nan")), u * u - 100.0))

    def jvp(Jv, v, u, p):
        Jv.copy_(2.0 * u * v)

    prob = nls.NonlinearProblem(nls.NonlinearFunction(f, jvp=jvp), torch.ones(4, dtype=torch.float64, device=dev))
    sol = nls.solve(prob, nls.NewtonRaphson(linsolve=nls.KrylovJL_GMRES()))
    assert sol.retcode == "Unstable"


def test_maxtime_retcode(nls):
    prob = nls.NonlinearProblem(nls.Bratu2D(128))
    sol = nls.solve(prob, nls.NewtonRaphson(linsolve=nls.KrylovJL_GMRES(fixed_iters=30, maxiters=30)), abstol=1e-300,
                    maxiters=10 ** 6, maxtime=0.05)
    assert sol.retcode == "MaxTime" and 1 <= sol.stats.nsteps < 10 ** 6


def test_best_iterate_rollback(nls):
    """Safe-best mode: the returned u is the iterate with the smallest ‖f‖∞ seen (termination_conditions.jl:440-453)."""
    prob = nls.NonlinearProblem(nls.Bratu2D(16))
    sol = nls.solve(prob, nls.NewtonRaphson(linsolve=nls.KrylovJL_GMRES(fixed_iters=3, maxiters=3, ortho="mgs")),
                    abstol=1e-14, maxiters=8, store_trace=True)
    ref = R.solve(R.Bratu2D(16), R.NewtonRaphson(linsolve=R.KrylovJL_GMRES(fixed_iters=3)), abstol=1e-14, maxiters=8)
    best = min(t["fnorm_inf"] for t in sol.trace)
    assert np.isclose(np.max(np.abs(sol.resid)), best, rtol=1e-12)
    assert sol.retcode == "MaxIters" == R.RETCODE_NAMES[ref.retcode]
    assert uerr(sol.u, ref.u) <= 1e-9


def test_right_preconditioner_hook(nls, dev):
    """precs hook (test/Core/core_tests__item21.jl): a Jacobi right preconditioner is applied through the device
    callback, is called, and GMRES converges to the same solution."""
    import scipy.sparse as sp
    import torch
    rng = np.random.default_rng(0)
    n = 400
    d = np.linspace(1.0, 1e4, n)                      # badly scaled diagonal + weak coupling
    A = sp.csr_matrix(sp.diags(d) + 0.1 * sp.diags([np.ones(n - 1), np.ones(n - 1)], [-1, 1]))
    b = rng.standard_normal(n)
    xref = np.linalg.solve(A.toarray(), b)
    M = nls.CSRMatrix.from_scipy(A)
    G0 = nls.GMRES(n, restart=30).set_operator(M)
    x0, i0 = G0.solve(b, reltol=1e-10, maxiters=3000)
    calls = [0]
    dinv = torch.tensor(1.0 / d, device=dev)

    def jacobi(x):
        calls[0] += 1
        return dinv * x

    G1 = nls.GMRES(n, restart=30).set_operator(M).set_right_preconditioner(jacobi)
    x1, i1 = G1.solve(b, reltol=1e-10, maxiters=3000)
    assert i1["converged"] and calls[0] >= i1["iters"]
    assert np.linalg.norm(x1 - xref) <= 1e-7 * np.linalg.norm(xref)
    assert i1["iters"] < i0["iters"]


def test_bitwise_reproducible(nls):
    """Fixed-order two-stage reductions, no float atomics: the same solve twice gives bit-identical iterates."""
    outs = []
    for _ in range(2):
        prob = nls.NonlinearProblem(nls.Bratu2D(96, 6.0))
        sol = nls.solve(prob, nls.NewtonRaphson(linsolve=nls.KrylovJL_GMRES(), forcing=nls.EisenstatWalkerForcing2()),
                        abstol=1e-8, maxiters=50)
        outs.append((np.asarray(sol.u).copy(), sol.stats.gmres_iters))
    assert outs[0][1] == outs[1][1]
    assert np.array_equal(outs[0][0], outs[1][0])


# ------------------------------------------------------------------ termination conditions
@pytest.mark.parametrize("idx", list(range(9)))
@pytest.mark.parametrize("algname", ["nr", "tr"])
def test_all_termination_conditions(nls, idx, algname):
    """rootfind_tests__item4.jl / __item7.jl: `solve(prob, alg; termination_condition)` over the nine
    TERMINATION_CONDITIONS on quadratic_f, err < 1e-9 — and the same step count / retcode as the oracle."""
    tc = nls.TERMINATION_CONDITIONS[idx]()
    alg = nls.NewtonRaphson(linsolve=nls.KrylovJL_GMRES()) if algname == "nr" else nls.TrustRegion(linsolve=nls.KrylovJL_GMRES())
    sol = nls.solve(nls.NonlinearProblem(nls.Quadratic(2, 2.0)), alg, termination_condition=tc)
    assert np.max(np.abs(sol.u * sol.u - 2.0)) < 1e-9 and sol.retcode == "Success"
    oalg = R.NewtonRaphson(linsolve=R.KrylovJL_GMRES()) if algname == "nr" else R.TrustRegion(linsolve=R.KrylovJL_GMRES())
    ref = R.solve(R.Quadratic(2, 2.0), oalg, termination_kwargs=dict(mode=tc.code, max_stalled_steps=None))
    assert sol.stats.nsteps == ref.stats.nsteps and sol.retcode == R.RETCODE_NAMES[ref.retcode]


@pytest.mark.parametrize("norm", ["inf", "l2"])
def test_termination_internalnorm_and_relative_modes_on_bratu(nls, norm):
    """Relative / 2-norm modes on a real problem: same stopping step as the oracle."""
    for cls in (nls.RelNormSafeBestTerminationMode, nls.NormTerminationMode, nls.AbsNormTerminationMode):
        tc = cls(internalnorm=norm)
        prob = nls.NonlinearProblem(nls.Bratu2D(24, 6.0), u0=np.full(24 * 24, 0.5))
        lin = dict(maxiters=4000, reltol=1e-11, abstol=0.0)
        sol = nls.solve(prob, nls.NewtonRaphson(linsolve=nls.KrylovJL_GMRES(**lin)), abstol=1e-9, reltol=1e-9,
                        termination_condition=tc, maxiters=30)
        oc = R.init(R.Bratu2D(24), R.NewtonRaphson(linsolve=R.KrylovJL_GMRES(maxiters=4000)), abstol=1e-9, reltol=1e-9,
                    maxiters=30, u0=np.full(24 * 24, 0.5),
                    termination_kwargs=dict(mode=tc.code, norm=norm, max_stalled_steps=None))
        oc.lin_reltol, oc.lin_abstol = 1e-11, 0.0
        ref = oc.solve()
        assert sol.retcode == R.RETCODE_NAMES[ref.retcode] == "Success"
        assert sol.stats.nsteps == ref.stats.nsteps
        assert uerr(sol.u, ref.u) <= 1e-8


# ------------------------------------------------------------------ built-in Chebyshev preconditioner (`precs`)
@pytest.mark.parametrize("op", ["csr", "matfree"])
def test_chebyshev_preconditioned_gmres(nls, dev, op):
    ns = 96
    p = R.Bratu2D(ns)
    u = 0.2 * np.random.default_rng(0).standard_normal(p.n)
    J = p.jac(u)
    b = np.random.default_rng(1).standard_normal(p.n)
    G = nls.GMRES(p.n, restart=30)
    if op == "csr":
        G.set_operator(nls.CSRMatrix.from_scipy(J))
    else:
        G.set_operator(nls.StatefulJacobianOperator(nls.JacobianOperator(nls.NonlinearProblem(nls.Bratu2D(ns))), u))
    x0, i0 = G.solve(b, reltol=1e-8, maxiters=20000)
    G.set_chebyshev_preconditioner(16, ratio=100.0)
    lmin, lmax = G.chebyshev_interval()
    gl = R.gershgorin_lambda(J)
    if op == "csr":
        assert np.isclose(lmax, gl, rtol=1e-12) and np.isclose(lmin, gl / 100, rtol=1e-12)
    else:  # power iteration ×1.15 must still bound the spectrum (8·c_lap is its supremum) without gross excess
        assert 0.97 * gl <= lmax <= 1.25 * gl
    x1, i1 = G.solve(b, reltol=1e-8, maxiters=2000)
    assert i1["converged"] and i1["iters"] * 10 < i0["iters"]
    xd = np.linalg.solve(J.toarray(), b) if p.n <= 4096 else __import__("scipy.sparse.linalg").sparse.linalg.spsolve(J.tocsc(), b)
    assert np.linalg.norm(x1 - xd) <= 1e-6 * np.linalg.norm(xd)
    if op == "csr":  # same algorithm, same interval ⇒ the oracle takes (almost) the same number of steps
        M = R.chebyshev_preconditioner(lambda z: J @ z, gl / 100, gl, 16)
        xr, ir = R.gmres(lambda z: J @ z, b, rtol=1e-8, itmax=2000, M=M)
        assert abs(ir.iters - i1["iters"]) <= 2 and np.linalg.norm(x1 - xr) <= 1e-6 * np.linalg.norm(xr)


def test_newton_with_chebyshev_precs_vs_oracle(nls):
    ns = 128
    alg = nls.NewtonRaphson(linsolve=nls.KrylovJL_GMRES(precs=nls.ChebyshevPrecs(16, 100.0)),
                            forcing=nls.EisenstatWalkerForcing2(), concrete_jac=True)
    sol = nls.solve(nls.NonlinearProblem(nls.Bratu2D(ns)), alg, abstol=1e-9, maxiters=50)
    ref = R.solve(R.Bratu2D(ns), R.NewtonRaphson(linsolve=R.KrylovJL_GMRES(precs=R.ChebyshevPrecs(16, 100.0)),
                                                 forcing=R.EisenstatWalkerForcing2(), concrete_jac=True),
                  abstol=1e-9, maxiters=50)
    direct = R.solve(R.Bratu2D(ns), R.NewtonRaphson(), abstol=1e-9, maxiters=50)
    assert sol.retcode == "Success" == R.RETCODE_NAMES[ref.retcode]
    assert abs(sol.stats.nsteps - ref.stats.nsteps) <= 1
    assert uerr(sol.u, direct.u) <= 5e-7 and uerr(sol.u, ref.u) <= 5e-7


# ------------------------------------------------------------------ line search globalisation
def test_backtracking_linesearch_vs_oracle(nls, dev):
    """rootfind_tests__item2.jl (NewtonRaphson × line searches) — BackTracking on quadratic_f, and on atan(u) where
    plain Newton diverges; α sequence, residual-evaluation counts and iterates equal the oracle's."""
    import scipy.sparse as sp
    import torch
    sol = nls.solve(nls.NonlinearProblem(nls.Quadratic(3, 2.0)),
                    nls.NewtonRaphson(linsolve=nls.KrylovJL_GMRES(), linesearch=nls.BackTracking()), abstol=1e-9)
    ref = R.solve(R.Quadratic(3, 2.0), R.NewtonRaphson(linsolve=R.KrylovJL_GMRES(), linesearch=R.BackTracking()), abstol=1e-9)
    assert sol.retcode == "Success" and sol.stats.nsteps == ref.stats.nsteps and sol.stats.nf == ref.stats.nf

    def f(du, u, p):
        du.copy_(torch.atan(u))

    def jvp(Jv, v, u, p):
        Jv.copy_(v / (1.0 + u * u))

    u0 = np.array([2.0, -3.0, 1.5])
    prob = nls.NonlinearProblem(nls.NonlinearFunction(f, jvp=jvp), torch.tensor(u0, device=dev))
    lin = nls.KrylovJL_GMRES(reltol=1e-14, abstol=0.0)
    plain = nls.solve(prob, nls.NewtonRaphson(linsolve=lin), abstol=1e-10, maxiters=30)
    ls = nls.solve(prob, nls.NewtonRaphson(linsolve=lin, linesearch=nls.BackTracking()), abstol=1e-10, maxiters=60,
                   store_trace=True)
    atan = R.FunctionProblem(np.arctan, u0, jac=lambda u: sp.diags(1.0 / (1.0 + u * u)))
    oref = R.solve(atan, R.NewtonRaphson(linesearch=R.BackTracking()), abstol=1e-10, maxiters=60)
    assert plain.retcode != "Success"
    assert ls.retcode == "Success" and float(ls.u.abs().max()) < 1e-9
    assert ls.stats.nsteps == oref.stats.nsteps and ls.stats.nf == oref.stats.nf
    for a, b in zip(ls.trace, oref.trace):
        assert abs(a["step_norm2"] - b["step_norm2"]) <= 1e-9 * max(b["step_norm2"], 1e-12) + 1e-13


def test_linesearch_with_trust_region_is_rejected(nls):
    import ctypes as C
    from nonlinearsolve_jl_amd import _lib as L
    o = L.Options()
    L.lib().nk_options_default(C.byref(o))
    o.algorithm, o.linesearch = L.ALG_TRUST_REGION, 1
    P = nls.Quadratic(2, 2.0)
    h = C.c_void_p()
    u0 = np.ones(2)
    st = L.lib().nk_solver_init(P._h, C.c_void_p(u0.ctypes.data), L.HOST, C.byref(o), C.byref(h))
    assert st != 0 and b"incompatible" in L.lib().nk_last_error()


# ------------------------------------------------------------------ NonlinearFunction with f only (no jvp / jac)
def _bratu1d(dev, n=200, lam=1.0):
    import torch
    h2 = (1.0 / (n + 1)) ** 2

    def F(du, u, p):
        du.copy_(2.0 * u - h2 * p * torch.exp(u))
        du[1:] -= u[:-1]
        du[:-1] -= u[1:]

    def J_dense(u):
        J = np.diag(2.0 - h2 * lam * np.exp(u)) - np.diag(np.ones(n - 1), 1) - np.diag(np.ones(n - 1), -1)
        return J
    return F, J_dense


def test_user_function_without_jvp_uses_forward_differences(nls, dev):
    """NonlinearFunction(f) only: the JacobianOperator falls back to a finite-difference pushforward
    (SciMLJacobianOperators.jl:396-414 with AutoFiniteDiff); Newton–GMRES converges to the analytic-J solution."""
    import torch
    n = 200
    F, J_dense = _bratu1d(dev, n)
    prob = nls.NonlinearProblem(nls.NonlinearFunction(F), torch.zeros(n, device=dev, dtype=torch.float64), 1.0)
    u = np.linspace(0.0, 1.0, n)
    v = np.cos(np.arange(n))
    Jop = nls.StatefulJacobianOperator(nls.JacobianOperator(prob), torch.tensor(u, device=dev))
    Jv = (Jop @ torch.tensor(v, device=dev)).cpu().numpy()
    assert np.max(np.abs(Jv - J_dense(u) @ v)) < 1e-6          # forward differences, ε = √eps
    sol = nls.solve(prob, nls.NewtonRaphson(linsolve=nls.KrylovJL_GMRES(gmres_restart=60, reltol=1e-10)), abstol=1e-10)
    assert sol.retcode == "Success"
    uref = np.zeros(n)
    for _ in range(20):                                         # dense Newton on the host as the check
        f = 2 * uref - (1.0 / (n + 1)) ** 2 * np.exp(uref)
        f[1:] -= uref[:-1]
        f[:-1] -= uref[1:]
        uref -= np.linalg.solve(J_dense(uref), f)
    assert np.max(np.abs(sol.u.cpu().numpy() - uref)) < 1e-7


def test_user_function_sparse_prototype_coloured_fd_jacobian(nls, dev):
    """sparsity_tests__item1.jl shape: NonlinearFunction(f; jac_prototype = pattern) — the concrete sparse J is
    assembled from ncolors seeded (finite-difference) JVPs + decompression; `linsolve = nothing` then factorises it."""
    import torch
    import scipy.sparse as sp
    n = 200
    F, J_dense = _bratu1d(dev, n)
    pat = sp.csr_matrix(sp.diags([np.ones(n - 1), np.ones(n), np.ones(n - 1)], [-1, 0, 1]))
    proto = nls.CSRMatrix.from_scipy(pat)
    prob = nls.NonlinearProblem(nls.NonlinearFunction(F, jac_prototype=proto),
                                torch.zeros(n, device=dev, dtype=torch.float64), 1.0)
    u = np.linspace(0.0, 1.0, n)
    ncol = prob.device_problem.jac_values(u, proto, colored=True)
    assert ncol == 3                                            # tridiagonal ⇒ 3 structurally orthogonal groups
    J = sp.csr_matrix((proto.values(), pat.indices, pat.indptr), shape=(n, n)).toarray()
    assert np.max(np.abs(J - J_dense(u))) < 1e-6
    for alg in (nls.NewtonRaphson(), nls.NewtonRaphson(linsolve=nls.KrylovJL_GMRES(gmres_restart=60, reltol=1e-10),
                                                       concrete_jac=True)):
        sol = nls.solve(prob, alg, abstol=1e-10)
        assert sol.retcode == "Success" and float(sol.resid.abs().max()) < 1e-10
        assert sol.stats.njacs >= sol.stats.nsteps - 1


@pytest.mark.parametrize("n,kl,ku", [(33, 1, 1), (257, 3, 40), (1000, 70, 5), (2048, 256, 256), (1500, 300, 129),
                                     (64, 31, 32), (31, 2, 2)])
def test_banded_lu_shapes_vs_scipy(nls, n, kl, ku):
    """Band shapes around the block size (NB = 32): kl ≠ ku, bandwidth < NB, > 256 rows per panel pass, n not a
    multiple of NB, and the C2 shape kl = ku = 256 — factor + both substitution sweeps against SuperLU."""
    import scipy.sparse as sp
    import scipy.sparse.linalg as spla
    rng = np.random.default_rng(n + kl)
    offs = list(range(-kl, ku + 1))
    A = sp.diags([rng.standard_normal(n - abs(k)) for k in offs], offs, format="csr")
    A = sp.csr_matrix(A + sp.identity(n) * (2.0 * (kl + ku)))   # diagonally dominant: no pivoting needed
    F = nls.BandedLU(nls.CSRMatrix.from_scipy(A))
    assert F.info()["kl"] == kl and F.info()["ku"] == ku
    for _ in range(2):
        b = rng.standard_normal(n)
        x = F.solve(b)
        assert np.max(np.abs(A @ x - b)) <= 1e-12 * np.max(np.abs(b)) * (kl + ku)
        assert np.allclose(x, spla.spsolve(A.tocsc(), b), rtol=1e-10, atol=1e-13)


def test_newton_fails_converges_with_trust_region_on_device(nls, dev):
    """rootfind_tests__item10.jl: `newton_fails` (7 unknowns) must converge with TrustRegion(); the residual is given as
    a device function only — the Jacobian comes from the coloured finite-difference assembly on a diagonal prototype
    (the system is separable), J and Jᵀ products from the assembled CSR — and the result matches the oracle's."""
    import scipy.sparse as sp
    import torch

    def nf(u):
        return (0.010000000000000002 + 10.000000000000002 / (1 + (0.21640425613334457 + 216.40425613334457 / (
            1 + (0.21640425613334457 + 216.40425613334457 / (1 + 0.0006250000000000001 * (u ** 2.0))) ** 2.0)) ** 2.0)
            - 0.0011552453009332421 * u)

    def F(du, u, p):
        du.copy_(nf(u))

    u0 = np.array([-10.0, -1.0, 1.0, 2.0, 3.0, 4.0, 10.0])
    proto = nls.CSRMatrix.from_scipy(sp.csr_matrix(sp.identity(7)))
    prob = nls.NonlinearProblem(nls.NonlinearFunction(F, jac_prototype=proto), torch.tensor(u0, device=dev))
    sol = nls.solve(prob, nls.TrustRegion(linsolve=nls.KrylovJL_GMRES(reltol=1e-12, abstol=0.0), concrete_jac=True),
                    abstol=1e-9, maxiters=200)
    assert sol.retcode == "Success" and float(sol.resid.abs().max()) < 1e-9
    ref = R.solve(R.FunctionProblem(lambda u: nf(u), u0, jac=lambda u: sp.diags((nf(u + 1e-7) - nf(u - 1e-7)) / 2e-7)),
                  R.TrustRegion(), abstol=1e-9)
    assert np.max(np.abs(sol.u.cpu().numpy() - ref.u)) < 1e-5   # same root as the oracle (FD Jacobians on both sides)


# ------------------------------------------------------------------ geometric multigrid `precs`
@pytest.mark.parametrize("ns,coarse", [(64, 15), (100, 15), (127, 31), (33, 8)])
def test_multigrid_preconditioner_vs_oracle(nls, ns, coarse):
    """The built-in V-cycle (rediscretised level operators, bilinear transfers between non-nested grids, Chebyshev
    smoothing, banded LU on the coarsest grid) applied to a vector, and right-preconditioned GMRES with it, agree with the
    oracle's restatement; the iteration count does not grow with the grid (7 ± 1)."""
    pb = R.Bratu2D(ns, 6.0)
    xs = np.arange(1, ns + 1) / (ns + 1)
    X, Y = np.meshgrid(xs, xs)
    u = (0.8 * np.sin(np.pi * X) * np.sin(np.pi * Y)).ravel()
    J = pb.jac(u)
    b = np.random.default_rng(0).standard_normal(pb.n)
    Mo = R.BratuMultigrid(pb, u, 2, coarse)
    P = nls.Bratu2D(ns, 6.0)
    prob = nls.NonlinearProblem(P)
    op = nls.StatefulJacobianOperator(nls.JacobianOperator(prob), u)
    G = nls.GMRES(pb.n, restart=30).set_operator(op)
    G.set_multigrid_preconditioner(P, u, nu=2, coarse_max=coarse)
    # one preconditioner application = GMRES's first direction; compare through a 1-step solve and a full solve
    xo, io = R.gmres(lambda z: J @ z, b, rtol=1e-9, restart=30, itmax=300, M=Mo, ortho="cgs2")
    x, info = G.solve(b, abstol=0.0, reltol=1e-9, maxiters=300)
    assert info["converged"] and abs(info["iters"] - io.iters) <= 1 and info["iters"] <= 9
    assert np.linalg.norm(x - xo) <= 1e-7 * np.linalg.norm(xo)
    assert np.linalg.norm(J @ x - b) <= 1.01e-9 * np.linalg.norm(b)
    x1, i1 = G.solve(b, fixed_iters=1)
    xo1, _ = R.gmres(lambda z: J @ z, b, restart=30, fixed_iters=1, M=Mo, ortho="cgs2")
    assert np.linalg.norm(x1 - xo1) <= 1e-10 * np.linalg.norm(xo1)     # V-cycle itself: agreement to rounding


@pytest.mark.parametrize("concrete", [False, True])
def test_bratu_newton_with_multigrid_precs_vs_oracle(nls, concrete):
    ns = 128
    alg = nls.NewtonRaphson(linsolve=nls.KrylovJL_GMRES(precs=nls.MultigridPrecs(2, 15)), forcing=nls.EisenstatWalkerForcing2(),
                            concrete_jac=concrete)
    sol = nls.solve(nls.NonlinearProblem(nls.Bratu2D(ns, 6.0)), alg, abstol=1e-8, maxiters=50)
    ref = R.solve(R.Bratu2D(ns), R.NewtonRaphson(linsolve=R.KrylovJL_GMRES(precs=R.MultigridPrecs(2, 15)),
                                                forcing=R.EisenstatWalkerForcing2(), concrete_jac=concrete), abstol=1e-8, maxiters=50)
    assert sol.retcode == "Success" and np.max(np.abs(sol.resid)) <= 1e-8
    assert sol.stats.nsteps == ref.stats.nsteps and abs(sol.stats.gmres_iters - ref.stats.gmres_iters) <= 2
    assert sol.stats.gmres_iters <= 3 * sol.stats.nsteps          # ≈ one or two Krylov iterations per Newton step
    assert uerr(sol.u, ref.u) <= 1e-6


def test_gmres_dcgs2_long_restart(nls):
    """The one-reduction DCGS2 sweeps take the column count at run time, so long restarts (here GMRES(60), up to 62) use
    them too: same iterates as the oracle's restatement and as CGS2."""
    p = R.Bratu2D(24)
    J = p.jac(0.1 * np.random.default_rng(5).standard_normal(p.n))
    b = np.random.default_rng(2).standard_normal(p.n)
    A = nls.CSRMatrix.from_scipy(J)
    for k in (45, 60, 61, 100):
        xref, iref = R.gmres_dcgs2_1r(lambda z: J @ z, b, restart=60, fixed_iters=k)
        x, info = nls.GMRES(p.n, restart=60, ortho="dcgs2").set_operator(A).solve(b, fixed_iters=k)
        assert info["iters"] == k == iref.iters and abs(info["rnorm"] - iref.rnorm) <= 1e-10 * iref.rnorm0
        assert np.linalg.norm(x - xref) <= 1e-9 * np.linalg.norm(xref)
    xc, ic = R.gmres(lambda z: J @ z, b, rtol=1e-10, restart=60, itmax=3000, ortho="cgs2")
    x, info = nls.GMRES(p.n, restart=60).set_operator(A).solve(b, abstol=0.0, reltol=1e-10, maxiters=3000)
    assert info["converged"] and info["iters"] == ic.iters and np.linalg.norm(x - xc) <= 1e-9 * np.linalg.norm(xc)


def test_plain_c_caller_runs_config_c2(tmp_path):
    """The C ABI from plain C (examples/bratu_c2.c): config C2 (Bratu 256², direct) and the Newton–Krylov path converge."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = tmp_path / "bratu_c2"
    libdir = os.path.join(root, "nonlinearsolve.jl_amd", "lib")
    subprocess.check_call(["gcc", "-std=c99", "-I", os.path.join(root, "include"),
                           os.path.join(root, "examples", "bratu_c2.c"), "-L", libdir, "-lmi355x_nk", "-lm", "-o", str(exe)])
    env = dict(os.environ, LD_LIBRARY_PATH=libdir + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    out = subprocess.run([str(exe), "256"], env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    lines = [l for l in out.stdout.splitlines() if "retcode=" in l]
    assert len(lines) == 2 and all("retcode=1 " in l for l in lines), out.stdout
    assert "nsteps=3 " in lines[0] and "nfactors=3 " in lines[0]  # C2: 3 Newton steps, 3 factorisations


# ------------------------------------------------------------------ degenerate inputs
def test_gmres_zero_right_hand_side(nls):
    """b = 0: ‖r₀‖ = 0 ≤ atol + rtol‖r₀‖ — converged at once with x = 0, no Arnoldi step, nothing non-finite (the lagged
    normalisation must not divide by ‖b‖ = 0)."""
    p = R.Bratu2D(16)
    J = p.jac(0.1 * np.random.default_rng(5).standard_normal(p.n))
    for ortho in ("dcgs2", "cgs2", "mgs"):
        x, info = nls.GMRES(p.n, restart=20, ortho=ortho).set_operator(nls.CSRMatrix.from_scipy(J)).solve(
            np.zeros(p.n), abstol=0.0, reltol=1e-8, maxiters=100)
        xo, io = R.gmres(lambda z: J @ z, np.zeros(p.n), rtol=1e-8, restart=20, itmax=100)
        assert info["converged"] and not info["failed"] and info["iters"] == io.iters == 0
        assert np.all(x == 0.0) and np.all(xo == 0.0)


@pytest.mark.parametrize("algname", ["NewtonRaphson", "TrustRegion"])
def test_solver_started_at_the_root(nls, algname):
    """u0 is already a root: the first step solves J δ = 0, moves nowhere and the termination check reports Success — same
    step and residual counts as the oracle, u untouched."""
    root = np.sqrt(2.0) * np.ones(6)
    kw = dict(linsolve=None)
    ref = R.solve(R.Quadratic(6, 2.0), getattr(R, algname)(**kw), u0=root.copy(), abstol=1e-9)
    sol = nls.solve(nls.NonlinearProblem(nls.Quadratic(6, 2.0), u0=root.copy()), getattr(nls, algname)(**kw), abstol=1e-9)
    assert sol.retcode == R.RETCODE_NAMES[ref.retcode] == "Success"
    assert sol.stats.nsteps == ref.stats.nsteps and sol.stats.nf == ref.stats.nf
    assert np.max(np.abs(np.asarray(sol.u) - root)) <= 1e-12


def test_eisenstat_walker_state_is_reset_by_reinit_misc_tests_item6(nls):
    """lib/NonlinearSolveFirstOrder/test/misc_tests__item6.jl:9-17 on the device solver (the oracle's pin of the same name is
    tests/test_oracle_pins.py): η off η₀ after solve!, back on η₀ after reinit!(cache; p = 3.0), the re-solve ≈ √3 — and the η
    history of both solves equal to the oracle's."""
    ew = nls.EisenstatWalkerForcing2()
    prob = nls.NonlinearProblem(nls.Quadratic(2, 2.0), u0=np.array([1.0, 1.0]))
    c = nls.init(prob, nls.NewtonRaphson(linsolve=nls.KrylovJL_GMRES(), forcing=ew))
    oc = R.init(R.Quadratic(2, 2.0), R.NewtonRaphson(linsolve=R.KrylovJL_GMRES(), forcing=R.EisenstatWalkerForcing2()),
                u0=np.array([1.0, 1.0]))
    assert c.eta == ew.eta0
    sol, osol = c.solve(), oc.solve()
    assert sol.retcode == "Success" and sol.stats.nsteps == osol.stats.nsteps
    assert c.eta != ew.eta0 and abs(c.eta - oc.ew_eta) <= 1e-12 * max(1.0, abs(oc.ew_eta))
    c.reinit(p=3.0)
    oc.reinit(p=3.0)
    assert c.eta == ew.eta0 == oc.ew_eta
    sol2, osol2 = c.solve(), oc.solve()
    assert sol2.retcode == "Success" and np.allclose(np.asarray(sol2.u), np.sqrt(3.0))
    assert np.allclose(np.asarray(sol2.u), osol2.u, rtol=1e-12) and sol2.stats.nsteps == osol2.stats.nsteps
    assert abs(c.eta - oc.ew_eta) <= 1e-12 * max(1.0, abs(oc.ew_eta))
    c.close()
