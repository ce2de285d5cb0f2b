"""s-step GMRES (NK_ORTHO_SSTEP, csrc/nk_sstep.hip) — the block sweeps against NumPy, the solver against the oracle's
restatement (oracle.gmres_sstep) and against the column-by-column schemes it must agree with (same Krylov space, same
minimisation as Krylov.jl's gmres [EXT]: the iterates differ by rounding only), whole Newton solves, the fall-back when a
monomial block loses rank, and the full-size fixed-work protocol against the C oracle."""
import ctypes as C

import numpy as np
import pytest
import scipy.sparse as sp

from oracle import reference_restatement as R

pytestmark = pytest.mark.gpu


def _sweep(nls, mode, n, k, s, rng):
    from nonlinearsolve_jl_amd import _lib as L
    f = L.lib().nk_ss_sweep_test
    f.restype = C.c_int
    f.argtypes = [C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                  C.POINTER(C.c_double)]
    V = np.asfortranarray(rng.standard_normal((n, k + s)))
    U = rng.standard_normal((k, s)) * 0.1
    Rinv = np.triu(rng.standard_normal((s, s))) + 2 * np.eye(s)
    coef = np.concatenate([U.ravel(), Rinv.ravel()])
    V0, gram, us = V.copy(), np.zeros((k + s, s)), C.c_double(0)
    assert f(nls.default_context()._h, mode, n, k, s, V.ctypes.data, coef.ctypes.data, gram.ctypes.data, 0, C.byref(us)) == 0, \
        L.lib().nk_last_error()
    Wn = V0[:, k:] if mode == 0 else (V0[:, k:] - V0[:, :k] @ U) @ Rinv
    assert np.array_equal(V[:, :k], V0[:, :k])
    assert np.max(np.abs(V[:, k:] - Wn)) <= 1e-13 * np.max(np.abs(Wn))
    if mode != 2:
        gref = np.concatenate([V0[:, :k], Wn], axis=1).T @ Wn
        assert np.max(np.abs(gram - gref)) <= 1e-12 * np.max(np.abs(gref))


@pytest.mark.parametrize("n,k,s", [(1000, 3, 2), (5000, 1, 6), (70001, 17, 5), (4096, 30, 6), (300, 40, 8), (257, 7, 1),
                                   (10000, 60, 3), (256, 10, 6), (1, 1, 1), (513, 26, 6), (100000, 42, 6)])
def test_block_sweeps_match_numpy(nls, n, k, s):
    """Sweep A ([V X]ᵀX on the matrix cores), sweep B (X ← (X − V U)R⁻¹, then the Gram block of the result), sweep C (update
    only): every register-resident size class (k + s ≤ 16, 32, 48), the streaming class, ragged last tiles, one row."""
    rng = np.random.default_rng(n + k)
    for mode in (0, 1, 2):
        _sweep(nls, mode, n, k, s, rng)


def _pair(nls, dev, which):
    import torch
    if which == "bratu":
        P, PD = R.Bratu2D(48), nls.Bratu2D(48)
    else:
        P, PD = R.Brusselator2D(24), nls.Brusselator2D(24)
    u = P.u0() + 0.1 * np.sin(np.arange(P.n) * 0.37)
    A, b = P.jac(u).tocsr(), P.f(u)
    J = PD.jac_csr()
    PD.jac_values(torch.tensor(u, device=dev), J)
    return P, PD, u, A, b, J


@pytest.mark.parametrize("which", ["bratu", "brusselator"])
@pytest.mark.parametrize("s", [1, 2, 3, 5, 6, 7, 8])
def test_sstep_gmres_matches_oracle_and_cgs2(nls, dev, which, s):
    import torch
    P, PD, u, A, b, J = _pair(nls, dev, which)
    xr, ir = R.gmres(lambda z: A @ z, b, restart=30, fixed_iters=30, ortho="cgs2")
    xo, io = R.gmres(lambda z: A @ z, b, restart=30, fixed_iters=30, ortho=("sstep", s))
    G = nls.GMRES(P.n, restart=30, ortho="sstep", sstep=s).set_operator(J)
    x, gi = G.solve(torch.tensor(b, device=dev), fixed_iters=30)
    x = x.cpu().numpy()
    tol = 1e-12 if s <= 5 else 2e-10          # κ of the monomial block grows with s; the block form keeps O(ε κ²) out of x
    assert gi["iters"] == 30 == io.iters
    assert np.linalg.norm(x - xo) <= tol * np.linalg.norm(xo) and np.linalg.norm(x - xr) <= tol * np.linalg.norm(xr)
    assert abs(gi["rnorm"] - ir.rnorm) <= 1e-9 * ir.rnorm
    assert abs(np.linalg.norm(b - A @ x) - gi["rnorm"]) <= 1e-9 * gi["rnorm0"]   # the recurrence residual is the true one


@pytest.mark.parametrize("which", ["bratu", "brusselator"])
def test_sstep_restarts_tolerance_and_ragged_last_block(nls, dev, which):
    """Restart cycles (r = b − A x into column 0), stopping inside a block (the columns of a block are tested together,
    the solution keeps the columns up to the one that met the tolerance), restart lengths that are no multiple of s."""
    import torch
    P, PD, u, A, b, J = _pair(nls, dev, which)
    bd = torch.tensor(b, device=dev)
    for m, s, rtol in ((30, 6, 1e-6), (20, 6, 1e-8), (7, 3, 1e-5), (30, 4, 1e-10)):
        xo, io = R.gmres(lambda z: A @ z, b, restart=m, rtol=rtol, itmax=400, ortho=("sstep", s))
        G = nls.GMRES(P.n, restart=m, ortho="sstep", sstep=s).set_operator(J)
        x, gi = G.solve(bd, abstol=0.0, reltol=rtol, maxiters=400)
        x = x.cpu().numpy()
        assert gi["iters"] == io.iters and bool(gi["converged"]) == io.converged and gi["restarts"] == io.restarts
        assert np.linalg.norm(x - xo) <= 1e-8 * np.linalg.norm(xo)
        if io.converged:
            assert np.linalg.norm(b - A @ x) <= 1.001 * rtol * np.linalg.norm(b) + 1e-12


def test_sstep_with_preconditioners_and_matrix_free(nls, dev):
    """The matrix powers go through the same operator hook as the column-by-column schemes: right preconditioners (Chebyshev
    polynomial, multigrid V-cycle) and the matrix-free JVP."""
    import torch
    P, PD = R.Bratu2D(64), nls.Bratu2D(64)
    u = P.u0() + 0.1 * np.sin(np.arange(P.n) * 0.37)
    A, b = P.jac(u).tocsr(), P.f(u)
    ud, bd = torch.tensor(u, device=dev), torch.tensor(b, device=dev)
    op = nls.StatefulJacobianOperator(nls.JacobianOperator(nls.NonlinearProblem(PD)), ud)
    xr, ir = R.gmres(lambda z: A @ z, b, restart=30, fixed_iters=30, ortho="cgs2")
    G = nls.GMRES(P.n, restart=30, ortho="sstep").set_operator(op)
    x, gi = G.solve(bd, fixed_iters=30)
    assert np.linalg.norm(x.cpu().numpy() - xr) <= 1e-10 * np.linalg.norm(xr)
    M = R.BratuMultigrid(P, u, 2, 8)
    xm, im = R.gmres(lambda z: A @ z, b, rtol=1e-10, restart=30, itmax=100, M=M, ortho="cgs2")
    Gm = nls.GMRES(P.n, restart=30, ortho="sstep", sstep=4).set_operator(op)
    Gm.set_multigrid_preconditioner(PD, ud, nu=2, coarse_max=8)
    x2, g2 = Gm.solve(bd, abstol=0.0, reltol=1e-10, maxiters=100)
    assert g2["converged"] and np.linalg.norm(x2.cpu().numpy() - xm) <= 1e-8 * np.linalg.norm(xm)
    assert abs(g2["iters"] - im.iters) <= 4          # the stopping test sees whole blocks


def test_rank_deficient_block_falls_back(nls, dev):
    """J = 2I: the monomial block [Av, A²v, …] has rank one, the Pythagorean Gram block is singular and its Cholesky
    factorisation breaks down — the solve is redone column by column (where β = 0 is the lucky breakdown) and succeeds."""
    import torch
    prob = nls.NonlinearProblem(nls.Quadratic(50, 2.0))
    sol = nls.solve(prob, nls.NewtonRaphson(linsolve=nls.KrylovJL_GMRES(ortho="sstep")), abstol=1e-10)
    ref = R.solve(R.Quadratic(50, 2.0), R.NewtonRaphson(linsolve=R.KrylovJL_GMRES()), abstol=1e-10)
    assert sol.retcode == "Success" and sol.stats.nsteps == ref.stats.nsteps
    assert np.max(np.abs(np.asarray(sol.u) - ref.u)) <= 1e-12
    with pytest.raises(R.SStepBreakdown):
        R.gmres(lambda z: 2.0 * z, np.ones(5), restart=5, ortho="sstep")


@pytest.mark.parametrize("case", ["bratu_nr_ew", "bratu_tr", "brusselator_tr_concrete", "bratu_lm"])
def test_newton_solves_with_sstep_take_the_reference_path(nls, case):
    """Whole solves with the s-step linear solver: step counts, accept/reject sequences and iterates of the oracle run with
    its column-by-column GMRES (tolerances far above the 1e-12 the two linear solvers differ by)."""
    kw = dict(gmres_restart=30, maxiters=300)
    if case == "bratu_nr_ew":
        rp, dp = R.Bratu2D(32), nls.Bratu2D(32)
        ra = R.NewtonRaphson(linsolve=R.KrylovJL_GMRES(**kw), forcing=R.EisenstatWalkerForcing2())
        da = nls.NewtonRaphson(linsolve=nls.KrylovJL_GMRES(ortho="sstep", **kw), forcing=nls.EisenstatWalkerForcing2())
    elif case == "bratu_tr":
        rp, dp = R.Bratu2D(32), nls.Bratu2D(32)
        ra = R.TrustRegion(linsolve=R.KrylovJL_GMRES(**kw))
        da = nls.TrustRegion(linsolve=nls.KrylovJL_GMRES(ortho="sstep", sstep=5, **kw))
    elif case == "brusselator_tr_concrete":
        rp, dp = R.Brusselator2D(16), nls.Brusselator2D(16)
        ra = R.TrustRegion(linsolve=R.KrylovJL_GMRES(**kw), concrete_jac=True)
        da = nls.TrustRegion(linsolve=nls.KrylovJL_GMRES(ortho="sstep", **kw), concrete_jac=True)
    else:
        rp, dp = R.Bratu2D(12), nls.Bratu2D(12)
        ra = R.LevenbergMarquardt(linsolve=R.KrylovJL_GMRES(gmres_restart=60, maxiters=600))
        da = nls.LevenbergMarquardt(linsolve=nls.KrylovJL_GMRES(ortho="sstep", gmres_restart=60, maxiters=600))
    ref = R.solve(rp, ra, abstol=1e-8, maxiters=100)
    sol = nls.solve(nls.NonlinearProblem(dp), da, abstol=1e-8, maxiters=100, store_trace=True)
    assert sol.retcode == "Success" == R.RETCODE_NAMES[ref.retcode]
    assert sol.stats.nsteps == ref.stats.nsteps
    assert [t["accepted"] for t in sol.trace] == [t["accepted"] for t in ref.trace]
    assert np.max(np.abs(np.asarray(sol.u) - ref.u)) <= 1e-7 * max(1.0, np.max(np.abs(ref.u)))


def test_c3_fixed_work_with_sstep_vs_c_oracle(nls, dev):
    """The headline protocol at full size (Bratu 1024², 30 Arnoldi steps per Newton step) with s = 6: ‖F‖∞ after every step and
    the iterate of the C oracle's MGS run, and of the device's own delayed-CGS2 run."""
    import torch
    from oracle import c_oracle as CO
    ns = 1024
    n = ns * ns
    traces, us = [], []
    for ortho in ("sstep", "dcgs2"):
        u0 = torch.zeros(n, dtype=torch.float64, device=dev)
        cache = nls.init(nls.NonlinearProblem(nls.Bratu2D(ns, 6.0), u0=u0),
                         nls.NewtonRaphson(linsolve=nls.KrylovJL_GMRES(fixed_iters=30, maxiters=30, ortho=ortho), concrete_jac=True),
                         abstol=1e-300, maxiters=100, store_trace=True)
        for _ in range(4):
            cache.step()
        traces.append(np.array([t["fnorm_inf"] for t in cache.trace]))
        us.append(cache.u.cpu().numpy())
        cache.close()
    uC, fnC, giC, _ = CO.bratu_newton(ns, 6.0, 0.0, np.zeros(n), 4, use_csr=True, m=30, itmax=30, fixed_iters=30, forcing=False)
    assert np.allclose(traces[0], fnC, rtol=1e-6) and np.max(np.abs(us[0] - uC)) <= 1e-9
    assert np.allclose(traces[0], traces[1], rtol=1e-9) and np.max(np.abs(us[0] - us[1])) <= 1e-11
    # … and of the C oracle's own s-step restatement (oracle/nk_oracle.c::orc_bratu_newton_fast_sstep): the same arithmetic
    # at full size, to rounding
    uS, fnS, _ = CO.bratu_newton_fast_sstep(ns, 6.0, 0.0, np.zeros(n), 4, use_csr=True, m=30, s=6)
    assert np.allclose(traces[0], fnS, rtol=1e-9) and np.max(np.abs(us[0] - uS)) <= 1e-10


def test_sstep_with_callable_operator_and_chebyshev(nls, dev):
    """Operators reached through callbacks do not fuse the 1/σ scaling (the vector is scaled first, then handed over), and the
    Chebyshev polynomial preconditioner sits inside every matrix power: both against the oracle's column-by-column GMRES."""
    import torch
    P = R.Bratu2D(40)
    u = P.u0() + 0.1 * np.sin(np.arange(P.n) * 0.37)
    A, b = P.jac(u).tocsr(), P.f(u)
    Ad = torch.sparse_csr_tensor(torch.tensor(A.indptr, dtype=torch.int64), torch.tensor(A.indices, dtype=torch.int64),
                                 torch.tensor(A.data), size=A.shape).to(dev)
    bd = torch.tensor(b, device=dev)
    xr, ir = R.gmres(lambda z: A @ z, b, restart=24, fixed_iters=24, ortho="cgs2")
    G = nls.GMRES(P.n, restart=24, ortho="sstep", sstep=4).set_operator(lambda x: Ad @ x)
    x, gi = G.solve(bd, fixed_iters=24)
    assert gi["iters"] == 24 and np.linalg.norm(x.cpu().numpy() - xr) <= 1e-10 * np.linalg.norm(xr)
    PD = nls.Bratu2D(40)
    J = PD.jac_csr()
    PD.jac_values(torch.tensor(u, device=dev), J)
    lmax = R.gershgorin_lambda(A)
    M = R.chebyshev_preconditioner(lambda v: A @ v, lmax / 30.0, lmax, 8)
    xc, ic = R.gmres(lambda z: A @ z, b, rtol=1e-9, restart=30, itmax=200, M=M, ortho="cgs2")
    Gc = nls.GMRES(P.n, restart=30, ortho="sstep", sstep=5).set_operator(J)
    Gc.set_chebyshev_preconditioner(8, lmax / 30.0, lmax)
    x2, g2 = Gc.solve(bd, abstol=0.0, reltol=1e-9, maxiters=200)
    assert g2["converged"] and np.linalg.norm(x2.cpu().numpy() - xc) <= 1e-7 * np.linalg.norm(xc)
    assert abs(g2["iters"] - ic.iters) <= 5


def test_c5_fixed_work_with_sstep_equals_column_form(nls, dev):
    """Config C5 at full size (Brusselator 512², TrustRegion, coloured concrete J, 30 Arnoldi steps per step): the s-step and
    the delayed-CGS2 runs accept/reject alike and keep the same iterate."""
    outs = []
    for ortho in ("sstep", "dcgs2"):
        PB = nls.Brusselator2D(512)
        cache = nls.init(nls.NonlinearProblem(PB, u0=PB.initial_guess(device=True)),
                         nls.TrustRegion(linsolve=nls.KrylovJL_GMRES(fixed_iters=30, maxiters=30, ortho=ortho), concrete_jac=True,
                                         jac_colored=True), abstol=1e-300, maxiters=100, store_trace=True)
        for _ in range(4):
            cache.step()
        outs.append(([t["accepted"] for t in cache.trace], np.array([t["fnorm_inf"] for t in cache.trace]), cache.u.cpu().numpy()))
        cache.close()
    assert outs[0][0] == outs[1][0]
    assert np.allclose(outs[0][1], outs[1][1], rtol=1e-7)
    assert np.max(np.abs(outs[0][2] - outs[1][2])) <= 1e-8 * np.max(np.abs(outs[1][2]))


def test_block_size_is_validated(nls):
    with pytest.raises(nls.NKError, match="block size"):
        nls.GMRES(100, restart=30, ortho="sstep", sstep=9)
    with pytest.raises(nls.NKError, match="block size"):
        nls.GMRES(100, restart=30, ortho="sstep", sstep=0)
