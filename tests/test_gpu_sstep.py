"""s-step GMRES (NK_ORTHO_SSTEP, csrc/nk_sstep.hip; the library's default Arnoldi process) — the block sweeps against NumPy,
the solver against the oracle's restatement (oracle.gmres_sstep: monomial AND Newton basis) and against the column-by-column
schemes it must agree with (same Krylov space, same minimisation as Krylov.jl's gmres [EXT]: the iterates differ by rounding
only), whole Newton solves, the fall-back when a block loses rank (first cycle, and a later cycle under a preconditioner), and
the full-size fixed-work protocol against the C oracle over 20 Newton steps."""
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
    Rup = np.triu(rng.standard_normal((s, s))) * 0.3 + 2 * np.eye(s)    # the block's triangular factor R
    Rinv = np.linalg.inv(Rup)
    coef = np.concatenate([U.ravel(), Rup.ravel()])
    V0, gram, us = V.copy(), np.zeros((k + s, s)), C.c_double(0)
    assert f(nls.default_context()._h, mode, n, k, s, V.ctypes.data, coef.ctypes.data, gram.ctypes.data, 0, C.byref(us)) == 0, \
        L.lib().nk_last_error()
    Wn = V0[:, k:] if mode == 0 else (V0[:, k:] - V0[:, :k] @ U) @ Rinv
    assert np.array_equal(V[:, :k], V0[:, :k])
    assert np.max(np.abs(V[:, k:] - Wn)) <= 1e-13 * np.max(np.abs(Wn))
    if mode != 2:
        gref = np.concatenate([V0[:, :k], Wn], axis=1).T @ Wn
        assert np.max(np.abs(gram - gref)) <= 1e-12 * np.max(np.abs(gref))


@pytest.mark.parametrize("n,k,s", [(1000, 3, 2), (5000, 1, 6), (70001, 17, 4), (4096, 30, 6), (300, 40, 8), (257, 7, 1),
                                   (10000, 60, 2), (256, 10, 6), (1, 1, 1), (513, 26, 6), (100000, 42, 6),
                                   (5000, 1, 15), (70001, 16, 15), (4096, 31, 15), (513, 21, 8), (300, 11, 8), (20000, 5, 8),
                                   (9999, 33, 8), (100000, 16, 15), (257, 1, 4), (3000, 52, 8)])   # (the compiled widths: 1, 2, 4, 6, 8, 15)
def test_block_sweeps_match_numpy(nls, n, k, s):
    """Sweep A ([V X]ᵀX on the matrix cores), sweep B (X ← (X − V U)R⁻¹, then the Gram block of the result), sweep C (update
    only): every register-resident size class (k + s ≤ 16, 32, 48), the streaming class, ragged last tiles, one row."""
    rng = np.random.default_rng(n + k)
    for mode in (0, 1, 2):
        _sweep(nls, mode, n, k, s, rng)


@pytest.mark.parametrize("n,k", [(4096, 16), (70002, 16), (70001, 16), (300, 16), (65538, 16), (2, 16), (1 << 20, 16), (262144 + 64 + 6, 16),
                                 (254, 16), (256, 16), (258, 16), (32, 16), (34, 16), (130, 16), (131072 + 128 + 30, 16), (7 * 256 * 256 + 98, 16),
                                 (100000, 1), (5001, 1)])
def test_sweep_b_that_stores_nothing(nls, n, k):
    """The cycle's last block: sweep B leaves the columns as the matrix powers wrote them and returns the Gram block of the update
    it formed in registers. Behind 16 columns with an even leading dimension that is the read-only kernel (operands loaded in the
    matrix instruction's layout, rows of a tile in a different order, only the basis columns staged in LDS): full tiles, ragged
    tiles inside a wavefront's row pair block and between wavefronts, fewer rows than a wavefront; an odd leading dimension and
    the block behind one column take the staging kernel."""
    from nonlinearsolve_jl_amd import _lib as L
    f = L.lib().nk_ss_sweep_test
    f.restype = C.c_int
    f.argtypes = [C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                  C.POINTER(C.c_double)]
    rng = np.random.default_rng(n + k)
    s = 15
    V = np.asfortranarray(rng.standard_normal((n, k + s)))
    U = rng.standard_normal((k, s)) * 0.1
    Rup = np.triu(rng.standard_normal((s, s))) * 0.3 + 2 * np.eye(s)
    coef = np.concatenate([U.ravel(), Rup.ravel()])
    V0, gram, us = V.copy(), np.zeros((k + s, s)), C.c_double(0)
    assert f(nls.default_context()._h, 3, n, k, s, V.ctypes.data, coef.ctypes.data, gram.ctypes.data, 0, C.byref(us)) == 0, \
        L.lib().nk_last_error()
    assert np.array_equal(V, V0)
    Wn = (V0[:, k:] - V0[:, :k] @ U) @ np.linalg.inv(Rup)
    gref = np.concatenate([V0[:, :k], Wn], axis=1).T @ Wn
    assert np.max(np.abs(gram - gref)) <= 1e-12 * np.max(np.abs(gref))


@pytest.mark.parametrize("k", [1, 16, 7])
def test_update_sweep_with_an_ill_conditioned_factor(nls, k):
    """X ← (X − V U) R⁻¹ with κ(R) = 1e5 (pivot ratio 1e-10: two decades above the rank-loss bar): the matrix-core form of the
    default cycle's shapes (k = 1, 16: one product with [−U N ; N], N = R⁻¹ explicit) and the substitution form (k = 7) both stay
    within a few ε κ(R) of the float64 reference — what pass 2 of the block scheme then repairs."""
    from nonlinearsolve_jl_amd import _lib as L
    f = L.lib().nk_ss_sweep_test
    f.restype = C.c_int
    f.argtypes = [C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                  C.POINTER(C.c_double)]
    rng = np.random.default_rng(k)
    n, s = 30000, 15
    V = np.asfortranarray(rng.standard_normal((n, k + s)))
    U = rng.standard_normal((k, s)) * 0.1
    Q, _ = np.linalg.qr(rng.standard_normal((s, s)))
    Rup = np.linalg.qr(Q @ np.diag(np.logspace(0, -5, s)) @ Q.T)[1]
    Rup = Rup * np.sign(np.diag(Rup))[:, None]                      # positive diagonal, κ₂(R) = 1e5
    coef = np.concatenate([U.ravel(), Rup.ravel()])
    V0, gram, us = V.copy(), np.zeros((k + s, s)), C.c_double(0)
    assert f(nls.default_context()._h, 1, n, k, s, V.ctypes.data, coef.ctypes.data, gram.ctypes.data, 0, C.byref(us)) == 0, \
        L.lib().nk_last_error()
    W = V0[:, k:] - V0[:, :k] @ U
    ref = np.linalg.solve(Rup.T, W.T).T
    err = np.max(np.abs(V[:, k:] - ref)) / np.max(np.abs(ref))
    assert err <= 4 * np.finfo(float).eps * 1e5, err
    gref = np.concatenate([V0[:, :k], V[:, k:]], axis=1).T @ V[:, k:]      # the Gram block of what the sweep actually wrote
    assert np.max(np.abs(gram - gref)) <= 1e-12 * np.max(np.abs(gref))


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
    G = nls.GMRES(P.n, restart=30, ortho="sstep", sstep=s, sstep_basis="monomial").set_operator(J)
    x, gi = G.solve(torch.tensor(b, device=dev), fixed_iters=30)
    x = x.cpu().numpy()
    tol = 1e-12 if s <= 5 else 2e-10          # κ of the monomial block grows with s; the block form keeps O(ε κ²) out of x
    assert gi["iters"] == 30 == io.iters
    assert np.linalg.norm(x - xo) <= tol * np.linalg.norm(xo) and np.linalg.norm(x - xr) <= tol * np.linalg.norm(xr)
    assert abs(gi["rnorm"] - ir.rnorm) <= 1e-9 * ir.rnorm
    assert abs(np.linalg.norm(b - A @ x) - gi["rnorm"]) <= 1e-9 * gi["rnorm0"]   # the recurrence residual is the true one


@pytest.mark.parametrize("which", ["bratu", "brusselator"])
@pytest.mark.parametrize("s", [0, 4, 6, 8, 10, 12, 13, 15])
def test_sstep_newton_basis_matches_oracle_and_cgs2(nls, dev, which, s):
    """The Newton basis (Leja-ordered Chebyshev shifts on the Gershgorin interval of the CSR operator — what the library picks
    by itself for a concrete J; s = 0: its own block size, 15): the device against the oracle's restatement of the same
    arithmetic with the interval the ORACLE computes from the matrix, and against CGS2. s = 13 is cut into 8 + 5."""
    import torch
    P, PD, u, A, b, J = _pair(nls, dev, which)
    iv = R.gershgorin_interval(A)
    xr, ir = R.gmres(lambda z: A @ z, b, restart=30, fixed_iters=30, ortho="cgs2")
    xo, io = R.gmres(lambda z: A @ z, b, restart=30, fixed_iters=30, ortho=("sstep", s, "newton", iv))
    G = nls.GMRES(P.n, restart=30, ortho="sstep", sstep=s).set_operator(J)   # sstep_basis = "auto" → Newton for a CSR operator
    x, gi = G.solve(torch.tensor(b, device=dev), fixed_iters=30)
    x = x.cpu().numpy()
    assert gi["iters"] == 30 == io.iters
    assert np.linalg.norm(x - xo) <= 2e-10 * np.linalg.norm(xo) and np.linalg.norm(x - xr) <= 2e-10 * np.linalg.norm(xr)
    assert abs(gi["rnorm"] - ir.rnorm) <= 1e-8 * ir.rnorm
    assert abs(np.linalg.norm(b - A @ x) - gi["rnorm"]) <= 1e-9 * gi["rnorm0"]
    # the same bounds handed over by the caller (the route for callback / matrix-free operators): identical iterate
    Ad = torch.sparse_csr_tensor(torch.tensor(A.indptr, dtype=torch.int64), torch.tensor(A.indices, dtype=torch.int64),
                                 torch.tensor(A.data), size=A.shape).to(dev)
    Gc = nls.GMRES(P.n, restart=30, ortho="sstep", sstep=s, sstep_basis="newton").set_operator(lambda z: Ad @ z)
    Gc.set_spectrum_interval(*iv)
    xc, gc = Gc.solve(torch.tensor(b, device=dev), fixed_iters=30)
    assert np.linalg.norm(xc.cpu().numpy() - xo) <= 2e-10 * np.linalg.norm(xo)
    with pytest.raises(nls.NKError, match="bounds"):   # insisting on the Newton basis without bounds is an error, not a guess
        nls.GMRES(P.n, restart=30, ortho="sstep", sstep_basis="newton").set_operator(lambda z: Ad @ z).solve(
            torch.tensor(b, device=dev), fixed_iters=30)


def test_sstep_block_left_at_its_first_pass_only_if_that_pass_was_good(nls, dev):
    """The implicit second pass and its limit: with a smooth right-hand side (u = 0) and bounds 1.3× too wide a block
    of 15 is still left at its first pass (departure 0.01) and the device reproduces the oracle; at 1.5× the first pass is 0.75
    away from orthonormal — the oracle raises, the device counts a broken block, narrows 15 → 8 and returns the CGS2 iterate."""
    import torch
    P = R.Bratu2D(24)
    u = P.u0()
    A, b = P.jac(u).tocsr(), P.f(u)
    lo, hi = R.gershgorin_interval(A)
    Ad = torch.sparse_csr_tensor(torch.tensor(A.indptr, dtype=torch.int64), torch.tensor(A.indices, dtype=torch.int64),
                                 torch.tensor(A.data), size=A.shape).to(dev)
    bd = torch.tensor(b, device=dev)
    xr, _ = R.gmres(lambda z: A @ z, b, restart=30, fixed_iters=30, ortho="cgs2")
    for f, accepted in ((0.0, True), (0.15, True), (0.25, False)):
        iv = (lo - f * (hi - lo), hi + f * (hi - lo))
        G = nls.GMRES(P.n, restart=30, ortho="sstep", sstep_basis="newton").set_operator(lambda z: Ad @ z)
        G.set_spectrum_interval(*iv)
        x, gi = G.solve(bd, fixed_iters=30)
        x = x.cpu().numpy()
        bs, newton, broken = G.sstep_state()
        assert gi["iters"] == 30 and newton
        assert np.linalg.norm(x - xr) <= 1e-8 * np.linalg.norm(xr)
        if accepted:
            xo, _ = R.gmres(lambda z: A @ z, b, restart=30, fixed_iters=30, ortho=("sstep", 15, "newton", iv))
            assert (bs, broken) == (15, 0) and np.linalg.norm(x - xo) <= 1e-9 * np.linalg.norm(xo)
        else:
            with pytest.raises(R.SStepBreakdown, match="first pass"):
                R.gmres(lambda z: A @ z, b, restart=30, fixed_iters=30, ortho=("sstep", 15, "newton", iv))
            x8, _ = R.gmres(lambda z: A @ z, b, restart=30, fixed_iters=30, ortho=("sstep", 8, "newton", iv))
            assert (bs, broken) == (8, 1) and np.linalg.norm(x - x8) <= 1e-9 * np.linalg.norm(x8)


def test_leja_nodes_and_interval_match_the_oracle(nls, dev):
    import ctypes as C
    from nonlinearsolve_jl_amd import _lib as L
    from oracle import c_oracle as CO
    f = L.lib().nk_ss_leja_nodes
    f.argtypes = [C.c_int, C.POINTER(C.c_double)]
    for s in range(1, 17):
        out = (C.c_double * 16)()
        assert f(s, out) == 0
        assert np.array_equal(np.array(out[:s]), R.leja_chebyshev_nodes(s))
        if s < 16:
            assert np.array_equal(np.array(out[:s]), CO.leja_nodes(s))


@pytest.mark.parametrize("which", ["bratu", "brusselator"])
def test_sstep_restarts_tolerance_and_ragged_last_block(nls, dev, which):
    """Restart cycles (r = b − A x into column 0), stopping inside a block (the columns of a block are tested together,
    the solution keeps the columns up to the one that met the tolerance), restart lengths that are no multiple of s."""
    import torch
    P, PD, u, A, b, J = _pair(nls, dev, which)
    bd = torch.tensor(b, device=dev)
    for m, s, rtol in ((30, 6, 1e-6), (20, 6, 1e-8), (7, 3, 1e-5), (30, 4, 1e-10)):
        for basis in ("monomial", "newton"):
            if basis == "newton" and s < 6:
                s = {3: 7, 4: 15}[s]   # (m, s) = (7, 7): one block per cycle; (30, 15): two
            ortho = ("sstep", s) if basis == "monomial" else ("sstep", s, "newton", R.gershgorin_interval(A))
            xo, io = R.gmres(lambda z: A @ z, b, restart=m, rtol=rtol, itmax=400, ortho=ortho)
            G = nls.GMRES(P.n, restart=m, ortho="sstep", sstep=s, sstep_basis=basis).set_operator(J)
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
    G = nls.GMRES(P.n, restart=30, ortho="sstep").set_operator(op)   # Newton basis from the stencil's closed-form discs, s = 15
    x, gi = G.solve(bd, fixed_iters=30)
    assert np.linalg.norm(x.cpu().numpy() - xr) <= 1e-10 * np.linalg.norm(xr)
    d_ = P.c_exp * np.exp(u)
    xo, _ = R.gmres(lambda z: A @ z, b, restart=30, fixed_iters=30,
                    ortho=("sstep", 0, "newton", (-float(d_.max()), 8.0 * P.c_lap - float(d_.min()))))
    assert np.linalg.norm(x.cpu().numpy() - xo) <= 1e-11 * np.linalg.norm(xo)
    Gmono = nls.GMRES(P.n, restart=30, ortho="sstep", sstep_basis="monomial").set_operator(op)
    x6, _ = Gmono.solve(bd, fixed_iters=30)
    assert np.linalg.norm(x6.cpu().numpy() - xr) <= 1e-10 * np.linalg.norm(xr)
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


@pytest.mark.parametrize("prec", ["chebyshev", "multigrid", "none"])
@pytest.mark.parametrize("cycle", [0, 1, 2])
def test_breakdown_in_a_later_cycle_finishes_column_by_column(nls, dev, prec, cycle):
    """A block that loses rank in restart cycle ≥ 1, with a right preconditioner in place: that cycle is discarded, the rest of
    the solve runs with delayed CGS2 from the same restart loop (r = b − A x through the raw operator), the tolerance stays the
    one of the first cycle, the iteration count excludes the discarded columns — and the NEXT solve uses the s-step form again.
    (The breakdown is injected by the development hook nk_gmres_debug_force_breakdown: no matrix breaks down on demand.)"""
    import torch
    from nonlinearsolve_jl_amd import _lib as L
    P, PD = R.Bratu2D(48), nls.Bratu2D(48)
    u = P.u0() + 0.1 * np.sin(np.arange(P.n) * 0.37)
    A, b = P.jac(u).tocsr(), P.f(u)
    ud, bd = torch.tensor(u, device=dev), torch.tensor(b, device=dev)
    J = PD.jac_csr()
    PD.jac_values(ud, J)
    # (the V-cycle converges in a handful of iterations: short cycles; unpreconditioned GMRES needs long ones to converge at all)
    m, sblk, rtol, cap = {"multigrid": (2, 2, 1e-9, 600), "chebyshev": (6, 3, 1e-9, 600), "none": (20, 6, 1e-5, 4000)}[prec]
    G = nls.GMRES(P.n, restart=m, ortho="sstep", sstep=sblk).set_operator(J)
    M = None
    if prec == "chebyshev":
        lmax = R.gershgorin_lambda(A)
        G.set_chebyshev_preconditioner(3, lmax / 30.0, lmax)
        M = R.chebyshev_preconditioner(lambda v: A @ v, lmax / 30.0, lmax, 3)
    elif prec == "multigrid":
        G.set_multigrid_preconditioner(PD, ud, nu=1, coarse_max=12)
        M = R.BratuMultigrid(P, u, 1, 12)
    x0, g0 = G.solve(bd, abstol=0.0, reltol=rtol, maxiters=cap)
    assert g0["converged"] and g0["restarts"] >= 3, g0
    f = L.lib().nk_gmres_debug_force_breakdown
    f.argtypes = [C.c_void_p, C.c_int]
    assert f(G._h, cycle) == 0
    x1, g1 = G.solve(bd, abstol=0.0, reltol=rtol, maxiters=cap)
    assert g1["converged"] and not g1["failed"]
    x1 = x1.cpu().numpy()
    assert np.linalg.norm(b - A @ x1) <= 1.001 * rtol * np.linalg.norm(b)          # the FIRST cycle's tolerance, not a re-based one
    # (two different Krylov paths to the same residual level: the iterates agree to rtol·κ(J))
    assert np.linalg.norm(x1 - x0.cpu().numpy()) <= (1e-6 if rtol <= 1e-9 else 0.05) * np.linalg.norm(x1)
    # the oracle's column-by-column GMRES takes the same number of iterations (the discarded cycle is not counted)
    xo, io = R.gmres(lambda z: A @ z, b, rtol=rtol, restart=m, itmax=cap, M=M, ortho="cgs2")
    assert abs(g1["iters"] - io.iters) <= m
    assert f(G._h, -1) == 0
    x2, g2 = G.solve(bd, abstol=0.0, reltol=rtol, maxiters=cap)                      # the s-step form is back
    assert g2["iters"] == g0["iters"] and np.array_equal(x2.cpu().numpy(), x0.cpu().numpy())


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
    """The headline protocol at full size (Bratu 1024², 30 Arnoldi steps per Newton step) over 20 NEWTON STEPS with the library
    default (s-step, Newton basis, s = 15): ‖F‖∞ after every step and the iterate against the C oracle's restatement of the
    same arithmetic (oracle/nk_oracle.c::orc_bratu_newton_fast_sstep2), against its column-by-column delayed-CGS2 leg, against
    its MGS run (4 steps), and against the device's own delayed-CGS2 and monomial s = 6 runs."""
    import torch
    from oracle import c_oracle as CO
    ns, nst = 1024, 20
    n = ns * ns
    traces, us = {}, {}
    for name, kw in (("default", {}), ("dcgs2", dict(ortho="dcgs2")), ("mono6", dict(ortho="sstep", sstep=6, sstep_basis="monomial"))):
        u0 = torch.zeros(n, dtype=torch.float64, device=dev)
        cache = nls.init(nls.NonlinearProblem(nls.Bratu2D(ns, 6.0), u0=u0),
                         nls.NewtonRaphson(linsolve=nls.KrylovJL_GMRES(fixed_iters=30, maxiters=30, **kw), concrete_jac=True),
                         abstol=1e-300, maxiters=100, store_trace=True)
        for _ in range(nst):
            cache.step()
        traces[name] = np.array([t["fnorm_inf"] for t in cache.trace])
        us[name] = cache.u.cpu().numpy()
        cache.close()
    uS, fnS, _ = CO.bratu_newton_fast_sstep(ns, 6.0, 0.0, np.zeros(n), nst, use_csr=True, m=30, s=15, basis="newton")
    assert np.allclose(traces["default"], fnS, rtol=1e-8) and np.max(np.abs(us["default"] - uS)) <= 1e-10
    uD, fnD, _ = CO.bratu_newton_fast(ns, 6.0, 0.0, np.zeros(n), nst, use_csr=True, m=30)
    assert np.allclose(traces["default"], fnD, rtol=1e-8) and np.max(np.abs(us["default"] - uD)) <= 1e-10
    assert np.allclose(traces["default"], traces["dcgs2"], rtol=1e-8) and np.max(np.abs(us["default"] - us["dcgs2"])) <= 1e-10
    assert np.allclose(traces["default"], traces["mono6"], rtol=1e-8) and np.max(np.abs(us["default"] - us["mono6"])) <= 1e-10
    uC, fnC, giC, _ = CO.bratu_newton(ns, 6.0, 0.0, np.zeros(n), 4, use_csr=True, m=30, itmax=30, fixed_iters=30, forcing=False)
    assert np.allclose(traces["default"][:4], fnC, rtol=1e-6)
    uM, fnM, _ = CO.bratu_newton_fast_sstep(ns, 6.0, 0.0, np.zeros(n), 4, use_csr=True, m=30, s=6)
    assert np.allclose(traces["mono6"][:4], fnM, rtol=1e-9)


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
    G = nls.GMRES(P.n, restart=24, ortho="sstep", sstep=4).set_operator(lambda x: Ad @ x)   # no bounds known: monomial
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
        nls.GMRES(100, restart=30, ortho="sstep", sstep=17)
    with pytest.raises(nls.NKError, match="block size"):
        nls.GMRES(100, restart=30, ortho="sstep", sstep=-1)
    nls.GMRES(100, restart=30, ortho="sstep", sstep=0)   # automatic


def _convection_diffusion(n, peclet, scale_rows=False):
    """−Δu + (β·∇)u on an n × n grid, first-order upwind: a nonsymmetric M-matrix whose spectrum leaves the real axis as the
    cell Péclet number grows; optionally with rows scaled over two decades (Gershgorin bounds far wider than the bulk of the
    spectrum; restarted GMRES(30) then needs ≈ 1000 iterations)."""
    import scipy.sparse as sp
    h = 1.0 / (n + 1)
    e = np.ones(n)
    T = sp.diags([-e[:-1], 2 * e, -e[:-1]], [-1, 0, 1]) / h ** 2
    bx, by = peclet * 2.0 / h, 0.5 * peclet * 2.0 / h
    Cx = sp.diags([-e[:-1], e], [-1, 0]) * (bx / h)          # upwind differences (β > 0)
    Cy = sp.diags([-e[:-1], e], [-1, 0]) * (by / h)
    I = sp.identity(n)
    A = (sp.kron(I, T + Cx) + sp.kron(T + Cy, I)).tocsr()
    if scale_rows:
        d = 10.0 ** (2.0 * np.arange(n * n) / (n * n) - 1.0)
        A = (sp.diags(d) @ A).tocsr()
    return A


@pytest.mark.parametrize("case", ["peclet0.5", "peclet5", "peclet50", "rows_scaled_1e2"])
def test_default_sstep_on_user_matrices_with_complex_spectra_and_bad_scaling(nls, dev, case):
    """The library default (s-step, automatic block size, Newton basis on the REAL Gershgorin interval) on general user CSR
    matrices the built-in problems do not cover: convection-dominated nonsymmetric operators (eigenvalues off the real axis —
    the real interval is then only the projection of the discs) and a matrix whose rows are scaled over two decades (bounds
    far wider than the bulk of the spectrum: blocks may lose rank, the solve narrows them and, if need be, finishes column by column).
    Whatever the blocks do, the answer is the column-by-column GMRES's: same residual history end point, same solution."""
    import torch
    A = _convection_diffusion(40, {"peclet0.5": 0.5, "peclet5": 5.0, "peclet50": 50.0, "rows_scaled_1e2": 2.0}[case],
                              scale_rows=(case == "rows_scaled_1e2"))
    n = A.shape[0]
    b = A @ np.sin(np.arange(n) * 0.05) + 1.0
    J = nls.CSRMatrix.from_scipy(A)
    rtol = 1e-8
    xr, ir = R.gmres(lambda z: A @ z, b, restart=30, rtol=rtol, atol=0.0, itmax=4000, ortho="cgs2")
    out = {}
    for ortho in ("sstep", "dcgs2"):
        G = nls.GMRES(n, restart=30, ortho=ortho).set_operator(J)
        x, gi = G.solve(torch.tensor(b, device=dev), reltol=rtol, abstol=0.0, maxiters=4000)
        out[ortho] = (x.cpu().numpy(), gi, G.sstep_state() if ortho == "sstep" else None)
        G.close()
    xs, gs, st = out["sstep"]
    xd, gd, _ = out["dcgs2"]
    assert gs["converged"] and gd["converged"] and ir.converged, (gs, gd)
    for x in (xs, xd):   # both reach the tolerance in the TRUE residual (the recurrence residual of a restarted solve is re-based)
        assert np.linalg.norm(b - A @ x) <= 5.0 * rtol * np.linalg.norm(b)
    # restarted GMRES(30) near stagnation is sensitive to rounding in its iteration count, not in what it converges to
    scale = np.linalg.norm(xr)
    assert np.linalg.norm(xs - xr) <= 1e-5 * scale and np.linalg.norm(xd - xr) <= 1e-5 * scale
    assert gs["iters"] <= 1.5 * gd["iters"] + 60, (gs["iters"], gd["iters"], st)
    J.close()


# ----------------------------------------------------------------------------- the forms behind the A/B switches stay tested
_SWITCH_CODE = (
    "import numpy as np, torch, nonlinearsolve_jl_amd as nls\n"
    "out = {}\n"
    "for name, ns, kw in (('fixed', 96, dict(fixed_iters=30, maxiters=30)), ('tol', 24, dict(maxiters=600, reltol=1e-13, abstol=0.0))):\n"
    "    prob = nls.NonlinearProblem(nls.Bratu2D(ns, 6.0))\n"
    "    alg = nls.NewtonRaphson(linsolve=nls.KrylovJL_GMRES(gmres_restart=30, ortho='sstep', **kw), concrete_jac=True)\n"
    "    cache = nls.init(prob, alg, abstol=1e-300, maxiters=50)\n"
    "    for _ in range(4): cache.step()\n"
    "    u = cache.u\n"
    "    out[name] = np.asarray(u.cpu() if hasattr(u, 'cpu') else u)\n"
    "np.savez(OUT, **out)\n")


@pytest.mark.parametrize("env", [dict(NK_SS_DEFER="0"), dict(NK_SS_DEFER_HESS="0"), dict(NK_SS_DEFER_HESS="1"),
                                 dict(NK_SS_TAIL_BACK="0"), dict(NK_SS_HOST_B="0"), dict(NK_SS_IMPLICIT="0"),
                                 dict(NK_SS_FUSED="0"), dict(NK_SS_HOST_A="0"), dict(NK_SS_HOST_A_WGS="2"),
                                 dict(NK_FUSED_UPDATE="0", NK_FUSED_RESIDUAL_NORMS="0"), dict(NK_PRELOADED_RHS="0"),
                                 dict(NK_SS_NOSTORE="0"), dict(NK_BEGIN_AHEAD="0"), dict(NK_FOLD_NORMS="0"),
                                 dict(NK_SS_RO="0"), dict(NK_SS_RO_GRID="0")])   # (round 6's five)
def test_every_form_behind_an_ab_switch_reaches_the_same_iterates(env):
    """The s-step cycle's forms that the defaults do not take — the second factorisation in a launch of its own (round 4's cycle),
    the Hessenberg work inside the scalar launch / hosted by sweep B whatever the protocol, the back-substitution as a launch of
    its own, the explicit third sweep, the unfused scalar launches, the first block closed by a launch of its own instead of by
    workgroups of the second block's sweep A (and, at this size, the reverse: two hosted workgroups where the default grid is too
    small to give any up), the Newton update and the residual's norms as launches of their own, the right-hand side copied into the basis by a pass of its own — are the same arithmetic in another order: four Newton steps
    (fixed work at 96², and GMRES to a tight tolerance at 24² — tight and small so that the linear solves are converged, not cut
    off at the iteration cap: two restarted solves cut off there differ by what rounding does to ten restart cycles) leave iterates
    equal to the default's to rounding."""
    import os
    import subprocess
    import sys
    import tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    for name, e in (("default", {}), ("switched", env)):
        with tempfile.NamedTemporaryFile(suffix=".npz") as tf:
            code = _SWITCH_CODE.replace("OUT", repr(tf.name))
            r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **e), capture_output=True, text=True, timeout=300,
                               cwd=root)
            assert r.returncode == 0, r.stderr[-2000:]
            d = np.load(tf.name)
            res[name] = {k: d[k] for k in d.files}
    for k in res["default"]:
        a, b = res["default"][k], res["switched"][k]
        assert np.max(np.abs(a - b)) <= (1e-10 if k == "fixed" else 1e-9) * max(1.0, float(np.max(np.abs(a)))), \
            (env, k, float(np.max(np.abs(a - b))))
