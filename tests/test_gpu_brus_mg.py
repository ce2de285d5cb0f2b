"""The geometric multigrid `precs` for the Brusselator Jacobian (two coupled species, periodic grid; config C5) vs the
oracle's restatement (oracle/reference_restatement.py::BrusselatorMultigrid), and config C5 at full size with it against
the direct-solve TrustRegion fixture. The reference's hook is `precs(A, p)` with an algebraic-multigrid preconditioner
(docs/src/tutorials/large_systems.md:244-316), which is problem agnostic; this is its device counterpart for the
built-in Brusselator."""
import os

import numpy as np
import pytest

from oracle import reference_restatement as R

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("N,coarse,nu", [(32, 8, 2), (64, 8, 2), (48, 8, 3), (128, 16, 2), (40, 31, 2)])
def test_brusselator_multigrid_preconditioner_vs_oracle(nls, N, coarse, nu):
    """One V-cycle equals the oracle's to rounding; right-preconditioned GMRES takes the oracle's iteration count, and the
    count does not grow with the grid (7 ± 1 to 1e-9)."""
    pb = R.Brusselator2D(N)
    u = pb.u0() + 0.05 * np.random.default_rng(1).standard_normal(pb.n)
    J = pb.jac(u)
    b = np.random.default_rng(2).standard_normal(pb.n)
    Mo = R.BrusselatorMultigrid(pb, u, nu, coarse)
    P = nls.Brusselator2D(N)
    op = nls.StatefulJacobianOperator(nls.JacobianOperator(nls.NonlinearProblem(P)), u)
    G = nls.GMRES(pb.n, restart=30).set_operator(op)
    G.set_multigrid_preconditioner(P, u, nu=nu, coarse_max=coarse)
    x1, _ = G.solve(b, fixed_iters=1)
    xo1, _ = R.gmres(lambda z: J @ z, b, restart=30, fixed_iters=1, M=Mo, ortho="cgs2")
    assert np.linalg.norm(x1 - xo1) <= 1e-9 * np.linalg.norm(xo1)
    xo, io = R.gmres(lambda z: J @ z, b, rtol=1e-9, restart=30, itmax=300, M=Mo, ortho="cgs2")
    x, info = G.solve(b, abstol=0.0, reltol=1e-9, maxiters=300)
    assert info["converged"] and abs(info["iters"] - io.iters) <= 1 and info["iters"] <= 9
    assert np.linalg.norm(x - xo) <= 1e-6 * np.linalg.norm(xo)
    assert np.linalg.norm(J @ x - b) <= 1.01e-9 * np.linalg.norm(b)


@pytest.mark.parametrize("concrete", [False, True])
def test_brusselator_trust_region_with_multigrid_precs_vs_oracle(nls, concrete):
    """TrustRegion + GMRES with the V-cycle behind `precs` (re-linearised for every new Jacobian) at N = 64: the oracle's
    step count, accept/reject sequence, radii and iterate; about a handful of Krylov iterations per step."""
    N = 64
    kw = dict(gmres_restart=30, maxiters=300)
    ref = R.solve(R.Brusselator2D(N), R.TrustRegion(linsolve=R.KrylovJL_GMRES(precs=R.MultigridPrecs(2, 8), **kw),
                                                    concrete_jac=concrete), abstol=1e-8, maxiters=40)
    sol = nls.solve(nls.NonlinearProblem(nls.Brusselator2D(N)),
                    nls.TrustRegion(linsolve=nls.KrylovJL_GMRES(precs=nls.MultigridPrecs(2, 8), **kw), concrete_jac=concrete),
                    abstol=1e-8, maxiters=40, store_trace=True)
    assert sol.retcode == "Success" == R.RETCODE_NAMES[ref.retcode]
    assert sol.stats.nsteps == ref.stats.nsteps
    assert [t["accepted"] for t in sol.trace] == [t["accepted"] for t in ref.trace]
    assert np.allclose([t["trust_region"] for t in sol.trace], [t["trust_region"] for t in ref.trace], rtol=1e-6)
    assert abs(sol.stats.gmres_iters - ref.stats.gmres_iters) <= 2 * sol.stats.nsteps
    assert sol.stats.gmres_iters <= 12 * sol.stats.nsteps
    assert np.max(np.abs(np.asarray(sol.u) - ref.u)) <= 1e-6 * np.max(np.abs(ref.u))


def test_c5_brusselator512_with_multigrid_precs_vs_direct_solve_oracle(nls, dev):
    """Config C5 at full size (N_g = 512, 524 288 unknowns): TrustRegion + GMRES(30) on the concrete Jacobian with the
    multigrid V-cycle behind `precs`, against the oracle's direct-solve TrustRegion fixture
    (tests/golden/c5_brusselator512_tr_direct.npz): same steps, same accept/reject sequence, radii and iterate — in a few
    Krylov iterations per step where the Chebyshev polynomial needs hundreds of operator applications."""
    import time
    import torch
    from oracle import c_oracle as COr
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "c5_brusselator512_tr_direct.npz")
    g = np.load(path)
    N = int(g["N"])
    P = nls.Brusselator2D(N)
    alg = nls.TrustRegion(linsolve=nls.KrylovJL_GMRES(gmres_restart=30, maxiters=300, reltol=1e-9, abstol=0.0,
                                                      precs=nls.MultigridPrecs(2, 16)), concrete_jac=True)
    for rep in range(2):
        prob = nls.NonlinearProblem(P, u0=P.initial_guess(device=True))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        sol = nls.solve(prob, alg, abstol=1e-7, maxiters=30, store_trace=True)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    print(f"C5 512^2 TrustRegion + GMRES(30) + multigrid precs: {dt * 1e3:.1f} ms, {sol.stats.nsteps} steps, "
          f"{sol.stats.gmres_iters} Krylov iterations")
    u = sol.u.cpu().numpy()
    assert sol.retcode == "Success" and sol.stats.nsteps == int(g["nsteps"])
    assert [int(t["accepted"]) for t in sol.trace] == list(g["accepted"])
    assert np.allclose([t["trust_region"] for t in sol.trace], g["trust_region"], rtol=1e-6)
    scale = float(g["u_inf"])
    assert np.max(np.abs(u[::int(g["stride"])] - g["u_samples"])) <= 1e-6 * scale
    assert sol.stats.gmres_iters <= 15 * sol.stats.nsteps
    f = COr.brusselator_residual(N, 3.4, 1.0, 10.0, 1.0 / (N - 1), u)
    assert np.max(np.abs(f)) <= 1e-7
