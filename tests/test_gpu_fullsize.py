"""Parity at BASELINE.json's full sizes (C3: 1024², C4: 4096² Bratu; C5: Brusselator 512²) — against the C oracle
where it finishes in seconds, and through size-independent properties (linearity, symmetry, CSR ≡ matrix-free
JVP, true-residual checks, monotone Newton residuals) where it does not."""
import numpy as np
import pytest

from oracle import c_oracle as CO

pytestmark = pytest.mark.gpu


def relerr(a, b):
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))


@pytest.mark.parametrize("ns", [1024, 4096])
def test_bratu_fullsize_kernels_vs_c_oracle(nls, dev, ns):
    import torch
    n = ns * ns
    rng = np.random.default_rng(ns)
    u = 0.3 * rng.standard_normal(n)
    v = rng.standard_normal(n)
    P = nls.Bratu2D(ns, 6.0)
    du, dv = torch.tensor(u, device=dev), torch.tensor(v, device=dev)
    f = P.residual(du).cpu().numpy()
    assert relerr(f, CO.bratu_residual(ns, 6.0, 0.0, u)) <= 1e-13
    jv = P.jvp(dv, du)
    assert relerr(jv.cpu().numpy(), CO.bratu_jvp(ns, 6.0, 0.0, u, v)) <= 1e-13
    # assembled CSR: values, SpMV (bit-exact row sums), and CSR·v ≡ matrix-free JVP
    J = P.jac_csr()
    P.jac_values(du, J)
    assert J.info()["nnz"] == 5 * n - 4 * ns
    rp, ci = CO.bratu_pattern(ns)
    val = CO.bratu_jac_values(ns, 6.0, 0.0, u, rp)
    y = J.matvec(dv)
    assert np.array_equal(y.cpu().numpy(), CO.spmv(rp, ci, val, v))
    assert float((y - jv).abs().max() / jv.abs().max()) <= 1e-13
    # symmetry of the Bratu Jacobian: xᵀ(Jy) = yᵀ(Jx), and Jᵀ via the transposed SpMV
    x2 = torch.tensor(rng.standard_normal(n), device=dev)
    ctx = nls.default_context()
    a, b = ctx.dot(x2, J.matvec(dv)), ctx.dot(dv, J.matvec(x2))
    assert abs(a - b) <= 1e-11 * abs(a)
    # linearity: J(αx + βy) = αJx + βJy
    lin = J.matvec(2.5 * x2 - 0.5 * dv) - (2.5 * J.matvec(x2) - 0.5 * J.matvec(dv))
    assert float(lin.abs().max()) <= 1e-10 * float(J.matvec(x2).abs().max())


def test_c3_gmres_true_residual_and_newton_descent(nls, dev):
    """C3 (1024², matrix-free JVP + GMRES(30)): the recurrence residual equals the true residual ‖b − J x‖, GMRES(30)
    is monotone, and fixed-work Newton steps reproduce the C oracle's ‖F‖∞ trace and iterate at full size."""
    import torch
    ns = 1024
    n = ns * ns
    prob = nls.NonlinearProblem(nls.Bratu2D(ns, 6.0))
    u = torch.zeros(n, dtype=torch.float64, device=dev)
    op = nls.StatefulJacobianOperator(nls.JacobianOperator(prob), u)
    G = nls.GMRES(n, restart=30).set_operator(op)
    b = prob.device_problem.residual(u)
    ctx = nls.default_context()
    last = None
    for iters in (30, 90):
        x, info = G.solve(b, fixed_iters=iters)
        r = b - (op @ x)
        true = ctx.nrm2(r)
        assert abs(true - info["rnorm"]) <= 1e-8 * info["rnorm0"]
        assert info["rnorm"] < info["rnorm0"] and (last is None or info["rnorm"] <= last)
        last = info["rnorm"]
    cache = nls.init(nls.NonlinearProblem(nls.Bratu2D(ns, 6.0), u0=u),
                     nls.NewtonRaphson(linsolve=nls.KrylovJL_GMRES(fixed_iters=30, maxiters=30)), abstol=1e-300,
                     maxiters=100, store_trace=True)
    nsteps = 4
    for _ in range(nsteps):
        cache.step()
    fn = np.array([t["fnorm_inf"] for t in cache.trace])
    # the same fixed-work protocol on the C oracle (MGS there, CGS2 here): ‖F‖∞ after every step and the iterate
    uC, fnC, giC, _ = CO.bratu_newton(ns, 6.0, 0.0, np.zeros(n), nsteps, use_csr=False, m=30, itmax=30, fixed_iters=30,
                                      forcing=False)
    assert np.all(giC == 30)
    assert np.allclose(fn, fnC, rtol=1e-6), (fn, fnC)
    assert np.max(np.abs(cache.u.cpu().numpy() - uC)) <= 1e-9
    assert cache.stats.gmres_iters == 30 * nsteps and cache.stats.nf == nsteps
    cache.close()


def test_c5_brusselator_512_kernels(nls, dev):
    """C5 size (N_g = 512, 524 288 unknowns): residual vs the C oracle, JVP/VJP adjointness vᵀ(J w) = wᵀ(Jᵀ v),
    CSR ≡ matrix-free JVP, and nnz = 6·unknowns."""
    import torch
    N = 512
    P = nls.Brusselator2D(N)
    u0 = P.initial_guess(device=True)
    rng = np.random.default_rng(5)
    u = u0 + torch.tensor(0.1 * rng.standard_normal(2 * N * N), device=dev)
    f = P.residual(u).cpu().numpy()
    ref = CO.brusselator_residual(N, 3.4, 1.0, 10.0, 1.0 / (N - 1), u.cpu().numpy())
    assert relerr(f, ref) <= 1e-12
    v = torch.tensor(rng.standard_normal(2 * N * N), device=dev)
    w = torch.tensor(rng.standard_normal(2 * N * N), device=dev)
    ctx = nls.default_context()
    a, b = ctx.dot(v, P.jvp(w, u)), ctx.dot(w, P.vjp(v, u))
    assert abs(a - b) <= 1e-10 * abs(a)
    J = P.jac_csr()
    P.jac_values(u, J)
    assert J.info()["nnz"] == 6 * 2 * N * N
    jw = P.jvp(w, u)
    assert float((J.matvec(w) - jw).abs().max() / jw.abs().max()) <= 1e-12
    jtv = P.vjp(v, u)
    assert float((J.rmatvec(v) - jtv).abs().max() / jtv.abs().max()) <= 1e-12


def test_c3_full_solve_with_chebyshev_precs_vs_c_oracle(nls, dev):
    """C3 at full size, solved to ‖h²F‖∞ ≤ 1e-8: NewtonRaphson + GMRES(30) + Eisenstat–Walker + Chebyshev(32, 300)
    right preconditioner on the assembled CSR Jacobian — against the C oracle running the same algorithm."""
    import torch
    ns = 1024
    prob = nls.NonlinearProblem(nls.Bratu2D(ns, 6.0), u0=torch.zeros(ns * ns, dtype=torch.float64, device=dev))
    alg = nls.NewtonRaphson(linsolve=nls.KrylovJL_GMRES(gmres_restart=30, maxiters=300, precs=nls.ChebyshevPrecs(32, 300.0)),
                            forcing=nls.EisenstatWalkerForcing2(), concrete_jac=True)
    sol = nls.solve(prob, alg, abstol=1e-8, maxiters=50)
    uC, fnC, giC = CO.bratu_newton_cheb(ns, 6.0, 0.0, np.zeros(ns * ns), 50, True, 30, 300, 32, 300.0, 1e-8)
    assert sol.retcode == "Success" and float(sol.resid.abs().max()) <= 1e-8 and fnC[-1] <= 1e-8
    assert abs(sol.stats.nsteps - len(fnC)) <= 1
    assert abs(sol.stats.gmres_iters - int(giC.sum())) <= 0.1 * giC.sum() + 5
    # both iterates only satisfy ‖h²F‖∞ ≤ 1e-8 and λmin(h²J) ≈ 2π²h² ≈ 2e-5, so they may differ by up to
    # ≈ 2·1e-8/2e-5 = 1e-3 in the worst case; observed 3e-6. The stated 1e-8 parity bound holds for runs converged
    # tightly (tests/test_gpu_solvers.py::test_bratu_newton_tight_inner_matches_direct).
    assert np.max(np.abs(sol.u.cpu().numpy() - uC)) <= 2e-5
    assert 0.79 < float(sol.u.max()) < 0.80


def test_bratu_1024_multigrid_precs_reaches_tolerance_in_a_handful_of_krylov_iterations(nls):
    """Config C3 at full size with the built-in geometric multigrid `precs`: ‖h²F‖∞ ≤ 1e-8 in ≤ 6 Newton steps and ≤ 8
    GMRES iterations in total (mesh-independent: the same counts as at 128² in test_gpu_solvers.py), and the solution agrees
    with the Chebyshev-preconditioned one to the conditioning-limited 1e-4."""
    import torch
    ns = 1024
    prob = nls.NonlinearProblem(nls.Bratu2D(ns, 6.0), u0=torch.zeros(ns * ns, dtype=torch.float64, device="cuda"))
    mg = nls.solve(prob, nls.NewtonRaphson(linsolve=nls.KrylovJL_GMRES(precs=nls.MultigridPrecs(2, 31)),
                                           forcing=nls.EisenstatWalkerForcing2()), abstol=1e-8, maxiters=50)
    assert mg.retcode == "Success" and float(mg.resid.abs().max()) <= 1e-8
    assert mg.stats.nsteps <= 6 and mg.stats.gmres_iters <= 8
    ch = nls.solve(prob, nls.NewtonRaphson(linsolve=nls.KrylovJL_GMRES(precs=nls.ChebyshevPrecs(32, 300.0)),
                                           forcing=nls.EisenstatWalkerForcing2()), abstol=1e-8, maxiters=50)
    assert ch.retcode == "Success" and float((mg.u - ch.u).abs().max()) < 1e-4
    assert abs(float(mg.u.max()) - 0.797) < 2e-3     # the λ = 6 lower-branch solution peaks at ≈ 0.797


def test_c5_brusselator512_trust_region_vs_direct_solve_oracle(nls, dev):
    """Config C5 as BASELINE.json words it, at full size: Brusselator 2-D steady state, N_g = 512 (524 288 unknowns),
    TrustRegion + GMRES(30) on the concrete sparse Jacobian assembled by colour-compressed sweeps every step (and, second
    run, closed-form values), Chebyshev polynomial behind the `precs` hook. Compared with the CPU oracle's TrustRegion()
    with a DIRECT linear solve at the same size (tests/golden/c5_brusselator512_tr_direct.npz, generated by
    tests/golden/make_c5_golden.py — SuperLU, minutes): same number of steps, same accept/reject sequence, the iterate
    at 8 192 sample points and its norms within 1e-6 relative (inner solves at rtol 1e-9 vs exact ones), and
    ‖f(u)‖∞ ≤ abstol = 1e-7 confirmed by the C oracle's residual on the whole vector (the residual carries ≈ 1.4e-8 of
    rounding at this size — α/dx² = 2.6e6 — so the reference-style 1e-8 is not reachable by any solver)."""
    import os
    import torch
    from oracle import c_oracle as COr
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "c5_brusselator512_tr_direct.npz")
    if not os.path.exists(path):
        pytest.skip("fixture not generated (tests/golden/make_c5_golden.py)")
    g = np.load(path)
    N = int(g["N"])
    assert N == 512
    P = nls.Brusselator2D(N)
    for colored in (True, False):
        prob = nls.NonlinearProblem(P, u0=P.initial_guess(device=True))
        alg = nls.TrustRegion(linsolve=nls.KrylovJL_GMRES(gmres_restart=30, maxiters=6000, reltol=1e-9, abstol=0.0,
                                                          precs=nls.ChebyshevPrecs(128, 1.0e4)),
                              concrete_jac=True, jac_colored=colored)
        sol = nls.solve(prob, alg, abstol=1e-7, maxiters=30, store_trace=True)   # (1e-8 is below the residual's rounding floor at this size)
        u = sol.u.cpu().numpy()
        assert sol.retcode == "Success"
        assert sol.stats.nsteps == int(g["nsteps"])
        assert [int(t["accepted"]) for t in sol.trace] == list(g["accepted"])
        assert np.allclose([t["trust_region"] for t in sol.trace], g["trust_region"], rtol=1e-6)
        scale = float(g["u_inf"])
        assert np.max(np.abs(u[::int(g["stride"])] - g["u_samples"])) <= 1e-6 * scale
        assert abs(np.linalg.norm(u) - float(g["u_l2"])) <= 1e-6 * float(g["u_l2"]) and abs(np.max(np.abs(u)) - scale) <= 1e-6 * scale
        f = COr.brusselator_residual(N, 3.4, 1.0, 10.0, 1.0 / (N - 1), u)
        assert np.max(np.abs(f)) <= 1e-7
        assert sol.stats.njacs == sol.stats.nsteps + 1 or sol.stats.njacs >= 1


# ----------------------------------------------------------------------------- config C4 at its full size (4096², 16.8 M unknowns)
@pytest.mark.parametrize("ortho", ["sstep", "dcgs2"])
def test_c4_fixed_work_newton_vs_c_oracle_at_full_size(nls, dev, ortho):
    """C4 at 4096² on one GPU, the headline protocol (assembled CSR, 30 Arnoldi steps per Newton step), four Newton steps: ‖F‖∞
    after every step (rtol 1e-6) and the iterate (1e-9) of the C oracle's restatement of the SAME arithmetic — the s-step form
    with the Newton basis (15 columns per block: the library default) and delayed CGS2."""
    import torch
    ns, nst = 4096, 4
    n = ns * ns
    kw = {} if ortho == "sstep" else dict(ortho="dcgs2")
    cache = nls.init(nls.NonlinearProblem(nls.Bratu2D(ns, 6.0), u0=torch.zeros(n, dtype=torch.float64, device=dev)),
                     nls.NewtonRaphson(linsolve=nls.KrylovJL_GMRES(fixed_iters=30, maxiters=30, **kw), concrete_jac=True),
                     abstol=1e-300, maxiters=100, store_trace=True)
    for _ in range(nst):
        cache.step()
    fn = np.array([t["fnorm_inf"] for t in cache.trace])
    u = cache.u.cpu().numpy()
    cache.close()
    if ortho == "sstep":
        uC, fnC, _ = CO.bratu_newton_fast_sstep(ns, 6.0, 0.0, np.zeros(n), nst, use_csr=True, m=30, s=15, basis="newton")
    else:
        uC, fnC, _ = CO.bratu_newton_fast(ns, 6.0, 0.0, np.zeros(n), nst, use_csr=True, m=30)
    assert np.allclose(fn, fnC, rtol=1e-6), (fn, fnC)
    assert np.max(np.abs(u - uC)) <= 1e-9


def test_c4_multigrid_solve_counts_and_residual_at_full_size(nls, dev):
    """The solve BASELINE.md quotes for 4096² — NewtonRaphson + Eisenstat–Walker + GMRES(30) with the geometric V-cycle behind
    `precs`: 3 Newton steps / 3 Krylov iterations. At 2048² the whole solve against the NumPy oracle's BratuMultigrid (steps,
    iterations, iterate); at 4096², where the SciPy hierarchy would take minutes and ≈ 8 GB, the mesh-independent counts and the
    residual of the returned iterate evaluated by the C oracle's own kernel (a size-independent property: ‖F(u)‖∞ ≤ abstol)."""
    import torch
    from oracle import reference_restatement as R
    outs = {}
    for ns in (2048, 4096):
        n = ns * ns
        alg = nls.NewtonRaphson(linsolve=nls.KrylovJL_GMRES(gmres_restart=30, maxiters=300, precs=nls.MultigridPrecs(2, 31)),
                                forcing=nls.EisenstatWalkerForcing2())
        sol = nls.solve(nls.NonlinearProblem(nls.Bratu2D(ns, 6.0), u0=torch.zeros(n, dtype=torch.float64, device=dev)), alg,
                        abstol=1e-8, maxiters=50)
        assert sol.retcode == "Success"
        u = sol.u.cpu().numpy()
        assert np.max(np.abs(CO.bratu_residual(ns, 6.0, 0.0, u))) <= 1e-8          # the oracle's residual of the device's root
        outs[ns] = (sol.stats.nsteps, sol.stats.gmres_iters, u)
    assert outs[4096][0] == 3 and outs[4096][1] <= 4 and outs[2048][0] in (3, 4) and outs[2048][1] <= 5   # mesh-independent
    ref = R.solve(R.Bratu2D(2048, 6.0), R.NewtonRaphson(linsolve=R.KrylovJL_GMRES(gmres_restart=30, maxiters=300,
                                                                                     precs=R.MultigridPrecs(2, 31), ortho="cgs2"),
                                                          forcing=R.EisenstatWalkerForcing2()), abstol=1e-8, maxiters=50)
    assert R.RETCODE_NAMES[ref.retcode] == "Success" and ref.stats.nsteps == outs[2048][0]
    assert abs(ref.stats.gmres_iters - outs[2048][1]) <= 1
    assert np.max(np.abs(outs[2048][2] - ref.u)) <= 5e-7
    # the two grids discretise the same boundary-value problem: the maxima over the grid points agree to O(h)
    assert abs(outs[4096][2].max() - outs[2048][2].max()) <= 2e-4


def test_c4_two_ranks_through_the_bench_code_path(tmp_path):
    """`bench.py --gpus 2 --workload c4` (two processes sharing this GPU, gloo rendezvous, peer-mapped arenas): the residual after
    the same fixed-work steps equals the single-rank run's — the partitioned 4096² path end to end, as the driver launches it."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    common = ["--workload", "c4", "--steps", "3", "--warmup", "1", "--cpu-seconds", "0", "--no-ttt", "--no-weak", "--no-profile-pass"]
    outs = {}

    def run(g):
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(g)] + common, env=env, capture_output=True,
                           text=True, timeout=900)
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        assert r.returncode == 0 and lines, r.stdout[-2000:] + r.stderr[-2000:]
        return json.loads(lines[-1])

    outs[1] = run(1)
    # ONE attempt. (Round 5 gave this leg three: two-process runs were occasionally wrong or hung. The cause was a race in the
    # Gram block's factorisation — profiles/r06_a_shared_device_root_cause.md —, fixed in round 6; tests/test_gpu_determinism.py
    # repeats the failing set-ups.)
    outs[2] = run(2)
    a, b = outs[1]["check"]["fnorm_inf_after_timed_steps"], outs[2]["check"]["fnorm_inf_after_timed_steps"]
    assert abs(a - b) <= 1e-8 * abs(a), (a, b)  # (another partition: another summation order in every inner product)
    assert outs[2]["n_gpus"] == 2 and outs[2]["config"]["unknowns_global"] == 4096 * 4096
    assert outs[2]["check"]["allreduces"] > 0 and outs[2]["check"]["halo_exchanges"] > 0
