"""GPU parity tests of LevenbergMarquardt through the C ABI against the oracle:
DampedNewtonDescent in :normal_form mode (descent/damped_newton.jl:297-313) on the device normal-form operator plus a
diagonal, the LM damping cache (levenberg_marquardt.jl:72-168), geodesic acceleration (geodesic_acceleration.jl:98-136)
and the damping-based trust region (levenberg_marquardt.jl:247-268); known answers of the reference's own tests
(rootfind_tests__item14/15/17.jl)."""
import numpy as np
import pytest
import scipy.sparse as sp

from oracle import reference_restatement as R

pytestmark = pytest.mark.gpu

CASES = {"quad": (lambda: R.Quadratic(20, 2.0), lambda nls: nls.Quadratic(20, 2.0)),
         "bratu12": (lambda: R.Bratu2D(12), lambda nls: nls.Bratu2D(12)),
         "brus6": (lambda: R.Brusselator2D(6), lambda nls: nls.Brusselator2D(6))}


def _algs(nls, geo, **kw):
    k = dict(gmres_restart=60, maxiters=600)
    return (R.LevenbergMarquardt(linsolve=R.KrylovJL_GMRES(**k), disable_geodesic=not geo, **kw),
            nls.LevenbergMarquardt(linsolve=nls.KrylovJL_GMRES(**k), disable_geodesic=not geo, **kw))


def test_colsumsq_is_the_diagonal_of_JtJ(nls, dev):
    """nk_csr_colsumsq: Σ_i A_ij² through the transposed pattern — `sum!(abs2, J_diag_cache, J')`
    (levenberg_marquardt.jl:133-148); a following Aᵀx still sees the plain values."""
    import torch
    pb = R.Brusselator2D(9)
    J = sp.csr_matrix(pb.jac(pb.u0()))
    A = nls.CSRMatrix.from_scipy(J)
    d = A.colsumsq()
    ref = np.asarray(J.multiply(J).sum(axis=0)).ravel()
    assert np.max(np.abs(d - ref)) <= 1e-13 * np.max(ref)
    x = np.linspace(-1.0, 2.0, J.shape[0])
    assert np.max(np.abs(A.rmatvec(x) - J.T @ x)) <= 1e-12 * np.max(np.abs(J.T @ x))
    dd = A.colsumsq(like=torch.empty(1, dtype=torch.float64, device=dev))
    assert dd.is_cuda and np.array_equal(dd.cpu().numpy(), d)


@pytest.mark.parametrize("which", list(CASES))
@pytest.mark.parametrize("geo", [True, False])
def test_levenberg_marquardt_matches_oracle(nls, which, geo):
    """Whole solves: retcode, step count, the per-step damping λ, the uphill cosine β, accept flags and the iterate."""
    mk_ref, mk_dev = CASES[which]
    ralg, dalg = _algs(nls, geo)
    ref = R.solve(mk_ref(), ralg, abstol=1e-8, maxiters=200)
    sol = nls.solve(nls.NonlinearProblem(mk_dev(nls)), dalg, abstol=1e-8, maxiters=200, store_trace=True)
    assert sol.retcode == R.RETCODE_NAMES[ref.retcode] == "Success"
    assert sol.stats.nsteps == ref.stats.nsteps
    assert sol.stats.nf == ref.stats.nf and sol.stats.njacs == ref.stats.njacs and sol.stats.nsolve == ref.stats.nsolve
    lam_d = np.array([t["trust_region"] for t in sol.trace]), np.array([t["trust_region"] for t in ref.trace])
    assert np.allclose(lam_d[0], lam_d[1], rtol=1e-12)
    assert [t["accepted"] for t in sol.trace] == [t["accepted"] for t in ref.trace]
    bd = np.array([t["rho"] for t in sol.trace]); br = np.array([t["rho"] for t in ref.trace])
    m = np.isfinite(br)
    assert np.array_equal(np.isfinite(bd), m) and np.allclose(bd[m], br[m], atol=1e-6)
    assert np.max(np.abs(np.asarray(sol.u) - ref.u)) <= 1e-7 * max(1.0, np.max(np.abs(ref.u)))
    assert np.max(np.abs(np.asarray(sol.resid))) <= 1e-8


def test_levenberg_marquardt_rejects_steps_with_large_acceleration(nls):
    """α_geodesic small enough that 2‖a‖ > α‖v‖ on some steps: those steps are not taken (u stays, the Jacobian is kept,
    λ doubles), and the device makes the same decisions as the oracle."""
    ralg, dalg = _algs(nls, True, alpha_geodesic=0.02, damping_initial=0.01)
    ref = R.solve(R.Bratu2D(10, 6.5), ralg, abstol=1e-9, maxiters=60)
    sol = nls.solve(nls.NonlinearProblem(nls.Bratu2D(10, 6.5)), dalg, abstol=1e-9, maxiters=60, store_trace=True)
    acc_r = [t["accepted"] for t in ref.trace]
    assert not all(acc_r), "the case is meant to contain rejected steps"
    assert [t["accepted"] for t in sol.trace] == acc_r
    assert np.allclose([t["trust_region"] for t in sol.trace], [t["trust_region"] for t in ref.trace], rtol=1e-12)
    assert sol.retcode == R.RETCODE_NAMES[ref.retcode] and sol.stats.nsteps == ref.stats.nsteps
    assert sol.stats.njacs == ref.stats.njacs          # a step that is not taken keeps the Jacobian (solve.jl:455-457)
    assert np.max(np.abs(np.asarray(sol.u) - ref.u)) <= 1e-7 * max(1.0, np.max(np.abs(ref.u)))


@pytest.mark.parametrize("tc", range(9))
def test_levenberg_marquardt_quadratic_all_termination_conditions(nls, tc):
    """rootfind_tests__item14.jl / item17.jl: quadratic_f from u0 = ones reaches |u² − 2| < 1e-9 under every mode."""
    cond = nls.TERMINATION_CONDITIONS[tc]
    sol = nls.solve(nls.NonlinearProblem(nls.Quadratic(2, 2.0)), nls.LevenbergMarquardt(linsolve=nls.KrylovJL_GMRES()),
                    termination_condition=cond)
    u = np.asarray(sol.u)
    assert np.max(np.abs(u * u - 2.0)) < 1e-9


def test_levenberg_marquardt_newton_fails_fixture(nls, dev):
    """rootfind_tests__item15.jl: the seven-start `newton_fails` system converges with LevenbergMarquardt()."""
    import torch

    def nf(u):
        return 0.010000000000000002 + 10.000000000000002 / (1 + (0.21640425613334457 + 216.40425613334457 / (
            1 + (0.21640425613334457 + 216.40425613334457 / (1 + 0.0006250000000000001 * (u ** 2.0))) ** 2.0)) ** 2.0) \
            - 0.0011552453009332421 * u

    def f(du, u, p):
        du.copy_(nf(u))

    def jac(Jv, u, p):       # values of the diagonal pattern
        Jv.copy_((nf(u + 1e-7) - nf(u - 1e-7)) / 2e-7)

    u0 = torch.tensor([-10.0, -1.0, 1.0, 2.0, 3.0, 4.0, 10.0], dtype=torch.float64, device=dev)
    fn = nls.NonlinearFunction(f, jac=jac, jac_prototype=nls.CSRMatrix.from_scipy(sp.identity(7, format="csr")))
    # (abstol is forwarded to the Krylov solver, FirstOrder/src/solve.jl:203: with b = Jᵀf and |J| ≈ 1e-3 here the default
    #  3e-13 makes the inner solves return x = 0 before ‖f‖∞ gets there — on the oracle as on the device, both end Stalled
    #  at 1.5e-11; the reference's test runs the QR form. 1e-10 reaches the test's bound |f| < 1e-9 with Success on both.)
    sol = nls.solve(nls.NonlinearProblem(fn, u0), nls.LevenbergMarquardt(linsolve=nls.KrylovJL_GMRES()), abstol=1e-10)
    pr = R.FunctionProblem(lambda u: nf(u), np.asarray(u0.cpu()), jac=lambda u: sp.diags((nf(u + 1e-7) - nf(u - 1e-7)) / 2e-7))
    ref = R.solve(pr, R.LevenbergMarquardt(linsolve=R.KrylovJL_GMRES()), abstol=1e-10)
    assert sol.retcode == "Success" == R.RETCODE_NAMES[ref.retcode] and sol.stats.nsteps == ref.stats.nsteps
    assert np.all(np.abs(nf(np.asarray(sol.u.cpu()))) < 1e-9)
    assert np.max(np.abs(np.asarray(sol.u.cpu()) - ref.u)) < 1e-6


def test_levenberg_marquardt_reinit_restores_the_damping(nls):
    """reinit!: λ back to damping_initial, DᵀD back to min_damping_D, v_cache = u0 (levenberg_marquardt.jl:119-131,235-245):
    a second solve from the same start repeats the first one step for step."""
    _, dalg = _algs(nls, True)
    u0 = np.asarray(R.Brusselator2D(5).u0())
    c = nls.init(nls.NonlinearProblem(nls.Brusselator2D(5), u0=u0.copy()), dalg, abstol=1e-10, maxiters=100, store_trace=True)
    assert c.trust_region == 1.0
    s1 = c.solve()
    lam1 = [t["trust_region"] for t in s1.trace]
    assert c.trust_region != 1.0
    c.reinit()
    assert c.trust_region == 1.0 and c.nsteps == 0
    c.reinit(u0.copy())
    s2 = c.solve()
    assert [t["trust_region"] for t in s2.trace] == lam1
    assert np.array_equal(np.asarray(s1.u), np.asarray(s2.u))


@pytest.mark.parametrize("which", ["quad", "bratu12", "brus6"])
@pytest.mark.parametrize("geo", [True, False])
def test_levenberg_marquardt_default_linsolve_matches_oracle(nls, which, geo):
    """`LevenbergMarquardt()` as the reference constructs it (linsolve = nothing): a factorising solver — the reference takes
    the QR least-squares form of the damped step (damped_newton.jl:258-296), the device assembles JᵀJ + λDᵀD on the pattern of
    JᵀJ and factorises it (block cyclic reduction / band LU); same minimiser, so the same λ sequence, steps and iterate as the
    oracle's dense least-squares solve."""
    mk_ref, mk_dev = CASES[which]
    ref = R.solve(mk_ref(), R.LevenbergMarquardt(disable_geodesic=not geo), abstol=1e-9, maxiters=200)
    sol = nls.solve(nls.NonlinearProblem(mk_dev(nls)), nls.LevenbergMarquardt(disable_geodesic=not geo), abstol=1e-9,
                    maxiters=200, store_trace=True)
    assert sol.retcode == R.RETCODE_NAMES[ref.retcode] == "Success"
    assert sol.stats.nsteps == ref.stats.nsteps and sol.stats.nf == ref.stats.nf and sol.stats.njacs == ref.stats.njacs
    assert sol.stats.gmres_iters == 0 and sol.stats.nfactors == sol.stats.nsteps
    assert np.allclose([t["trust_region"] for t in sol.trace], [t["trust_region"] for t in ref.trace], rtol=1e-12)
    assert [t["accepted"] for t in sol.trace] == [t["accepted"] for t in ref.trace]
    assert np.max(np.abs(np.asarray(sol.u) - ref.u)) <= 1e-8 * max(1.0, np.max(np.abs(ref.u)))


def test_gauss_newton_default_linsolve_is_a_plain_direct_solve(nls):
    """GaussNewton() with linsolve = nothing on a square problem: no normal form (needs_square_A(nothing) = false,
    descent/newton.jl:71) — J δ = f through the factorisation, like NewtonRaphson()."""
    ref = R.solve(R.Bratu2D(16), R.GaussNewton(), abstol=1e-9, maxiters=30,
                  termination_kwargs=dict(mode=0, norm="l2", max_stalled_steps=32))   # the least-squares default: 2-norm
    sol = nls.solve(nls.NonlinearLeastSquaresProblem(nls.Bratu2D(16)), nls.GaussNewton(), abstol=1e-9, maxiters=30)
    assert sol.retcode == "Success" == R.RETCODE_NAMES[ref.retcode] and sol.stats.nsteps == ref.stats.nsteps
    assert sol.stats.gmres_iters == 0 and np.max(np.abs(np.asarray(sol.u) - ref.u)) <= 1e-9


def test_kwargs_sets_of_the_reference_tests(nls):
    """rootfind_tests__item11.jl (TrustRegion kwargs) and item18 (LevenbergMarquardt kwargs): the three option sets each, on
    quadratic_f with u0 = [1, 1], p = 2 — err < 1e-9 and the oracle's step counts (every option reaches the device)."""
    u0 = np.array([1.0, 1.0])
    opts = zip([10.0, 100.0, 1000.0], [10.0, 1.0, 0.1], [0.0, 0.01, 0.25], [0.25, 0.3, 0.5], [0.5, 0.8, 0.9], [0.1, 0.3, 0.5],
               [1.5, 2.0, 3.0], [10, 20, 30])
    for mtr, itr, st, sht, et, sf, ef, mst in opts:
        kw = dict(max_trust_radius=mtr, initial_trust_radius=itr, step_threshold=st, shrink_threshold=sht, expand_threshold=et,
                  shrink_factor=sf, expand_factor=ef, max_shrink_times=mst)
        ref = R.solve(R.Quadratic(2, 2.0), R.TrustRegion(**kw), u0=u0)
        sol = nls.solve(nls.NonlinearProblem(nls.Quadratic(2, 2.0), u0=u0), nls.TrustRegion(**kw))
        assert sol.retcode == "Success" == R.RETCODE_NAMES[ref.retcode] and sol.stats.nsteps == ref.stats.nsteps
        assert np.max(np.abs(np.asarray(sol.u) ** 2 - 2.0)) < 1e-9
    opts = zip([0.5, 2.0, 5.0], [1.5, 3.0, 10.0], [2.0, 5.0, 10.0], [0.02, 0.2, 0.3], [0.6, 0.8, 0.9], [0.0, 1.0, 2.0],
               [1e-12, 1e-9, 1e-4])
    for di, dif, ddf, fd, ag, bu, md in opts:
        kw = dict(damping_initial=di, damping_increase_factor=dif, damping_decrease_factor=ddf, finite_diff_step_geodesic=fd,
                  alpha_geodesic=ag, b_uphill=bu, min_damping_D=md)
        ref = R.solve(R.Quadratic(2, 2.0), R.LevenbergMarquardt(**kw), u0=u0, maxiters=10000)
        sol = nls.solve(nls.NonlinearProblem(nls.Quadratic(2, 2.0), u0=u0), nls.LevenbergMarquardt(**kw), maxiters=10000)
        # (the device factorises the damped normal equations where the oracle solves the least-squares form: at the default
        #  abstol the last step or two sit on the rounding floor — 17 against 16 steps on the first set)
        assert sol.retcode == "Success" == R.RETCODE_NAMES[ref.retcode] and abs(sol.stats.nsteps - ref.stats.nsteps) <= 3
        assert np.max(np.abs(np.asarray(sol.u) ** 2 - 2.0)) < 1e-9
