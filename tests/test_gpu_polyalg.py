"""Polyalgorithms over the device caches vs the oracle's restatement of NonlinearSolvePolyAlgorithm
(lib/NonlinearSolveBase/src/polyalg.jl, solve.jl:465-790) and of RobustMultiNewton / FastShortcutNLLSPolyalg
(lib/NonlinearSolveFirstOrder/src/poly_algs.jl); the cases of test/PolyAlgorithms/core_tests__item2.jl and
test/Core/polyalg_retention_tests__item1.jl with first-order rungs."""
import numpy as np
import pytest
import scipy.sparse as sp

from oracle import reference_restatement as R

pytestmark = pytest.mark.gpu
ROOT = 2.0 ** (1.0 / 3.0)


def _same_stats(a, b):
    return (a.nf, a.njacs, a.nfactors, a.nsolve, a.nsteps) == (b.nf, b.njacs, b.nfactors, b.nsolve, b.nsteps)


def _cubic_ref(u0):
    return R.FunctionProblem(lambda u: u ** 3 - 2.0, u0, jac=lambda u: sp.diags(3.0 * u * u))


def _cubic_dev(nls, dev, u0):
    import torch
    n = len(u0)
    proto = nls.CSRMatrix.from_scipy(sp.identity(n, format="csr"))

    def f(du, u, p):
        du.copy_(u ** 3 - 2.0)

    def jvp(Jv, v, u, p):
        Jv.copy_(3.0 * u * u * v)

    def jac(nzval, u, p):
        nzval.copy_(3.0 * u * u)

    return nls.NonlinearProblem(nls.NonlinearFunction(f, jvp=jvp, vjp=jvp, jac=jac, jac_prototype=proto),
                                torch.tensor(np.asarray(u0, dtype=float), device=dev))


@pytest.mark.parametrize("lin", ["direct", "krylov"])
@pytest.mark.parametrize("poly", ["RobustMultiNewton", "FastShortcutNLLSPolyalg"])
def test_polyalgorithms_quadratic_three_interfaces(nls, poly, lin):
    """core_tests__item2.jl: direct solve, caching interface (+ reinit!), step interface — f(u) = u² − 2, abstol 1e-9."""
    rls, dls = (None, None) if lin == "direct" else (R.KrylovJL_GMRES(), nls.KrylovJL_GMRES())
    ralg, dalg = getattr(R, poly)(linsolve=rls), getattr(nls, poly)(linsolve=dls)
    ref = R.solve(R.Quadratic(2, 2.0), ralg, abstol=1e-9)
    sol = nls.solve(nls.NonlinearProblem(nls.Quadratic(2, 2.0)), dalg, abstol=1e-9)
    assert sol.retcode == "Success" and np.max(np.abs(np.asarray(sol.u) ** 2 - 2.0)) < 1e-9 and _same_stats(sol.stats, ref.stats)
    c = nls.init(nls.NonlinearProblem(nls.Quadratic(2, 2.0)), dalg, abstol=1e-9)
    rc = R.init(R.Quadratic(2, 2.0), ralg, abstol=1e-9)
    s, rs = nls.solve_(c), rc.solve()
    assert s.retcode == "Success" and c.best == rc.best == 1 and _same_stats(s.stats, rs.stats)
    nls.reinit_(c, np.array([1.0, 1.0]))
    assert nls.solve_(c).retcode == "Success"
    c.close()
    c = nls.init(nls.NonlinearProblem(nls.Quadratic(2, 2.0)), dalg, abstol=1e-9)
    for _ in range(10000):
        c.step()
        if c.force_stop:
            break
    assert c.retcode == "Success" and c.nsteps == rs.stats.nsteps
    c.close()


def test_ladder_escalates_on_pde_problems(nls):
    """A ladder whose first rungs fail on a real problem: too few iterations for plain Newton on Bratu, a line search that
    is then enough — rung by rung the oracle's ladder (winner, summed statistics, solution)."""
    kw = dict(gmres_restart=30, maxiters=300)
    def ladder(M):
        return M.NonlinearSolvePolyAlgorithm((
            M.NewtonRaphson(linsolve=M.KrylovJL_GMRES(**kw)),
            M.TrustRegion(linsolve=M.KrylovJL_GMRES(**kw)),
            M.NewtonRaphson(linsolve=M.KrylovJL_GMRES(**kw), linesearch=M.BackTracking())))
    for mk_ref, mk_dev, maxit in ((lambda: R.Bratu2D(16), lambda: nls.Bratu2D(16), 3), (lambda: R.Brusselator2D(8), lambda: nls.Brusselator2D(8), 4)):
        ref = R.solve(mk_ref(), ladder(R), abstol=1e-9, maxiters=maxit)
        sol = nls.solve(nls.NonlinearProblem(mk_dev()), ladder(nls), abstol=1e-9, maxiters=maxit)
        assert sol.retcode == R.RETCODE_NAMES[ref.retcode] and _same_stats(sol.stats, ref.stats)
        assert np.max(np.abs(np.asarray(sol.u) - ref.u)) <= 1e-7 * max(1.0, np.max(np.abs(ref.u)))
        rc = R.init(mk_ref(), ladder(R), abstol=1e-9, maxiters=maxit)
        c = nls.init(nls.NonlinearProblem(mk_dev()), ladder(nls), abstol=1e-9, maxiters=maxit)
        rs, s = rc.solve(), nls.solve_(c)
        assert c.best == rc.best and c.current == rc.current and s.retcode == R.RETCODE_NAMES[rs.retcode]
        assert _same_stats(s.stats, rs.stats) and [x.nsteps for x in c.caches] == [x.nsteps for x in rc.caches]
        c.close()


def test_polyalgorithm_retention(nls, dev):
    """polyalg_retention_tests__item1.jl: sticky start on the last winner with lazy sub-cache reinit, escalation, wrap-around
    floored at start_index, the re-probe on the 8th retained reinit!, and the status-quo restart when retention is off."""
    NRr, PTr = R.NewtonRaphson(), R.PseudoTransient(alpha_initial=1.0)
    NRd, PTd = nls.NewtonRaphson(), nls.PseudoTransient(alpha_initial=1.0)
    rc = R.init(_cubic_ref([0.0]), R.NonlinearSolvePolyAlgorithm((NRr, PTr)))
    c = nls.init(_cubic_dev(nls, dev, [0.0]), nls.NonlinearSolvePolyAlgorithm((NRd, PTd)))
    s1, r1 = nls.solve_(c), rc.solve()
    assert s1.retcode == "Success" and c.best == rc.best == 2 and c.caches[0].nsteps == rc.caches[0].nsteps > 0
    assert _same_stats(s1.stats, r1.stats) and abs(float(s1.u[0]) - ROOT) < 1e-6
    n_newton, nf_newton = c.caches[0].nsteps, c.caches[0].stats.nf
    nls.reinit_(c, np.array([1.2]), retain_best=True); rc.reinit(np.array([1.2]), retain_best=True)
    assert c.current == rc.current == 2
    s2, r2 = nls.solve_(c), rc.solve()
    assert s2.retcode == "Success" and abs(float(s2.u[0]) - ROOT) < 1e-6 and c.best == 2 and _same_stats(s2.stats, r2.stats)
    assert c.caches[0].nsteps == n_newton and c.caches[0].stats.nf == nf_newton      # Newton's cache was not touched again
    c.close()
    # escalation from a sticky rung that now fails
    c = nls.init(_cubic_dev(nls, dev, [100.0]), nls.NonlinearSolvePolyAlgorithm((NRd, PTd)))
    assert nls.solve_(c).retcode == "Success" and c.best == 1
    nls.reinit_(c, np.array([0.0]), retain_best=True)
    assert c.current == 1
    s = nls.solve_(c)
    assert s.retcode == "Success" and abs(float(s.u[0]) - ROOT) < 1e-6 and c.best == 2
    c.close()
    # wrap-around to the skipped cheaper rung; floored at start_index
    c = nls.init(_cubic_dev(nls, dev, [0.0]), nls.NonlinearSolvePolyAlgorithm((PTd, NRd)))
    c.best = 2
    nls.reinit_(c, np.array([0.0]), retain_best=True)
    s = nls.solve_(c)
    assert c.wrapped and s.retcode == "Success" and c.best == 1 and abs(float(s.u[0]) - ROOT) < 1e-6
    c.close()
    c = nls.init(_cubic_dev(nls, dev, [0.0]), nls.NonlinearSolvePolyAlgorithm((PTd, NRd), start_index=2))
    c.best = 2
    nls.reinit_(c, np.array([0.0]), retain_best=True)
    s = nls.solve_(c)
    assert not c.wrapped and s.retcode != "Success" and c.caches[0].nsteps == 0
    c.close()
    # periodic re-probe
    c = nls.init(_cubic_dev(nls, dev, [1.2]), nls.NonlinearSolvePolyAlgorithm((PTd, NRd)))
    c.best = 2
    for _ in range(7):
        nls.reinit_(c, np.array([1.2]), retain_best=True)
        assert c.current == 2
    nls.reinit_(c, np.array([1.2]), retain_best=True)
    assert c.current == 1
    assert nls.solve_(c).retcode == "Success" and c.best == 1
    c.close()
    # retention off: full restart, every sub-cache reinitialised
    c = nls.init(_cubic_dev(nls, dev, [0.0]), nls.NonlinearSolvePolyAlgorithm((NRd, PTd)))
    assert nls.solve_(c).retcode == "Success" and c.best == 2
    nls.reinit_(c, np.array([1.2]))
    assert c.current == 1 and not c.retain_best and c.caches[0].nsteps == c.caches[1].nsteps == 0
    s = nls.solve_(c)
    assert s.retcode == "Success" and c.best == 1 and abs(float(s.u[0]) - ROOT) < 1e-6
    c.close()


def test_all_rungs_fail_returns_lowest_residual(nls, dev):
    ralg = R.NonlinearSolvePolyAlgorithm((R.NewtonRaphson(), R.PseudoTransient(alpha_initial=1e-3)))
    dalg = nls.NonlinearSolvePolyAlgorithm((nls.NewtonRaphson(), nls.PseudoTransient(alpha_initial=1e-3)))
    for oneshot in (True, False):
        if oneshot:
            ref = R.solve(_cubic_ref([0.0]), ralg, maxiters=3)
            sol = nls.solve(_cubic_dev(nls, dev, [0.0]), dalg, maxiters=3)
        else:
            ref = R.init(_cubic_ref([0.0]), ralg, maxiters=3).solve()
            c = nls.init(_cubic_dev(nls, dev, [0.0]), dalg, maxiters=3)
            sol = nls.solve_(c)
            c.close()
        assert sol.retcode == "MaxIters" == R.RETCODE_NAMES[ref.retcode] and _same_stats(sol.stats, ref.stats)
        assert abs(float(sol.u[0]) - ref.u[0]) <= 1e-12 and abs(float(sol.resid[0]) - ref.resid[0]) <= 1e-10


def test_least_squares_problem_type_drives_the_norms(nls):
    """A NonlinearLeastSquaresProblem is a type of its own: every algorithm's default termination measures the residual in
    the 2-norm (default_termination_mode(::NonlinearLeastSquaresProblem)) and a polyalgorithm's best-of fallback ranks its
    rungs by the 2-norm (findmin_resids with least squares, NonlinearSolveBase/src/polyalg.jl) — LM and the ladders, not only
    Gauss–Newton."""
    assert issubclass(nls.NonlinearLeastSquaresProblem, nls.NonlinearProblem) and nls.NonlinearLeastSquaresProblem is not nls.NonlinearProblem
    tk = dict(mode=0, norm="l2", max_stalled_steps=32)
    lm, rlm = nls.LevenbergMarquardt(linsolve=nls.KrylovJL_GMRES()), R.LevenbergMarquardt(linsolve=R.KrylovJL_GMRES())
    for ab in (1e-3, 1e-5, 1e-7):   # (‖f‖₂ = 20 ‖f‖∞ here: the 2-norm needs one step more at each of these tolerances)
        ref2 = R.solve(R.Quadratic(400, 2.0), rlm, abstol=ab, maxiters=200, termination_kwargs=tk)
        refi = R.solve(R.Quadratic(400, 2.0), rlm, abstol=ab, maxiters=200)
        s2 = nls.solve(nls.NonlinearLeastSquaresProblem(nls.Quadratic(400, 2.0)), lm, abstol=ab, maxiters=200)
        si = nls.solve(nls.NonlinearProblem(nls.Quadratic(400, 2.0)), lm, abstol=ab, maxiters=200)
        assert s2.retcode == si.retcode == "Success"
        assert s2.stats.nsteps == ref2.stats.nsteps == refi.stats.nsteps + 1 == si.stats.nsteps + 1
        assert np.linalg.norm(np.asarray(s2.resid)) <= ab and np.max(np.abs(np.asarray(si.resid))) <= ab
    # a ladder none of whose rungs can finish: the winner is the oracle's under the 2-norm ranking, through both entries
    kw = dict(gmres_restart=30, maxiters=300)
    def ladder(M):
        return M.NonlinearSolvePolyAlgorithm((M.NewtonRaphson(linsolve=M.KrylovJL_GMRES(**kw)), M.TrustRegion(linsolve=M.KrylovJL_GMRES(**kw)),
                                              M.LevenbergMarquardt(linsolve=M.KrylovJL_GMRES(**kw))))
    ref = R.solve(R.Brusselator2D(6), ladder(R), abstol=1e-12, maxiters=2, least_squares=True, termination_kwargs=tk)
    sol = nls.solve(nls.NonlinearLeastSquaresProblem(nls.Brusselator2D(6)), ladder(nls), abstol=1e-12, maxiters=2)
    assert sol.retcode == R.RETCODE_NAMES[ref.retcode] != "Success" and _same_stats(sol.stats, ref.stats)
    assert np.max(np.abs(np.asarray(sol.u) - ref.u)) <= 1e-8 * max(1.0, np.max(np.abs(ref.u)))
    c = nls.init(nls.NonlinearLeastSquaresProblem(nls.Brusselator2D(6)), ladder(nls), abstol=1e-12, maxiters=2)
    rc = R.init(R.Brusselator2D(6), ladder(R), abstol=1e-12, maxiters=2, least_squares=True, termination_kwargs=tk)
    assert c.least_squares and all(cc._opts.termination_norm == 1 for cc in c.caches)
    s, rs = nls.solve_(c), rc.solve()
    assert c.best == rc.best and s.retcode == R.RETCODE_NAMES[rs.retcode]
    c.close()
