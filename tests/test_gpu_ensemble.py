"""Ensembles of small systems, one per GPU thread (the reference's kernel-generation tutorial,
docs/src/tutorials/nonlinear_solve_gpus.md:70-176): hiprtc-compiled residual + dual-number Jacobian + pivoted LU,
SimpleNewtonRaphson semantics, compared system by system with the oracle's restatement of raphson.jl."""
import numpy as np
import pytest

import ensemble_sources as E
from oracle import reference_restatement as R

pytestmark = pytest.mark.gpu


def _oracle(f, jac, u0, P, **kw):
    out = [R.simple_newton_raphson(f, jac, u0 if u0.ndim == 1 else u0[b], P[b], **kw) for b in range(P.shape[0])]
    return (np.array([o[0] for o in out]), np.array([o[1] for o in out]), np.array([o[2] for o in out]),
            np.array([o[3] for o in out]))


def test_quadratic_parameter_sweep(nls):
    """f(u,p) = u.*u .- p over 1000 parameters (nonlinear_solve_gpus.md:47-62): u ≈ sqrt.(p), every system Success."""
    nb, n = 1000, 3
    P = np.repeat(np.arange(1.0, nb + 1.0)[:, None], n, axis=1)
    prob = nls.ImmutableNonlinearProblem(E.QUADRATIC, np.ones(n), P)
    sol = nls.vectorized_solve(prob, nls.SimpleNewtonRaphson())
    assert (sol.retcode == "Success").all()
    assert np.max(np.abs(sol.u - np.sqrt(P)) / np.sqrt(P)) < 1e-12
    xo, fo, rco, ito = _oracle(E.quadratic_f, E.quadratic_jac, np.ones(n), P)
    assert (sol.iters == ito).all()                       # same number of Newton steps per system
    assert np.max(np.abs(sol.u - xo)) <= 1e-13 * np.max(np.abs(xo))
    assert np.max(np.abs(sol.resid - fo)) <= 1e-9         # residual of the previous iterate, as the reference returns


def test_tutorial_p2_problem_vs_oracle(nls, dev):
    """p2_f with u0 = (1,2,3,4) and 1024 random parameter sets (nonlinear_solve_gpus.md:120-131), device-resident
    inputs. The problem has a rank-deficient Jacobian at its roots (squared equations): convergence is linear, so the
    check is parity with the oracle — retcodes, iteration counts and iterates — under the same maxiters."""
    import torch
    rng = np.random.default_rng(7)
    P = rng.random((1024, 4)) + 0.05
    u0 = np.array([1.0, 2.0, 3.0, 4.0])
    prob = nls.ImmutableNonlinearProblem(E.P2, torch.tensor(u0, device=dev), torch.tensor(P, device=dev))
    sol = nls.vectorized_solve(prob, nls.SimpleNewtonRaphson(), maxiters=200)
    xo, fo, rco, ito = _oracle(E.p2_f, E.p2_jac, u0, P, maxiters=200)
    rc = sol.retcode_raw
    assert (rc == rco).mean() > 0.99                      # borderline systems may stop one step apart
    same = (rc == rco) & (np.abs(sol.iters - ito) == 0)
    assert same.mean() > 0.95
    u = sol.u.cpu().numpy()
    assert np.max(np.abs(u[same] - xo[same])) < 1e-6
    assert np.max(np.abs(sol.resid.cpu().numpy()[rc == 1])) <= np.finfo(float).eps ** 0.8


def test_analytic_jacobian_and_per_system_u0(nls):
    import numpy as np
    rng = np.random.default_rng(3)
    nb = 300
    utrue = rng.uniform(0.2, 1.2, (nb, 3))
    P = np.array([E.trig_f(u, np.zeros(3)) for u in utrue])          # p chosen so that utrue is a root
    u0 = utrue + 0.05 * rng.standard_normal((nb, 3))
    xo, fo, rco, ito = _oracle(E.trig_f, E.trig_jac, u0, P)
    for alg in (nls.SimpleNewtonRaphson(jac=True), nls.SimpleNewtonRaphson()):   # analytic nk_jac | dual numbers
        sol = nls.vectorized_solve(nls.ImmutableNonlinearProblem(E.TRIG_WITH_JAC, u0, P), alg)
        assert (sol.retcode == "Success").all() and (rco == R.SUCCESS).all()
        assert (sol.iters == ito).all() and np.max(np.abs(sol.u - xo)) < 1e-12
        assert np.max(np.abs([E.trig_f(u, p) for u, p in zip(sol.u, P)])) < 1e-12
    assert np.mean(np.max(np.abs(sol.u - utrue), axis=1) < 1e-10) > 0.85   # (some starts reach another root — on the oracle too)


def test_shortcut_maxiters_and_nan(nls):
    n = 2
    P = np.array([[9.0, 9.0], [2.0, 2.0], [2.0, 2.0]])
    u0 = np.array([[3.0, 3.0], [0.0, 0.0], [1.0, 1.0]])     # exact root | singular J (NaN forever) | regular
    sol = nls.vectorized_solve(nls.ImmutableNonlinearProblem(E.QUADRATIC, u0, P), maxiters=9)
    assert list(sol.retcode) == ["Success", "MaxIters", "Success"]
    assert list(sol.iters) == [0, 9, R.simple_newton_raphson(E.quadratic_f, E.quadratic_jac, u0[2], P[2])[3]]
    assert np.isnan(sol.u[1]).all()


def test_larger_systems_scratch_path(nls):
    """n = 24 > 8 takes the looped (non-unrolled) code path with the matrix in scratch memory."""
    nb, n = 257, 24
    rng = np.random.default_rng(11)
    P = rng.uniform(1.0, 5.0, (nb, n))
    sol = nls.vectorized_solve(nls.ImmutableNonlinearProblem(E.QUADRATIC, np.ones(n), P))
    assert (sol.retcode == "Success").all() and np.max(np.abs(sol.u - np.sqrt(P))) < 1e-12


NEWTON_FAILS = """
template <typename T> __device__ void nk_f(const T *u, const double *p, T *f) {
  const T a = 0.21640425613334457 + 216.40425613334457 / (1.0 + 0.0006250000000000001 * (u[0] * u[0]));
  const T b = 0.21640425613334457 + 216.40425613334457 / (1.0 + a * a);
  f[0] = 0.010000000000000002 + 10.000000000000002 / (1.0 + b * b) - 0.0011552453009332421 * u[0] - p[0];
}
"""


def _newton_fails(u, p):
    a = 0.21640425613334457 + 216.40425613334457 / (1 + 0.0006250000000000001 * u ** 2.0)
    b = 0.21640425613334457 + 216.40425613334457 / (1 + a ** 2.0)
    return 0.010000000000000002 + 10.000000000000002 / (1 + b ** 2.0) - 0.0011552453009332421 * u - p


def test_simple_trust_region_newton_fails_fixture(nls):
    """rootfind_tests__item10.jl: `newton_fails` must converge with a trust-region method — here every component is its own
    scalar system of an ensemble (the function is separable), solved by SimpleTrustRegion on the device; plain
    SimpleNewtonRaphson does not converge from all seven starts. Iteration counts and iterates equal the oracle's."""
    u0 = np.array([-10.0, -1.0, 1.0, 2.0, 3.0, 4.0, 10.0])[:, None]
    P = np.zeros((7, 1))
    prob = nls.ImmutableNonlinearProblem(NEWTON_FAILS, u0, P)
    tr = nls.vectorized_solve(prob, nls.SimpleTrustRegion(), abstol=1e-9, maxiters=1000)
    assert (tr.retcode == "Success").all() and np.max(np.abs(_newton_fails(tr.u, 0.0))) < 1e-9
    jac = lambda u, p: np.array([[(_newton_fails(u[0] + 1e-7, p[0]) - _newton_fails(u[0] - 1e-7, p[0])) / 2e-7]])
    ref = [R.simple_trust_region(lambda u, p: _newton_fails(u, p), jac, u0[b], P[b], abstol=1e-9) for b in range(7)]
    assert all(r[2] == R.SUCCESS for r in ref)
    assert np.max(np.abs(tr.u[:, 0] - np.array([r[0][0] for r in ref]))) < 1e-6     # same roots (FD Jacobian on the oracle)
    nr = nls.vectorized_solve(prob, nls.SimpleNewtonRaphson(), abstol=1e-9, maxiters=200)
    assert not (nr.retcode == "Success").all()


def test_simple_trust_region_vs_oracle_on_coupled_systems(nls):
    """SimpleTrustRegion with the dual-number Jacobian on the 3-unknown transcendental system from rough starts: retcodes,
    iteration counts and iterates follow the oracle's restatement of trust_region.jl (including its handling of rejected
    trials)."""
    rng = np.random.default_rng(5)
    nb = 200
    utrue = rng.uniform(0.2, 1.2, (nb, 3))
    P = np.array([E.trig_f(u, np.zeros(3)) for u in utrue])
    u0 = utrue + 0.8 * rng.standard_normal((nb, 3))
    sol = nls.vectorized_solve(nls.ImmutableNonlinearProblem(E.TRIG_WITH_JAC, u0, P), nls.SimpleTrustRegion(), maxiters=300)
    ref = [R.simple_trust_region(E.trig_f, E.trig_jac, u0[b], P[b], maxiters=300) for b in range(nb)]
    rco, ito = np.array([r[2] for r in ref]), np.array([r[3] for r in ref])
    xo = np.array([r[0] for r in ref])
    same = (sol.retcode_raw == rco) & (sol.iters == ito)
    assert same.mean() > 0.97                       # chaotic far-from-root paths may split at a rounding-level tie
    assert np.nanmax(np.abs(sol.u[same] - xo[same])) < 1e-8
    ok = sol.retcode_raw == 1
    assert ok.mean() > 0.8 and np.max(np.abs(sol.resid[ok])) <= np.finfo(float).eps ** 0.8


DENSE_COUPLED = """
template <typename T> __device__ void nk_f(const T *u, const double *p, T *f) {
  T s = u[0];
  for (int i = 1; i < NK_N; ++i) s = s + u[i];
  for (int i = 0; i < NK_N; ++i) f[i] = u[i] * u[i] - p[i] + (0.1 / NK_N) * s + 0.05 * u[(i + 1) % NK_N] * u[i];
}
"""


@pytest.mark.parametrize("n", [9, 16, 33, 64])
def test_medium_systems_one_per_wavefront_vs_oracle(nls, n):
    """8 < n ≤ 64: one system per wavefront — lane j owns column j of the dual-number Jacobian, LU with column pivoting
    through v_readlane. Dense coupled residual; retcodes, iteration counts and solutions equal the oracle's
    SimpleNewtonRaphson (row-pivoted LAPACK solve) to 1e-10."""
    rng = np.random.default_rng(n)
    nb = 130                                   # not a multiple of the 4 systems per workgroup
    P = rng.uniform(1.0, 4.0, (nb, n))
    u0 = rng.uniform(0.5, 2.0, (nb, n))

    def f(u, p):
        return u * u - p + (0.1 / n) * u.sum() + 0.05 * np.roll(u, -1) * u

    def jac(u, p):
        J = np.diag(2.0 * u + 0.05 * np.roll(u, -1)) + (0.1 / n) * np.ones((n, n))
        for i in range(n):
            J[i, (i + 1) % n] += 0.05 * u[i]
        return J

    sol = nls.vectorized_solve(nls.ImmutableNonlinearProblem(DENSE_COUPLED, u0, P), nls.SimpleNewtonRaphson(), maxiters=100)
    ref = [R.simple_newton_raphson(f, jac, u0[b], P[b], maxiters=100) for b in range(nb)]
    assert (sol.retcode_raw == np.array([r[2] for r in ref])).all() and (sol.retcode == "Success").all()
    assert (sol.iters == np.array([r[3] for r in ref])).all()
    assert np.max(np.abs(sol.u - np.array([r[0] for r in ref]))) < 1e-10
    assert np.max(np.abs([f(u, p) for u, p in zip(sol.u, P)])) < 1e-11
