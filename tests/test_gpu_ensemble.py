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
