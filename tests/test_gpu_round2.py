"""GPU parity tests added in round 2 (through the C ABI, against the oracle):
  * the deferred residual — step!(…; evaluate_residual = false), supports_deferred_residual, refresh_residual!
    (lib/NonlinearSolveFirstOrder/src/solve.jl:303-340,448-452; test: misc_tests__item7.jl)
  * the retcodes the round-1 suite never produced on the device: Stalled (step-norm stall and patience stall,
    termination_conditions.jl:286-316), ShrinkThresholdExceeded (FirstOrder/src/solve.jl:431-435),
    InternalLinearSolveFailed with and without the recompute-the-Jacobian retry (solve.jl:367-382)
  * the trust-region step after its scalar work was fused (one trial-point kernel, one multi-reduction, one fetch):
    accept/reject sequence, radii and iterates vs the oracle for every radius-update scheme."""
import numpy as np
import pytest

from oracle import reference_restatement as R

pytestmark = pytest.mark.gpu


def _cubic(nls, dev, calls):
    import torch

    def f(du, u, p):
        calls[0] += 1
        du.copy_(u ** 3)

    def jvp(Jv, v, u, p):
        Jv.copy_(3.0 * u * u * v)

    return nls.NonlinearProblem(nls.NonlinearFunction(f, jvp=jvp), torch.ones(1, dtype=torch.float64, device=dev))


def _cubic_oracle(calls):
    import scipy.sparse as sp

    def f(u):
        calls[0] += 1
        return u ** 3
    return R.FunctionProblem(f, np.array([1.0]), jac=lambda u: sp.diags(3.0 * u ** 2))


def test_deferred_residual_offered_only_where_unobservable(nls, dev):
    calls = [0]
    G = nls.KrylovJL_GMRES
    absnorm = nls.AbsNormTerminationMode()
    assert nls.init(_cubic(nls, dev, calls), nls.NewtonRaphson(linsolve=G()), termination_condition=absnorm).supports_deferred_residual()
    assert nls.init(_cubic(nls, dev, calls), nls.NewtonRaphson(linsolve=G()),
                    termination_condition=nls.AbsTerminationMode()).supports_deferred_residual()
    refusing = [nls.init(_cubic(nls, dev, calls), nls.NewtonRaphson(linsolve=G())),
                nls.init(_cubic(nls, dev, calls), nls.TrustRegion(linsolve=G()), termination_condition=absnorm),
                nls.init(_cubic(nls, dev, calls), nls.NewtonRaphson(linsolve=G(), linesearch=nls.BackTracking()),
                         termination_condition=absnorm),
                nls.init(_cubic(nls, dev, calls), nls.NewtonRaphson(linsolve=G()), termination_condition=absnorm, store_trace=True)]
    for c in refusing:
        assert not c.supports_deferred_residual()
        calls[0] = 0
        assert c.refresh_residual() is None and calls[0] == 0


def test_deferred_step_that_would_be_misread_evaluates_anyway(nls, dev):
    calls = [0]
    c = nls.init(_cubic(nls, dev, calls), nls.NewtonRaphson(linsolve=nls.KrylovJL_GMRES()), abstol=1e-300, maxiters=40)
    while not c.force_stop and c.nsteps < 40:
        c.step(evaluate_residual=False)
        c.refresh_residual()
    assert c.nsteps == 40 and c.retcode != "Stalled"


def test_deferred_step_skips_one_residual_and_refresh_pays_it(nls, dev):
    calls = [0]
    mk = lambda: nls.init(_cubic(nls, dev, calls), nls.NewtonRaphson(linsolve=nls.KrylovJL_GMRES()),  # noqa: E731
                          termination_condition=nls.AbsNormTerminationMode())
    c = mk()
    calls[0] = 0
    c.step(evaluate_residual=False)
    assert calls[0] == 0
    c.refresh_residual()
    assert calls[0] == 1
    c.refresh_residual()
    assert calls[0] == 1
    c2 = mk()
    c2.solve()
    calls[0] = 0
    c2.step(evaluate_residual=False)
    c2.refresh_residual()
    assert calls[0] == 0


def test_deferral_does_not_move_the_iterates(nls, dev):
    calls, ocalls = [0], [0]
    mk = lambda: nls.init(_cubic(nls, dev, calls), nls.NewtonRaphson(linsolve=nls.KrylovJL_GMRES()),  # noqa: E731
                          termination_condition=nls.AbsNormTerminationMode())
    plain, deferred = mk(), mk()
    ref = R.init(_cubic_oracle(ocalls), R.NewtonRaphson(), termination_kwargs=dict(mode=R.TM_ABSNORM, max_stalled_steps=None),
                 store_trace=False)
    for _ in range(40):
        plain.step()
        deferred.step(evaluate_residual=False)
        deferred.refresh_residual()
        ref.step(evaluate_residual=False)
        ref.refresh_residual()
        up, ud = plain.u.cpu().numpy(), deferred.u.cpu().numpy()
        assert np.array_equal(up, ud) and np.array_equal(plain.fu.cpu().numpy(), deferred.fu.cpu().numpy())
        assert plain.retcode == deferred.retcode == R.RETCODE_NAMES[ref.retcode]
        assert abs(ud[0] - ref.u[0]) <= 1e-13 and plain.nsteps == deferred.nsteps == ref.nsteps
    assert plain.force_stop and plain.stats.nf == deferred.stats.nf == ref.stats.nf


# ----------------------------------------------------------------------------- retcodes
def _lying_jacobian_problem(nls, dev, fconst=None):
    """δu = −f / 1e12: the iterate barely moves, the residual stays where it is."""
    import torch

    def f(du, u, p):
        if fconst is None:
            du.copy_(u * u - 2.0)
        else:
            du.copy_(0.0 * u + fconst + 1e-30 * u)

    def jvp(Jv, v, u, p):
        Jv.copy_(1e12 * v)

    return nls.NonlinearProblem(nls.NonlinearFunction(f, jvp=jvp), torch.ones(4, dtype=torch.float64, device=dev))


def test_stalled_by_step_norm(nls, dev):
    """32 consecutive steps with ‖u − u_prev‖₂ ≤ abstol while ‖f‖∞ > abstol ⇒ Stalled on step 33
    (termination_conditions.jl:303-316, default max_stalled_steps = 32)."""
    ref = R.solve(R.FunctionProblem(lambda u: u * u - 2.0, np.ones(4), jvp=lambda v, u: 1e12 * v),
                  R.NewtonRaphson(linsolve=R.KrylovJL_GMRES()), abstol=1e-9, maxiters=200)
    sol = nls.solve(_lying_jacobian_problem(nls, dev), nls.NewtonRaphson(linsolve=nls.KrylovJL_GMRES()), abstol=1e-9, maxiters=200)
    assert sol.retcode == "Stalled" == R.RETCODE_NAMES[ref.retcode]
    assert sol.stats.nsteps == ref.stats.nsteps == 33


def test_stalled_by_patience(nls, dev):
    """objective ≤ 3·abstol for more than patience_steps steps with min < 1.3·max ⇒ Stalled (termination_conditions.jl:286-301)."""
    tk = dict(mode=0, max_stalled_steps=None, patience_steps=10)
    ref = R.solve(R.FunctionProblem(lambda u: 0 * u + 2e-9 + 1e-30 * u, np.ones(4), jvp=lambda v, u: 1e12 * v),
                  R.NewtonRaphson(linsolve=R.KrylovJL_GMRES()), abstol=1e-9, maxiters=300, termination_kwargs=tk)
    sol = nls.solve(_lying_jacobian_problem(nls, dev, 2e-9), nls.NewtonRaphson(linsolve=nls.KrylovJL_GMRES()), abstol=1e-9,
                    maxiters=300, termination_kwargs=dict(max_stalled_steps=-1, patience_steps=10))
    assert sol.retcode == "Stalled" == R.RETCODE_NAMES[ref.retcode]
    assert sol.stats.nsteps == ref.stats.nsteps == 11


@pytest.mark.parametrize("mst", [1, 2, 3])
def test_shrink_threshold_exceeded(nls, dev, mst):
    """More than max_shrink_times consecutive shrinks ⇒ ShrinkThresholdExceeded (FirstOrder/src/solve.jl:431-435)."""
    import scipy.sparse as sp
    import torch
    fo = lambda u: np.arctan(5 * u) + 0.5 * np.sin(7 * u)                      # noqa: E731
    dfo = lambda u: 5.0 / (1.0 + 25.0 * u * u) + 3.5 * np.cos(7 * u)           # noqa: E731
    ref = R.solve(R.FunctionProblem(fo, 3 * np.ones(3), jac=lambda u: sp.diags(dfo(u))),
                  R.TrustRegion(linsolve=R.KrylovJL_GMRES(), max_shrink_times=mst), abstol=1e-10, maxiters=100)

    def f(du, u, p):
        du.copy_(torch.atan(5 * u) + 0.5 * torch.sin(7 * u))

    def jv(Jv, v, u, p):   # diagonal Jacobian: JVP and VJP coincide
        Jv.copy_((5.0 / (1.0 + 25.0 * u * u) + 3.5 * torch.cos(7 * u)) * v)

    prob = nls.NonlinearProblem(nls.NonlinearFunction(f, jvp=jv, vjp=jv), 3 * torch.ones(3, dtype=torch.float64, device=dev))
    sol = nls.solve(prob, nls.TrustRegion(linsolve=nls.KrylovJL_GMRES(), max_shrink_times=mst), abstol=1e-10, maxiters=100,
                    store_trace=True)
    assert R.RETCODE_NAMES[ref.retcode] == "ShrinkThresholdExceeded" == sol.retcode
    assert sol.stats.nsteps == ref.stats.nsteps
    assert [t["accepted"] for t in sol.trace] == [t["accepted"] for t in ref.trace]
    assert np.allclose([t["trust_region"] for t in sol.trace], [t["trust_region"] for t in ref.trace], rtol=1e-9)


def _linear_problem(nls, dev, state):
    """W z − b with a user `jac` into a tridiagonal jac_prototype; state['poison'] makes the callback hand out NaN."""
    import scipy.sparse as sp
    import torch
    N = 12
    W = sp.csr_matrix(sp.diags([-np.ones(N - 1), 4.0 * np.ones(N), -np.ones(N - 1)], [-1, 0, 1]))
    bvec = np.arange(1.0, N + 1)
    Wd, proto = nls.CSRMatrix.from_scipy(W), nls.CSRMatrix.from_scipy(W)
    bd, wvals = torch.tensor(bvec, device=dev), torch.tensor(W.data, device=dev)

    def resid(F, z, p):
        Wd.matvec(z, out=F)
        F.add_(0.1 * z ** 3).sub_(bd)

    def jac(nzval, z, p):
        state["jac_calls"] += 1
        if state["poison"]:
            nzval.fill_(float("nan"))
            return
        nzval.copy_(wvals)
        # + diag(0.3 z²): the diagonal is the middle entry of each row of the tridiagonal pattern
        dpos = torch.tensor(np.flatnonzero(W.indices == np.repeat(np.arange(N), np.diff(W.indptr))), device=dev)
        nzval[dpos] += 0.3 * z * z

    f = nls.NonlinearFunction(resid, jac=jac, jac_prototype=proto)
    return nls.NonlinearProblem(f, torch.zeros(N, dtype=torch.float64, device=dev)), proto, W, bvec


def _linear_oracle(W, bvec, state):
    import scipy.sparse as sp

    def jac(u):
        state["jac_calls"] += 1
        J = sp.csr_matrix(W + sp.diags(0.3 * u * u))
        if state["poison"]:
            J = J * np.nan
        return J
    return R.FunctionProblem(lambda u: W @ u + 0.1 * u ** 3 - bvec, np.zeros(len(bvec)), jac=jac)


@pytest.mark.parametrize("lin", ["gmres_concrete", "direct"])
def test_linear_solve_failure_with_fresh_jacobian_stops(nls, dev, lin):
    """A failed linear solve on a Jacobian that was just recomputed ⇒ InternalLinearSolveFailed, force_stop (solve.jl:367-375)."""
    st, so = dict(poison=False, jac_calls=0), dict(poison=False, jac_calls=0)
    prob, proto, W, bvec = _linear_problem(nls, dev, st)
    alg = nls.NewtonRaphson(linsolve=nls.KrylovJL_GMRES(), concrete_jac=True) if lin == "gmres_concrete" else nls.NewtonRaphson()
    ralg = R.NewtonRaphson(linsolve=R.KrylovJL_GMRES(), concrete_jac=True) if lin == "gmres_concrete" else R.NewtonRaphson()
    c, r = nls.init(prob, alg, abstol=1e-10), R.init(_linear_oracle(W, bvec, so), ralg, abstol=1e-10)
    c.step(); r.step()
    st["poison"] = so["poison"] = True
    c.step(); r.step()
    assert c.retcode == "InternalLinearSolveFailed" == R.RETCODE_NAMES[r.retcode]
    assert c.force_stop and r.force_stop and c.nsteps == r.nsteps == 2
    assert st["jac_calls"] == so["jac_calls"]


@pytest.mark.parametrize("retry_works", [True, False])
def test_linear_solve_failure_on_stale_jacobian_retries_with_a_new_one(nls, dev, retry_works):
    """step!(…; recompute_jacobian = false) on a Jacobian that makes the linear solve fail: the step recomputes J and tries
    again (solve.jl:376-382); only a failure on the fresh J ends the solve."""
    st, so = dict(poison=False, jac_calls=0), dict(poison=False, jac_calls=0)
    prob, proto, W, bvec = _linear_problem(nls, dev, st)
    c = nls.init(prob, nls.NewtonRaphson(linsolve=nls.KrylovJL_GMRES(), concrete_jac=True), abstol=1e-10)
    r = R.init(_linear_oracle(W, bvec, so), R.NewtonRaphson(linsolve=R.KrylovJL_GMRES(), concrete_jac=True), abstol=1e-10)
    c.step(); r.step()
    # spoil the STORED Jacobian (the solver keeps the user's jac_prototype object): the stale J is what fails
    proto.set_values(np.full(proto.info()["nnz"], np.nan))
    r.J = r.J * np.nan
    st["poison"] = so["poison"] = not retry_works
    n0, m0 = st["jac_calls"], so["jac_calls"]
    c.step(recompute_jacobian=False); r.step(recompute_jacobian=False)
    assert st["jac_calls"] - n0 == so["jac_calls"] - m0 == 1          # exactly the retry's evaluation
    if retry_works:
        assert c.retcode == R.RETCODE_NAMES[r.retcode] and not c.force_stop
        assert np.max(np.abs(c.u.cpu().numpy() - r.u)) <= 1e-9
        assert c.stats.nsolve == r.stats.nsolve == 3                   # one good solve, the failed one, the retry
    else:
        assert c.retcode == "InternalLinearSolveFailed" == R.RETCODE_NAMES[r.retcode] and c.force_stop


# ----------------------------------------------------------------------------- fused trust-region step vs the oracle
@pytest.mark.parametrize("scheme", ["Simple", "NLsolve", "NocedalWright", "Hei", "Yuan", "Fan", "Bastin"])
@pytest.mark.parametrize("concrete", [False, True])
def test_trust_region_fused_step_matches_oracle(nls, scheme, concrete):
    N = 12
    rs = getattr(nls.RadiusUpdateSchemes, scheme)
    ro = {"Simple": R.SIMPLE, "NLsolve": R.NLSOLVE, "NocedalWright": R.NOCEDAL_WRIGHT, "Hei": R.HEI, "Yuan": R.YUAN,
          "Fan": R.FAN, "Bastin": R.BASTIN}[scheme]
    pb = R.Brusselator2D(N)
    ref = R.solve(pb, R.TrustRegion(linsolve=R.KrylovJL_GMRES(maxiters=600), radius_update_scheme=ro, concrete_jac=concrete),
                  abstol=1e-9, maxiters=60, lin_x0_zero=True)
    P = nls.Brusselator2D(N)
    sol = nls.solve(nls.NonlinearProblem(P, u0=P.initial_guess(device=True)),
                    nls.TrustRegion(linsolve=nls.KrylovJL_GMRES(maxiters=600), radius_update_scheme=rs, concrete_jac=concrete),
                    abstol=1e-9, maxiters=60, store_trace=True)
    assert sol.retcode == R.RETCODE_NAMES[ref.retcode]
    k = min(len(sol.trace), len(ref.trace), 6)
    assert [t["accepted"] for t in sol.trace[:k]] == [t["accepted"] for t in ref.trace[:k]]
    # radii are compared tightly while the residual is far above the inner solves' tolerance; close to convergence the
    # schemes that take the radius from ‖Jᵀf(u_trial)‖ or ‖f(u_trial)‖ (Yuan, Fan) inherit the GMRES tolerance (1e-3 relative)
    for a, b in zip(sol.trace[:k], ref.trace[:k]):
        tol = 1e-6 if b["fnorm_inf"] > 1e-4 else 2e-2
        assert abs(a["trust_region"] - b["trust_region"]) <= tol * abs(b["trust_region"]), (scheme, a, b)
    if sol.retcode == "Success":
        assert np.max(np.abs(np.asarray(sol.u.cpu()) - ref.u)) <= 1e-7 * max(1.0, np.max(np.abs(ref.u)))


# ----------------------------------------------------------------------------- seam 1 from plain C
def test_linsolve_seam_c_caller_matches_oracle(tmp_path):
    """examples/linsolve_seam.c replays the reference's own call sequence against the linsolve seam
    (NonlinearSolveBaseLinearSolveExt.jl:16-32,102-115: a new A every step → update_tolerances! → solve!) with a callback
    operator that receives DEVICE pointers, with the bound device JVP, and with a concrete CSR; vectors stay resident
    (nk_device_alloc, memspace NK_DEVICE). All three variants must reproduce the oracle's NewtonRaphson + GMRES(30) +
    EisenstatWalkerForcing2 solve: same number of steps (±1), iterates within 5e-7 (loose inner solves on both sides)."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = tmp_path / "linsolve_seam"
    libdir = os.path.join(root, "nonlinearsolve.jl_amd", "lib")
    subprocess.check_call(["gcc", "-std=c99", "-I", os.path.join(root, "include"), os.path.join(root, "examples", "linsolve_seam.c"),
                           "-L", libdir, "-lmi355x_nk", "-lm", "-o", str(exe)])
    env = dict(os.environ, LD_LIBRARY_PATH=libdir + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    ns = 48
    out = subprocess.run([str(exe), str(ns), str(tmp_path / "u")], env=env, capture_output=True, text=True, timeout=180)
    assert out.returncode == 0, out.stdout + out.stderr
    ref = R.solve(R.Bratu2D(ns, 6.0), R.NewtonRaphson(linsolve=R.KrylovJL_GMRES(), forcing=R.EisenstatWalkerForcing2()),
                  abstol=1e-8, maxiters=50)
    lines = {l.split(":")[0]: l for l in out.stdout.splitlines() if "steps=" in l}
    assert set(lines) == {"fn", "jvp", "csr", "precs"}, out.stdout   # precs: Pl = device ILU(0), re-evaluated for every new A
    it = {k: int(v.split("gmres_iters=")[1].split()[0]) for k, v in lines.items()}
    assert it["precs"] < it["csr"] / 2                               # the left preconditioner did precondition
    for name, line in lines.items():
        steps = int(line.split("steps=")[1].split()[0])
        # (precs: the forcing tolerance η is tested against the PRECONDITIONED residual ‖Pl⁻¹ r‖ — Krylov.jl's measure [EXT] —, a
        #  looser one for the true residual the nonlinear iteration watches: a few more, cheaper Newton steps to the same root)
        assert (abs(steps - ref.stats.nsteps) <= 1 if name != "precs" else steps <= ref.stats.nsteps + 6) and "failed=0" in line, line
        u = np.fromfile(str(tmp_path / f"u_{name}.bin"))
        assert u.size == ns * ns and np.max(np.abs(u - ref.u)) <= (5e-7 if name != "precs" else 2e-6) * max(1.0, np.max(np.abs(ref.u))), name
    assert int(lines["fn"].split("callback_applies=")[1]) > 0     # the device-pointer callback really carried the solve


# ----------------------------------------------------------------------------- GaussNewton through the normal-form operator
@pytest.mark.parametrize("which", ["bratu8", "brus5", "quad"])
@pytest.mark.parametrize("concrete", [False, True])
def test_gauss_newton_normal_form_matches_oracle(nls, which, concrete):
    """GaussNewton(linsolve = KrylovJL_GMRES()) on a (square) least-squares problem: NewtonDescent in normal form,
    JᵀJ δ = Jᵀ f through the normal-form operator (descent/newton.jl:58-95,107-118; SciMLJacobianOperators.jl:252-291) —
    matrix-free (JVP then VJP) and on the concrete J (SpMV then transposed SpMV)."""
    pb, P = {"bratu8": (R.Bratu2D(8), lambda: nls.Bratu2D(8)), "brus5": (R.Brusselator2D(5), lambda: nls.Brusselator2D(5)),
             "quad": (R.Quadratic(20, 2.0), lambda: nls.Quadratic(20, 2.0))}[which]
    P = P()
    tk = dict(mode=0, norm="l2", max_stalled_steps=32)
    ref = R.solve(pb, R.GaussNewton(linsolve=R.KrylovJL_GMRES(gmres_restart=60, maxiters=600), concrete_jac=concrete),
                  abstol=1e-9, maxiters=50, termination_kwargs=tk)
    prob = nls.NonlinearLeastSquaresProblem(P)
    sol = nls.solve(prob, nls.GaussNewton(linsolve=nls.KrylovJL_GMRES(gmres_restart=60, maxiters=600), concrete_jac=concrete),
                    abstol=1e-9, maxiters=50)
    assert sol.retcode == "Success" == R.RETCODE_NAMES[ref.retcode]
    assert abs(sol.stats.nsteps - ref.stats.nsteps) <= 1
    assert np.max(np.abs(np.asarray(sol.u) - ref.u)) <= 1e-8 * max(1.0, np.max(np.abs(ref.u)))


# ----------------------------------------------------------------------------- prepare_vjp / prepare_jvp fallbacks (a11)
def _nonsym_problem(dev, n):
    """A residual with a non-symmetric tridiagonal Jacobian (so that Jᵀv ≠ Jv): f_i = 3u_i − 2u_{i−1} − 0.5u_{i+1} + 0.1u_i³ − b_i"""
    import torch
    bvec = torch.linspace(0.5, 1.5, n, dtype=torch.float64, device=dev)

    def F(du, u, p):
        du.copy_(3.0 * u + 0.1 * u ** 3 - bvec)
        du[1:] -= 2.0 * u[:-1]
        du[:-1] -= 0.5 * u[1:]

    def J_dense(u):
        return np.diag(3.0 + 0.3 * u * u) - 2.0 * np.diag(np.ones(n - 1), -1) - 0.5 * np.diag(np.ones(n - 1), 1)

    def Fo(u):
        f = 3.0 * u + 0.1 * u ** 3 - np.linspace(0.5, 1.5, n)
        f[1:] -= 2.0 * u[:-1]
        f[:-1] -= 0.5 * u[1:]
        return f
    return F, Fo, J_dense


@pytest.mark.parametrize("with_jac", [True, False])
def test_vjp_and_jvp_fall_back_to_the_jacobian(nls, dev, with_jac):
    """prepare_vjp / prepare_jvp without f.vjp / f.jvp (SciMLJacobianOperators.jl:296-362, 373-431): with f.jac the
    operators are J·v and Jᵀ·v on the jac_prototype pattern; with only a jac_prototype the Jacobian comes from the
    colour-compressed finite-difference assembly (the AutoSparse(AutoFiniteDiff) analogue). TrustRegion — which needs
    Jᵀ fu every step — then runs matrix-free on a problem that supplies neither jvp nor vjp."""
    import scipy.sparse as sp
    import torch
    n = 60
    F, Fo, J_dense = _nonsym_problem(dev, n)
    pat = sp.csr_matrix(sp.diags([np.ones(n - 1), np.ones(n), np.ones(n - 1)], [-1, 0, 1]))
    proto = nls.CSRMatrix.from_scipy(pat)
    dpos = torch.tensor(np.flatnonzero(pat.indices == np.repeat(np.arange(n), np.diff(pat.indptr))), device=dev)
    lpos = torch.tensor(np.flatnonzero(pat.indices == np.repeat(np.arange(n), np.diff(pat.indptr)) - 1), device=dev)
    upos = torch.tensor(np.flatnonzero(pat.indices == np.repeat(np.arange(n), np.diff(pat.indptr)) + 1), device=dev)

    def jac(nzval, u, p):
        nzval[dpos] = 3.0 + 0.3 * u * u
        nzval[lpos] = -2.0
        nzval[upos] = -0.5

    f = nls.NonlinearFunction(F, jac=jac if with_jac else None, jac_prototype=proto)
    prob = nls.NonlinearProblem(f, torch.zeros(n, dtype=torch.float64, device=dev))
    u = np.linspace(-1.0, 1.0, n)
    v = np.cos(np.arange(n))
    ud, vd = torch.tensor(u, device=dev), torch.tensor(v, device=dev)
    Jop = nls.StatefulJacobianOperator(nls.JacobianOperator(prob), ud)
    tol = 1e-12 if with_jac else 1e-6
    assert np.max(np.abs((Jop @ vd).cpu().numpy() - J_dense(u) @ v)) <= (tol if with_jac else 1e-6) * 10
    assert np.max(np.abs((Jop.T @ vd).cpu().numpy() - J_dense(u).T @ v)) <= tol * 10
    ref = R.solve(R.FunctionProblem(Fo, np.zeros(n), jac=lambda w: sp.csr_matrix(J_dense(w))),
                  R.TrustRegion(linsolve=R.KrylovJL_GMRES(gmres_restart=60, maxiters=600)), abstol=1e-10, maxiters=50)
    sol = nls.solve(prob, nls.TrustRegion(linsolve=nls.KrylovJL_GMRES(gmres_restart=60, maxiters=600)), abstol=1e-10, maxiters=50)
    assert sol.retcode == "Success" == R.RETCODE_NAMES[ref.retcode]
    assert abs(sol.stats.nsteps - ref.stats.nsteps) <= 1
    assert np.max(np.abs(sol.u.cpu().numpy() - ref.u)) <= 1e-7


def test_direct_linsolve_falls_back_when_the_band_lu_needs_pivoting(nls, dev):
    """`linsolve = nothing` on a Jacobian whose LU needs row exchanges (zero main diagonal): the reference's default
    solver pivots (and falls back LU → QR when a factorisation is unusable), so the solve succeeds there; the device's band
    LU does not pivot — its a-posteriori check (‖Jx − b‖ ≤ 1e-10‖b‖ after one refinement step) rejects the factorisation
    and GMRES on the same concrete J takes over instead of reporting InternalLinearSolveFailed."""
    import scipy.sparse as sp
    import torch
    n = 40
    W = sp.csr_matrix(sp.diags([np.ones(n - 1), 1e-300 * np.ones(n), np.ones(n - 1)], [-1, 0, 1]))   # pattern incl. diagonal
    Wz = sp.csr_matrix(sp.diags([np.ones(n - 1), np.ones(n - 1)], [-1, 1]))
    bvec = np.arange(1.0, n + 1)
    proto = nls.CSRMatrix.from_scipy(W)
    Wd = nls.CSRMatrix.from_scipy(Wz)
    bd = torch.tensor(bvec, device=dev)
    vals = torch.tensor(sp.csr_matrix(W).data, device=dev).clone()
    vals[torch.tensor(np.flatnonzero(W.indices == np.repeat(np.arange(n), np.diff(W.indptr))), device=dev)] = 0.0

    def resid(F, z, p):
        Wd.matvec(z, out=F)
        F.sub_(bd)

    def jac(nzval, z, p):
        nzval.copy_(vals)

    prob = nls.NonlinearProblem(nls.NonlinearFunction(resid, jac=jac, jac_prototype=proto), torch.zeros(n, dtype=torch.float64, device=dev))
    sol = nls.solve(prob, nls.NewtonRaphson(), abstol=1e-9)
    ref = R.solve(R.FunctionProblem(lambda u: Wz @ u - bvec, np.zeros(n), jac=lambda u: Wz), R.NewtonRaphson(), abstol=1e-9)
    assert sol.retcode == "Success" == R.RETCODE_NAMES[ref.retcode]
    assert sol.stats.nsteps == ref.stats.nsteps
    assert np.max(np.abs(sol.u.cpu().numpy() - ref.u)) <= 1e-8 * np.max(np.abs(ref.u))


def test_speculative_jacobian_fill_is_invisible(nls, dev):
    """Plain Newton on a built-in problem with a concrete J writes the NEXT step's J(u_new) into a second value set while the
    host waits for the step's norms (csrc/nk_solver.hip: speculate_J). Nothing of it may show: after every step the solver's J
    holds the values at the iterate that step STARTED from (as the reference's cache does: J is evaluated at the top of
    `step!`, FirstOrder/src/solve.jl:330-350), the Jacobian count follows the steps, a `reinit!` with new parameters starts from
    a fresh J, and a parameter change between two steps is honoured."""
    import ctypes as C
    import torch
    from nonlinearsolve_jl_amd import _lib as L
    ns = 24
    Pr = R.Bratu2D(ns, 6.0)
    cache = nls.init(nls.NonlinearProblem(nls.Bratu2D(ns, 6.0)), nls.NewtonRaphson(linsolve=nls.KrylovJL_GMRES(), concrete_jac=True),
                     abstol=1e-12, maxiters=50)
    J = nls.CSRMatrix(L.lib().nk_solver_jacobian(cache._h), cache.prob.ctx, owned=False)

    def jac_at(P, u):
        A = P.jac(np.asarray(u)).tocsr()
        A.sort_indices()
        return A.data
    for i in range(4):
        u_prev = np.asarray(cache.u).copy()
        cache.step()
        assert np.max(np.abs(J.values() - jac_at(Pr, u_prev))) <= 1e-13 * np.max(np.abs(J.values())), i
        assert cache.stats.njacs == i + 2          # (one at init: the cache is built with J(u0), FirstOrder/src/solve.jl:171-186)
    # a parameter change between two steps: the set filled ahead (for λ = 6) must not be taken
    u_prev = np.asarray(cache.u).copy()
    cache.prob.device_problem.set_params([ns, 3.0, 0.0])
    cache.step()
    assert np.max(np.abs(J.values() - jac_at(R.Bratu2D(ns, 3.0), u_prev))) <= 1e-13 * np.max(np.abs(J.values()))
    # reinit! (new u0) after a finished solve: same steps and solution as a cache built from scratch
    cache.prob.device_problem.set_params([ns, 6.0, 0.0])
    nls.reinit_(cache, np.zeros(ns * ns))
    s1 = nls.solve_(cache)
    rc = R.init(R.Bratu2D(ns, 6.0), R.NewtonRaphson(linsolve=R.KrylovJL_GMRES(), concrete_jac=True), abstol=1e-12, maxiters=50)
    rc.solve()
    rc.reinit(np.zeros(ns * ns))          # (reinit! does not evaluate a Jacobian: jacobian.jl:184-186)
    ref = rc.solve()
    assert s1.retcode == "Success" == R.RETCODE_NAMES[ref.retcode] and s1.stats.nsteps == ref.stats.nsteps
    assert s1.stats.njacs == ref.stats.njacs and np.max(np.abs(np.asarray(s1.u) - ref.u)) <= 1e-9
    cache.close()


_AHEAD_CODE = (
    "import numpy as np, hashlib, nonlinearsolve_jl_amd as nls\n"
    "ns = 48\n"
    "alg = nls.NewtonRaphson(linsolve=nls.KrylovJL_GMRES(gmres_restart=30, maxiters=30, ortho='sstep', fixed_iters=30), concrete_jac=True)\n"
    "cache = nls.init(nls.NonlinearProblem(nls.Bratu2D(ns, 6.0)), alg, abstol=1e-9, maxiters=50)\n"
    "out = []\n"
    "def h():\n"
    "    u = cache.u\n"
    "    out.append(hashlib.sha256(np.asarray(u.cpu() if hasattr(u, 'cpu') else u).tobytes()).hexdigest()[:16])\n"
    "for _ in range(2): cache.step(); h()\n"
    "cache.prob.device_problem.set_params([ns, 3.0, 0.0])      # the begin that ran ahead was for lambda = 6: must not be taken\n"
    "cache.step(); h()\n"
    "G = nls.GMRES(ns * ns, restart=30)                          # (an unrelated object: nothing to do with the cache's)\n"
    "s = nls.solve_(cache); h()                                  # terminates with a begin ahead in the queue: dropped\n"
    "out.append(s.retcode + str(s.stats.nsteps))\n"
    "cache.prob.device_problem.set_params([ns, 6.0, 0.0])\n"
    "nls.reinit_(cache, np.zeros(ns * ns))\n"
    "s = nls.solve_(cache); h()\n"
    "out.append(s.retcode + str(s.stats.nsteps) + ' ' + str(s.stats.gmres_iters))\n"
    "print('TRACE', ' '.join(out))\n")


def test_cycle_begin_run_ahead_is_invisible():
    """The next linear solve's cycle begin rides in the speculative Jacobian fill (csrc/nk_solver.hip: speculate_J →
    nk_gmres_begin_ahead). Whatever happens between the fill and the solve — nothing, a parameter change (the fill and the begin
    were for the old parameters), a termination with the begin still in the queue, a reinit! — the iterates are those of the
    form that launches the begin with the solve (NK_BEGIN_AHEAD=0): the same arithmetic, the same bits."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tr = {}
    for name, env in (("ahead", {}), ("launched", dict(NK_BEGIN_AHEAD="0")), ("round5", dict(NK_BEGIN_AHEAD="0", NK_FOLD_NORMS="0"))):
        r = subprocess.run([sys.executable, "-c", _AHEAD_CODE], env=dict(os.environ, **env), capture_output=True, text=True, timeout=300,
                           cwd=root)
        assert r.returncode == 0, r.stderr[-2000:]
        tr[name] = [ln for ln in r.stdout.splitlines() if ln.startswith("TRACE")][-1]
    assert tr["ahead"] == tr["launched"] == tr["round5"], tr
    assert "Success" in tr["ahead"]
