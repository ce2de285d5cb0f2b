"""Preconditioning through the reference's `precs(A, p) -> (Pl, Pr)` hook on the device (csrc/nk_precond.hip, the left side of
csrc/nk_gmres.hip, nk_solver_set_precs): ILU(0) / Jacobi objects against the oracle's sequential restatement, left- and
two-sided GMRES against the oracle's, the call protocol of test/Core/core_tests__item21.jl, and the tutorial's
`incompletelu` Newton solve (docs/src/tutorials/large_systems.md:252-260) with a device ILU(0) as Pl."""
import numpy as np
import pytest
import scipy.sparse as sp

from oracle import reference_restatement as R

pytestmark = pytest.mark.gpu


def _system(nls, dev, which):
    import torch
    if which == "bratu":
        P, PD = R.Bratu2D(24), nls.Bratu2D(24)
    else:
        P, PD = R.Brusselator2D(16), nls.Brusselator2D(16)
    u = P.u0() + 0.1 * np.sin(np.arange(P.n) * 0.37)
    A, b = sp.csr_matrix(P.jac(u)), P.f(u)
    J = PD.jac_csr()
    PD.jac_values(torch.tensor(u, device=dev), J)
    return P, PD, u, A, b, J


@pytest.mark.parametrize("which", ["bratu", "brusselator"])
@pytest.mark.parametrize("ordering", ["natural", "multicolor"])
def test_ilu0_factors_and_apply_match_the_sequential_oracle(nls, dev, which, ordering):
    """The device factorisation runs the sequential IKJ algorithm's operations in its order, row by row inside dependency
    levels: factors equal to the oracle's (rounding: no FMA contraction on either side), the same permutation, the defining
    property (L U = A on the pattern), and M⁻¹ x equal to two SciPy triangular solves."""
    import torch
    P, PD, u, A, b, J = _system(nls, dev, which)
    M = nls.ILU0Preconditioner(J, ordering=ordering)
    Ld, Ud, perm = M.factors()
    perm_o = R.multicolor_permutation(A)[0] if ordering == "multicolor" else np.arange(P.n)
    assert np.array_equal(perm, perm_o)
    Lo, Uo = R.ilu0(A, perm_o if ordering == "multicolor" else None)
    assert abs(Ld - Lo).max() <= 1e-14 * abs(Lo).max() and abs(Ud - Uo).max() <= 1e-14 * abs(Uo).max()
    Ap = A[perm_o][:, perm_o].tocsr()
    pat = Ap.copy(); pat.data[:] = 1.0
    assert abs((Ld @ Ud - Ap).multiply(pat)).max() <= 1e-12 * abs(Ap).max()
    info = M.info()
    if ordering == "multicolor":
        assert info["levels_lower"] == info["ncolors"] == R.multicolor_permutation(A)[1] <= 8
    else:
        assert info["levels_lower"] > 8          # the natural ordering of a stencil is a dependency chain
    Mo = R.ilu0_preconditioner(A, ordering)
    x = np.random.default_rng(3).standard_normal(P.n)
    y = M.apply(torch.tensor(x, device=dev)).cpu().numpy()
    assert np.max(np.abs(y - Mo(x))) <= 1e-12 * np.max(np.abs(Mo(x)))
    assert np.max(np.abs(M.apply(x) - y)) == 0.0            # host vectors in, host vectors out: the same kernels
    # new values on the same pattern: update() refactorises
    u2 = u + 0.05 * np.cos(np.arange(P.n) * 0.11)
    PD.jac_values(torch.tensor(u2, device=dev), J)
    M.update()
    Mo2 = R.ilu0_preconditioner(sp.csr_matrix(P.jac(u2)), ordering)
    assert np.max(np.abs(M.apply(x) - Mo2(x))) <= 1e-12 * np.max(np.abs(Mo2(x)))
    Mj = nls.JacobiPreconditioner(J)
    assert np.max(np.abs(Mj.apply(x) - x / sp.csr_matrix(P.jac(u2)).diagonal())) <= 1e-15 * np.max(np.abs(x))


def test_ilu0_natural_ordering_on_a_long_dependency_chain(nls, dev):
    """Bratu 96²: 191 levels of ≤ 96 rows — the persistent single-workgroup walk (a barrier per level instead of a launch per
    level) — and a tridiagonal matrix (n levels of one row), where ILU(0) is the exact LU: M⁻¹ A x = x."""
    import torch
    P, PD = R.Bratu2D(96), nls.Bratu2D(96)
    u = 0.2 * np.sin(np.arange(P.n) * 0.01)
    A = sp.csr_matrix(P.jac(u))
    J = PD.jac_csr()
    PD.jac_values(torch.tensor(u, device=dev), J)
    M = nls.ILU0Preconditioner(J, ordering="natural")
    assert M.info()["levels_lower"] == 2 * 96 - 1
    x = np.random.default_rng(5).standard_normal(P.n)
    yo = R.ilu0_preconditioner(A, "natural")(x)
    assert np.max(np.abs(M.apply(x) - yo)) <= 1e-12 * np.max(np.abs(yo))
    n = 3000
    T = sp.diags([-1.0 * np.ones(n - 1), 2.5 + np.sin(np.arange(n)), -1.3 * np.ones(n - 1)], [-1, 0, 1]).tocsr()
    Td = nls.CSRMatrix.from_scipy(T)
    Mt = nls.ILU0Preconditioner(Td, ordering="natural")
    assert Mt.info()["levels_lower"] == n
    assert np.max(np.abs(Mt.apply(T @ x[:n]) - x[:n])) <= 1e-11 * np.max(np.abs(x[:n]))
    with pytest.raises(nls.NKError, match="pivot|diagonal"):
        nls.ILU0Preconditioner(nls.CSRMatrix.from_scipy(sp.csr_matrix(np.array([[0.0, 1.0], [1.0, 0.0]]))), ordering="natural")


@pytest.mark.parametrize("ortho", ["sstep", "dcgs2", "cgs2", "mgs"])
@pytest.mark.parametrize("which", ["bratu", "brusselator"])
def test_left_and_two_sided_gmres_match_the_oracle(nls, dev, which, ortho):
    """Pl as an object (ILU(0), Jacobi), Pl as a device callback, Pl + Pr together, Pl + the built-in Chebyshev Pr: iterates,
    iteration counts and the PRECONDITIONED residual norms of the oracle's left-preconditioned GMRES (same (m, rtol))."""
    import torch
    P, PD, u, A, b, J = _system(nls, dev, which)
    bd = torch.tensor(b, device=dev)
    Mo = R.ilu0_preconditioner(A, "natural")
    Mi = nls.ILU0Preconditioner(J, ordering="natural")
    rtol = 1e-9
    xo, io = R.gmres(lambda v: A @ v, b, rtol=rtol, restart=20, itmax=400, ortho="cgs2", Ml=Mo)
    G = nls.GMRES(P.n, restart=20, ortho=ortho).set_operator(J).set_preconditioner(Mi, side="left")
    x, gi = G.solve(bd, abstol=0.0, reltol=rtol, maxiters=400)
    x = x.cpu().numpy()
    assert gi["converged"] and abs(gi["iters"] - io.iters) <= (3 if ortho == "sstep" else 0)   # (the s-step test sees whole blocks)
    assert np.linalg.norm(x - xo) <= 1e-7 * np.linalg.norm(xo)
    assert abs(gi["rnorm0"] - np.linalg.norm(Mo(b))) <= 1e-10 * gi["rnorm0"]                 # the preconditioned ‖r₀‖
    assert np.linalg.norm(Mo(b - A @ x)) <= 1.001 * rtol * np.linalg.norm(Mo(b))
    # the same Pl as a device callback (any linear map on device tensors): identical arithmetic → identical result
    Gc = nls.GMRES(P.n, restart=20, ortho=ortho).set_operator(J).set_left_preconditioner(lambda r: Mi.apply(r))
    xc, gc = Gc.solve(bd, abstol=0.0, reltol=rtol, maxiters=400)
    assert gc["iters"] == gi["iters"] and np.linalg.norm(xc.cpu().numpy() - x) <= 1e-12 * np.linalg.norm(x)
    # both sides: Pl = ILU(0), Pr = Jacobi
    Mj = nls.JacobiPreconditioner(J)
    x2o, i2o = R.gmres(lambda v: A @ v, b, rtol=rtol, restart=20, itmax=400, ortho="cgs2", Ml=Mo, M=R.jacobi_preconditioner(A))
    G2 = nls.GMRES(P.n, restart=20, ortho=ortho).set_operator(J).set_preconditioner(Mi, "left").set_preconditioner(Mj, "right")
    x2, g2 = G2.solve(bd, abstol=0.0, reltol=rtol, maxiters=400)
    assert g2["converged"] and abs(g2["iters"] - i2o.iters) <= (3 if ortho == "sstep" else 0)
    assert np.linalg.norm(x2.cpu().numpy() - x2o) <= 1e-7 * np.linalg.norm(x2o)
    # removing the left side again gives the plain right-preconditioned solve
    G2.set_left_preconditioner(None)
    x3o, i3o = R.gmres(lambda v: A @ v, b, rtol=1e-6, restart=20, itmax=60, ortho="cgs2", M=R.jacobi_preconditioner(A))
    x3, g3 = G2.solve(bd, abstol=0.0, reltol=1e-6, maxiters=60)
    assert abs(g3["iters"] - i3o.iters) <= (3 if ortho == "sstep" else 0)
    assert abs(g3["rnorm0"] - np.linalg.norm(b)) <= 1e-12 * gi["rnorm0"] * 1e6


def test_right_side_objects_and_restarts_with_left_preconditioning(nls, dev):
    """ILU(0) on the right (x = Pr⁻¹ z, true residual norms), and a left-preconditioned solve through several restart cycles
    (every restart residual goes through Pl⁻¹) with the fixed-work protocol."""
    import torch
    P, PD, u, A, b, J = _system(nls, dev, "brusselator")
    bd = torch.tensor(b, device=dev)
    Mo = R.ilu0_preconditioner(A, "multicolor")
    Mi = nls.ILU0Preconditioner(J, ordering="multicolor")
    xo, io = R.gmres(lambda v: A @ v, b, rtol=1e-9, restart=30, itmax=400, ortho="cgs2", M=Mo)
    G = nls.GMRES(P.n, restart=30).set_operator(J).set_preconditioner(Mi, side="right")
    x, gi = G.solve(bd, abstol=0.0, reltol=1e-9, maxiters=400)
    assert gi["converged"] and abs(gi["iters"] - io.iters) <= 3 and np.linalg.norm(x.cpu().numpy() - xo) <= 1e-7 * np.linalg.norm(xo)
    assert np.linalg.norm(b - A @ x.cpu().numpy()) <= 1.001e-9 * np.linalg.norm(b)
    for ortho in ("sstep", "dcgs2"):
        xo2, io2 = R.gmres(lambda v: A @ v, b, restart=5, fixed_iters=20, ortho="cgs2", Ml=Mo)
        G2 = nls.GMRES(P.n, restart=5, ortho=ortho).set_operator(J).set_preconditioner(Mi, side="left")
        x2, g2 = G2.solve(bd, fixed_iters=20)
        assert g2["iters"] == 20 == io2.iters and g2["restarts"] == io2.restarts
        assert np.linalg.norm(x2.cpu().numpy() - xo2) <= 1e-9 * np.linalg.norm(xo2)
        assert abs(g2["rnorm"] - io2.rnorm) <= 1e-7 * io2.rnorm


def test_precs_protocol_call_counts_core_tests_item21(nls, dev):
    """test/Core/core_tests__item21.jl:10-37 on the device solver: the same DummyPreconditioners, the same assertions (see the
    oracle's pin of the same name); the hook runs between the Jacobian refresh and the linear solve of every step."""
    import torch

    class Dummy:
        def __init__(self):
            self.i, self.reinit_check, self.seen = 0, 0, []

        def __call__(self, W, p=None):
            assert isinstance(p, nls.LinearSolveParameters) and p.p == self.reinit_check
            assert torch.is_tensor(p.u) and p.u.is_cuda and p.u.numel() == 2
            self.seen.append(type(W).__name__)
            self.i += 1
            return nls.IDENTITY, nls.IDENTITY

    def F(du, u, p):
        du.copy_(-(u - 0.1) ** 3)

    def JVP(out, v, u, p):
        out.copy_(-3.0 * (u - 0.1) ** 2 * v)

    f = nls.NonlinearFunction(F, jvp=JVP, vjp=JVP)
    prob = nls.NonlinearProblem(f, torch.zeros(2, dtype=torch.float64, device=dev), 0)
    precs = Dummy()
    it = nls.init(prob, nls.NewtonRaphson(linsolve=nls.KrylovJL_GMRES(precs=precs), concrete_jac=False))
    iinit = precs.i
    it.solve()
    assert precs.i > 0 and set(precs.seen) == {"StatefulJacobianOperator"}
    iprev = precs.i
    precs.i, precs.reinit_check = 0, 1
    it.reinit(torch.zeros(2, dtype=torch.float64, device=dev), p=1)
    ireinit = precs.i
    it.solve()
    assert precs.i - ireinit == iprev - iinit
    precs.i, precs.reinit_check = 0, 2
    it.reinit(p=2)
    assert precs.i == 0
    it.solve()
    assert precs.i == 1
    # … and the oracle's restatement counts the same calls
    class Cubic:
        n, p = 2, 0
        def u0(self): return np.zeros(2)
        def f(self, u): return -(u - 0.1) ** 3
        def jvp(self, v, u): return -3.0 * (u - 0.1) ** 2 * v
        def vjp(self, v, u): return -3.0 * (u - 0.1) ** 2 * v
    cnt = [0]
    def op(W, p=None):
        cnt[0] += 1
        return None, None
    oc = R.init(Cubic(), R.NewtonRaphson(linsolve=R.KrylovJL_GMRES(precs=op)))
    oc.solve()
    assert cnt[0] == iprev and iinit == 1
    it.close()


@pytest.mark.parametrize("how", ["callable_object", "callable_callback", "options_natural", "options_multicolor"])
def test_newton_with_device_ilu0_as_left_preconditioner(nls, dev, how):
    """docs/src/tutorials/large_systems.md:252-260 on the device: NewtonRaphson(linsolve = KrylovJL_GMRES(precs = incompletelu),
    concrete_jac = true) on the Brusselator of sparsity_tests__item1.jl (N = 32), `incompletelu(W, p) = (ILU(0) of W, I)` —
    Pl as a device object returned from a Python `precs`, as a Python callback on device tensors, and through nk_options (no
    host callback at all) — against the oracle's Newton solve with the same left preconditioner: steps, Krylov iterations, root."""
    import torch
    PB = nls.Brusselator2D(32)
    calls = []
    state = {}
    if how in ("callable_object", "callable_callback"):
        def incompletelu(W, p=None):
            calls.append(W)
            assert isinstance(W, nls.CSRMatrix) and isinstance(p, nls.LinearSolveParameters)
            if "M" not in state:
                state["M"] = nls.ILU0Preconditioner(W, ordering="natural")
            else:
                state["M"].update()                 # the same pattern, the new Jacobian's values
            M = state["M"]
            return (M if how == "callable_object" else (lambda r: M.apply(r))), None
        precs, okind = incompletelu, "ilu0_natural"
    else:
        okind = "ilu0_natural" if how == "options_natural" else "ilu0"
        precs = nls.ObjectPrecs(okind, "left")
    alg = nls.NewtonRaphson(linsolve=nls.KrylovJL_GMRES(precs=precs, gmres_restart=30, maxiters=3000, reltol=1e-8, abstol=0.0),
                            concrete_jac=True)
    sol = nls.solve(nls.NonlinearProblem(PB, u0=PB.initial_guess(device=True)), alg, abstol=1e-8, maxiters=50)
    oc = R.init(R.Brusselator2D(32), R.NewtonRaphson(linsolve=R.KrylovJL_GMRES(precs=R.ObjectPrecs(okind, "left"), gmres_restart=30,
                                                                                maxiters=3000, ortho="cgs2"), concrete_jac=True),
                abstol=1e-8, maxiters=50)
    oc.lin_reltol, oc.lin_abstol = 1e-8, 0.0
    ref = oc.solve()
    assert sol.retcode == "Success" == R.RETCODE_NAMES[ref.retcode] and sol.stats.nsteps == ref.stats.nsteps
    # natural ordering: ≈ 270 iterations per solve, the count follows the oracle's to a few per cent. The weaker multicolour M
    # needs 550–1100 iterations per solve through 18–37 restarts of GMRES(30) — near stagnation, where rounding decides the
    # count (s-step blocks: 1804, delayed CGS2: 3341, the oracle's CGS2: 1683 — three correct GMRES(30)): no count is pinned there
    if okind == "ilu0_natural":
        assert abs(sol.stats.gmres_iters - ref.stats.gmres_iters) <= 0.1 * ref.stats.gmres_iters + 3 * sol.stats.nsteps
    assert np.max(np.abs(np.asarray(sol.u.cpu()) - ref.u)) <= 1e-7 * np.max(np.abs(ref.u))
    if calls:
        assert len(calls) == sol.stats.nsteps + 1                      # once when the cache is built, once per new Jacobian


def test_csc_value_refresh_matches_a_fresh_ingest(nls, dev):
    """nk_csr_set_values_csc: the next Jacobian's `nonzeros(J)` (CSC order) land on the device with one gather — SpMV equal to
    a matrix ingested from scratch."""
    import torch
    rng = np.random.default_rng(7)
    A1 = sp.random(300, 300, density=0.03, random_state=3, format="csc") + sp.identity(300, format="csc")
    A1.sort_indices()
    A2 = A1.copy(); A2.data = rng.standard_normal(A2.nnz)
    J = nls.CSRMatrix.from_csc(A1.indptr + 1, A1.indices + 1, A1.data)
    x = rng.standard_normal(300)
    xd = torch.tensor(x, device=dev)
    assert np.max(np.abs(J.matvec(xd).cpu().numpy() - A1 @ x)) <= 1e-13 * np.max(np.abs(A1 @ x))
    J.set_values_csc(A2.data)
    assert np.max(np.abs(J.matvec(xd).cpu().numpy() - A2 @ x)) <= 1e-13 * np.max(np.abs(A2 @ x))
    J.set_values_csc(torch.tensor(A1.data, device=dev))                 # a resident nzval: no host copy
    assert np.max(np.abs(J.matvec(xd).cpu().numpy() - A1 @ x)) <= 1e-13 * np.max(np.abs(A1 @ x))
    with pytest.raises(nls.NKError, match="entries"):
        J.set_values_csc(A2.data[:-1])
    with pytest.raises(nls.NKError, match="CSC"):
        nls.CSRMatrix.from_scipy(A1.tocsr()).set_values_csc(A2.data)


# ----------------------------------------------------------------------------- ILU with a drop tolerance (the tutorial's other precs)
@pytest.mark.parametrize("case", ["random", "brusselator"])
def test_ilut_factors_and_apply_match_the_oracle(nls, dev, case):
    """nk_precond_create_ilut: Crout ILU(τ) of the host (as IncompleteLU.jl's runs on the CPU) against the NumPy restatement — the
    same pattern, the factors to 1e-12, the device's level-scheduled triangular solves against SciPy's; τ = 0 is the complete LU;
    `update()` re-plans for new values (the pattern is a function of the numbers)."""
    import torch
    rng = np.random.default_rng(3)
    if case == "random":
        n = 400
        A = (sp.random(n, n, density=0.02, random_state=5, format="csr") + sp.diags(3.0 + rng.random(n))).tocsr()
        tau = 0.05
    else:
        pb = R.Brusselator2D(16)
        A = pb.jac(pb.u0() + 0.05 * rng.standard_normal(pb.n)).tocsr()
        tau = 50.0
    A.sort_indices()
    M = nls.CSRMatrix.from_scipy(A)
    P = nls.ILUTPreconditioner(M, tau)
    Lo, Uo = R.ilut(A, tau)
    Ld, Ud, perm = P.factors()
    assert np.array_equal(perm, np.arange(A.shape[0]))
    for Fd, Fo in ((Ld, Lo), (Ud, Uo)):
        Fd.sort_indices(); Fo.sort_indices()
        assert np.array_equal(Fd.indptr, Fo.indptr) and np.array_equal(Fd.indices, Fo.indices)
        assert np.max(np.abs(Fd.data - Fo.data)) <= 1e-12 * np.max(np.abs(Fo.data))
    assert Lo.nnz + Uo.nnz > A.nnz or case == "random"                  # fill is kept
    x = rng.standard_normal(A.shape[0])
    want = R.ilut_preconditioner(A, tau)(x)
    got = P.apply(torch.tensor(x, device=dev)).cpu().numpy()
    assert np.max(np.abs(got - want)) <= 1e-11 * np.max(np.abs(want))
    assert P.info()["kind"] == "ilut" and P.info()["levels_lower"] >= 1
    # τ = 0: the complete LU (no pivoting) — M⁻¹ A x = x
    P0 = nls.ILUTPreconditioner(M, 0.0)
    L0, U0, _ = P0.factors()
    assert abs(L0 @ U0 - A).max() <= 1e-11 * abs(A).max()
    # new values, same object
    A2 = A.copy(); A2.data = A.data * (1.0 + 0.1 * rng.standard_normal(A.nnz)) ; A2 = (A2 + sp.diags(np.full(A.shape[0], 0.5 * abs(A).max()))).tocsr()
    A2.sort_indices()
    if A2.nnz == A.nnz:
        M.set_values(A2.data)
        P.update()
        want2 = R.ilut_preconditioner(A2, tau)(x)
        got2 = P.apply(torch.tensor(x, device=dev)).cpu().numpy()
        assert np.max(np.abs(got2 - want2)) <= 1e-11 * np.max(np.abs(want2))


def test_newton_with_ilut_as_left_preconditioner_as_in_the_tutorial(nls, dev):
    """docs/src/tutorials/large_systems.md:252-260: `incompletelu(W, p = nothing) = (ilu(W, τ = 50.0), LinearAlgebra.I)`,
    NewtonRaphson(linsolve = KrylovJL_GMRES(precs = incompletelu), concrete_jac = true) on the Brusselator (N = 32) — the device
    object returned from a Python `precs` against the oracle's solve with the restated factors: steps, root, Krylov iterations;
    and the fill pays: far fewer iterations than ILU(0) in the same ordering."""
    PB = nls.Brusselator2D(32)
    state = {}

    def incompletelu(W, p=None):
        if "M" not in state:
            state["M"] = nls.ILUTPreconditioner(W, 50.0)
        else:
            state["M"].update()
        return state["M"], None

    kw = dict(gmres_restart=30, maxiters=3000)
    alg = nls.NewtonRaphson(linsolve=nls.KrylovJL_GMRES(precs=incompletelu, reltol=1e-8, abstol=0.0, **kw), concrete_jac=True)
    sol = nls.solve(nls.NonlinearProblem(PB, u0=PB.initial_guess(device=True)), alg, abstol=1e-8, maxiters=50)
    oc = R.init(R.Brusselator2D(32), R.NewtonRaphson(linsolve=R.KrylovJL_GMRES(precs=lambda W, p: (R.ilut_preconditioner(W, 50.0), None),
                                                                                ortho="cgs2", **kw), concrete_jac=True),
                abstol=1e-8, maxiters=50)
    oc.lin_reltol, oc.lin_abstol = 1e-8, 0.0
    ref = oc.solve()
    assert sol.retcode == "Success" == R.RETCODE_NAMES[ref.retcode] and sol.stats.nsteps == ref.stats.nsteps
    assert abs(sol.stats.gmres_iters - ref.stats.gmres_iters) <= 0.1 * ref.stats.gmres_iters + 3 * sol.stats.nsteps
    assert np.max(np.abs(np.asarray(sol.u.cpu()) - ref.u)) <= 1e-7 * np.max(np.abs(ref.u))
    ilu0 = nls.solve(nls.NonlinearProblem(PB, u0=PB.initial_guess(device=True)),
                     nls.NewtonRaphson(linsolve=nls.KrylovJL_GMRES(precs=nls.ObjectPrecs("ilu0_natural", "left"), reltol=1e-8, abstol=0.0, **kw),
                                       concrete_jac=True), abstol=1e-8, maxiters=50)
    assert sol.stats.gmres_iters < 0.6 * ilu0.stats.gmres_iters, (sol.stats.gmres_iters, ilu0.stats.gmres_iters)
