"""CPU ORACLE — TEST INFRASTRUCTURE ONLY (never imported by the product path).

A NumPy/SciPy restatement of SciML/NonlinearSolve.jl's first-order step path, written to be read next
to the Julia sources (paths relative to the reference tree, v4.27.0):

  step!/init/driver   lib/NonlinearSolveFirstOrder/src/solve.jl:140-301,325-465
                      lib/NonlinearSolveBase/src/solve.jl:360-387,835-859
  NewtonRaphson       lib/NonlinearSolveFirstOrder/src/raphson.jl:30-43
  TrustRegion         lib/NonlinearSolveFirstOrder/src/trust_region.jl:25-43,204-258,320-384,396-514
  EisenstatWalker     lib/NonlinearSolveFirstOrder/src/eisenstat_walker.jl:42-107
  NewtonDescent       lib/NonlinearSolveBase/src/descent/newton.jl:28-56,97-141
  Dogleg / Steepest   lib/NonlinearSolveBase/src/descent/dogleg.jl:86-151, steepest.jl:57-80
  termination         lib/NonlinearSolveBase/src/termination_conditions.jl:243-336,385-453
  defaults / norms    lib/NonlinearSolveBase/src/common_defaults.jl:19-48
  Jacobian operators  lib/SciMLJacobianOperators/src/SciMLJacobianOperators.jl:116-182,210-291

PARITY PINNING.  The reference is Julia and Julia is not installed in the build container, so the
reference itself cannot be executed; it also ships no golden vectors (every test is an outcome /
tolerance test).  This oracle is therefore pinned against (tests/test_oracle_pins.py):
  * every known-answer fixture the reference's own tests hold for this path (quadratic → sqrt(p),
    tridiagonal N=40 `sol.u ≈ Wmat\\bvec`, custom-JVP N=100 `max|resid|<1e-6`, Brusselator N=32
    `‖resid‖∞<1e-8`, JVP/VJP/JᵀJ vs the analytic Jacobian, newton_fails+TrustRegion, TR reinit rules),
  * SciPy as an independent second opinion (scipy.sparse.linalg.gmres / spsolve, scipy.optimize.root).
GMRES arithmetic lives in Krylov.jl (reached through LinearSolve.jl compat "5.4", version unpinned, source
absent from /root/reference): its *iterates* are **parity unpinned**; `gmres()` below restates the published
algorithm (Saad & Schultz restarted GMRES, modified Gram–Schmidt Arnoldi + Givens, stop on
‖r‖ ≤ atol + rtol‖r0‖).  Defaults of SciMLBase's termination-mode struct (patience_steps=100,
patience_objective_multiplier=3, min_max_factor=1.3) are from memory of SciMLBase [EXT].
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Callable, Optional

import numpy as np
import scipy.sparse as sp

# ----------------------------------------------------------------------------- return codes
(DEFAULT, SUCCESS, MAXITERS, UNSTABLE, STALLED, LINSOLVE_FAILED, SHRINK_EXCEEDED, MAXTIME, FAILURE,
 LINESEARCH_FAILED) = range(10)
RETCODE_NAMES = ["Default", "Success", "MaxIters", "Unstable", "Stalled", "InternalLinearSolveFailed",
                 "ShrinkThresholdExceeded", "MaxTime", "Failure", "InternalLineSearchFailed"]


def L2_NORM(x):  # common_defaults.jl:19-30
    x = np.asarray(x, dtype=np.float64).ravel()
    return math.sqrt(float(np.dot(x, x)))


def Linf_NORM(x):  # common_defaults.jl:32-33 / Base.Fix1(maximum, abs)
    x = np.asarray(x, dtype=np.float64)
    return float(np.max(np.abs(x))) if x.size else 0.0


DEFAULT_TOL = 3.0e-13  # common_defaults.jl:44-48 (Float64)


# ----------------------------------------------------------------------------- problems
class Problem:
    """NonlinearProblem(NonlinearFunction(f; jvp, vjp, jac, jac_prototype), u0, p)."""

    n: int

    def f(self, u):
        raise NotImplementedError

    def jac(self, u) -> sp.csr_matrix:
        raise NotImplementedError

    def jvp(self, v, u):  # SciMLJacobianOperators.jl:373-431: f.jvp, else jac*v
        return self.jac(u) @ v

    def vjp(self, v, u):  # SciMLJacobianOperators.jl:296-362
        return self.jac(u).T @ v

    def u0(self):
        raise NotImplementedError


class Quadratic(Problem):
    """quadratic_f(u, p) = u .* u .- p   (common/common_rootfind_testing.jl:15-17)."""

    def __init__(self, n, p=2.0):
        self.n, self.p = int(n), p

    def f(self, u):
        return u * u - self.p

    def jac(self, u):
        return sp.diags(2.0 * u, format="csr")

    def jvp(self, v, u):
        return 2.0 * u * v

    def vjp(self, v, u):
        return 2.0 * u * v

    def u0(self):
        return np.ones(self.n)


class Bratu2D(Problem):
    """2-D Bratu, 5-point stencil, homogeneous Dirichlet, lexicographic k = j*n + i (SURVEY.md §8d;
    not present in the reference).  F_k = s*[(4u_k − u_W − u_E − u_S − u_N)/h² − λ exp(u_k)],
    s = `scale` (None → h², the h²-scaled residual the convergence test uses)."""

    def __init__(self, n, lam=6.0, scale=None):
        self.ns = int(n)
        self.n = self.ns * self.ns
        self.lam = float(lam)
        self.h = 1.0 / (self.ns + 1)
        s = self.h * self.h if scale is None or scale == 0 else float(scale)
        self.c_lap = s / (self.h * self.h)
        self.c_exp = s * self.lam
        self._L = None

    def laplacian(self):
        if self._L is None:
            n = self.ns
            T = sp.diags([-np.ones(n - 1), 4.0 * np.ones(n), -np.ones(n - 1)], [-1, 0, 1])
            I = sp.identity(n)
            S = sp.diags([-np.ones(n - 1), -np.ones(n - 1)], [-1, 1])
            L = sp.kron(I, T) + sp.kron(S, I)
            L = sp.csr_matrix(L)
            L.sort_indices()
            self._L = L
        return self._L

    def f(self, u):
        return self.c_lap * (self.laplacian() @ u) - self.c_exp * np.exp(u)

    def jac(self, u):
        J = (self.c_lap * self.laplacian() - sp.diags(self.c_exp * np.exp(u))).tocsr()
        J.sort_indices()
        return J

    def jvp(self, v, u):
        return self.c_lap * (self.laplacian() @ v) - self.c_exp * np.exp(u) * v

    def vjp(self, v, u):  # symmetric
        return self.jvp(v, u)

    def u0(self):
        return np.zeros(self.n)


class Brusselator2D(Problem):
    """brusselator_2d_loop (lib/NonlinearSolveFirstOrder/test/sparsity_tests__item1.jl:13-36),
    unknowns column-major (i, j, k): idx = i + N*j + N*N*k (0-based), periodic `limit`."""

    def __init__(self, N, A=3.4, B=1.0, alpha=10.0, dx=None):
        self.N = int(N)
        self.n = 2 * self.N * self.N
        self.A, self.B = float(A), float(B)
        self.dx = float(dx) if dx is not None else 1.0 / (self.N - 1)  # step(range(0,1,length=N))
        self.alpha = float(alpha) / (self.dx * self.dx)
        xy = np.linspace(0.0, 1.0, self.N)
        X, Y = np.meshgrid(xy, xy, indexing="ij")  # X[i,j] = x_i, Y[i,j] = y_j
        self.forcing = (((X - 0.3) ** 2 + (Y - 0.6) ** 2) <= 0.1 ** 2) * 5.0
        self._xy = xy
        self._Lp = None

    def _lap(self, w):  # periodic 5-point sum minus 4w on an (N,N) array indexed [i,j]
        return (np.roll(w, 1, 0) + np.roll(w, -1, 0) + np.roll(w, 1, 1) + np.roll(w, -1, 1) - 4.0 * w)

    def _split(self, u):
        N = self.N
        U = u.reshape(2, N, N)  # U[k, j, i]
        return U[0].T, U[1].T  # [i, j]

    def _join(self, a, b):
        return np.concatenate([a.T.ravel(), b.T.ravel()])

    def f(self, u):
        uu, vv = self._split(u)
        du = self.alpha * self._lap(uu) + self.B + uu * uu * vv - (self.A + 1.0) * uu + self.forcing
        dv = self.alpha * self._lap(vv) + self.A * uu - uu * uu * vv
        return self._join(du, dv)

    def lap_matrix(self):
        if self._Lp is None:
            N = self.N
            e = np.ones(N)
            C = sp.diags([e[:-1], e[:-1]], [-1, 1], shape=(N, N)).tolil()
            C[0, N - 1] = 1.0
            C[N - 1, 0] = 1.0
            C = sp.csr_matrix(C)
            I = sp.identity(N)
            self._Lp = sp.csr_matrix(sp.kron(I, C) + sp.kron(C, I) - 4.0 * sp.identity(N * N))
        return self._Lp

    def jac(self, u):
        uu, vv = self._split(u)
        uf, vf = uu.T.ravel(), vv.T.ravel()
        Lp = self.alpha * self.lap_matrix()
        J11 = Lp + sp.diags(2.0 * uf * vf - (self.A + 1.0))
        J12 = sp.diags(uf * uf)
        J21 = sp.diags(self.A - 2.0 * uf * vf)
        J22 = Lp - sp.diags(uf * uf)
        J = sp.bmat([[J11, J12], [J21, J22]], format="csr")
        J.sort_indices()
        return J

    def jvp(self, v, u):
        uu, vv = self._split(u)
        a, b = self._split(v)
        ja = self.alpha * self._lap(a) + (2.0 * uu * vv - (self.A + 1.0)) * a + uu * uu * b
        jb = self.alpha * self._lap(b) + (self.A - 2.0 * uu * vv) * a - uu * uu * b
        return self._join(ja, jb)

    def vjp(self, v, u):
        uu, vv = self._split(u)
        a, b = self._split(v)
        ja = self.alpha * self._lap(a) + (2.0 * uu * vv - (self.A + 1.0)) * a + (self.A - 2.0 * uu * vv) * b
        jb = self.alpha * self._lap(b) + uu * uu * a - uu * uu * b
        return self._join(ja, jb)

    def u0(self):  # init_brusselator_2d, sparsity_tests__item1.jl:40-50
        x = self._xy
        X, Y = np.meshgrid(x, x, indexing="ij")
        return self._join(22.0 * (Y * (1.0 - Y)) ** 1.5, 27.0 * (X * (1.0 - X)) ** 1.5)


class FunctionProblem(Problem):
    """Generic problem from Python callables (mirrors NonlinearFunction(f; jvp, vjp, jac))."""

    def __init__(self, f, u0, jac=None, jvp=None, vjp=None):
        self._f, self._u0 = f, np.asarray(u0, dtype=np.float64)
        self._jac, self._jvp, self._vjp = jac, jvp, vjp
        self.n = self._u0.size

    def f(self, u):
        return np.asarray(self._f(u), dtype=np.float64)

    def jac(self, u):
        if self._jac is None:
            # forward differences, only for tiny fixtures (AutoFiniteDiff stand-in)
            f0 = self.f(u)
            J = np.zeros((f0.size, u.size))
            for j in range(u.size):
                e = np.zeros_like(u)
                hh = math.sqrt(np.finfo(float).eps) * max(1.0, abs(u[j]))
                e[j] = hh
                J[:, j] = (self.f(u + e) - f0) / hh
            return sp.csr_matrix(J)
        return sp.csr_matrix(self._jac(u))

    def jvp(self, v, u):
        return self._jvp(v, u) if self._jvp is not None else self.jac(u) @ v

    def vjp(self, v, u):
        return self._vjp(v, u) if self._vjp is not None else self.jac(u).T @ v

    def u0(self):
        return self._u0.copy()


# ----------------------------------------------------------------------------- GMRES (Krylov.jl restated)
@dataclass
class GmresInfo:
    iters: int = 0
    restarts: int = 0
    converged: bool = False
    failed: bool = False
    rnorm0: float = 0.0
    rnorm: float = 0.0
    residuals: list = field(default_factory=list)


def chebyshev_preconditioner(matvec: Callable, lmin: float, lmax: float, degree: int) -> Callable:
    """M⁻¹ v = p_d(A) v: `degree` steps of the Chebyshev iteration for A y = v from y = 0 on [lmin, lmax]
    (Saad, Iterative Methods for Sparse Linear Systems, Alg. 12.1) — the polynomial preconditioner a user would
    return from the reference's `precs(A, p)` hook (test/Core/core_tests__item21.jl:10-18)."""
    theta, delta = 0.5 * (lmax + lmin), 0.5 * (lmax - lmin)
    sigma1 = theta / delta

    def M(v):
        rho = 1.0 / sigma1
        r = np.array(v, dtype=np.float64, copy=True)
        d = r / theta
        y = d.copy()
        for _ in range(1, degree):
            r = r - matvec(d)
            rho_new = 1.0 / (2.0 * sigma1 - rho)
            d = rho_new * rho * d + (2.0 * rho_new / delta) * r
            y = y + d
            rho = rho_new
        return y
    return M


def gershgorin_lambda(J) -> float:
    """max_i Σ_j |a_ij| with the sign of the trace: the bound the device library uses for a concrete J."""
    J = sp.csr_matrix(J)
    mx = float(abs(J).sum(axis=1).max())
    return -mx if J.diagonal().sum() < 0 else mx


def gmres(matvec: Callable, b, x0=None, atol=0.0, rtol=1e-8, restart=30, itmax=300, fixed_iters=0,
          ortho="mgs", allreduce: Optional[Callable] = None, M: Optional[Callable] = None, Ml: Optional[Callable] = None):
    """Restarted GMRES(m) with MGS Arnoldi and Givens rotations, right-hand side b, zero (or given)
    initial guess.  Stop when the recurrence residual ≤ atol + rtol*‖r0‖ or after itmax Arnoldi steps.
    fixed_iters>0: run exactly that many Arnoldi steps (tolerances ignored).
    `allreduce(x)` (optional) sums partial inner products across ranks (distributed oracle, tests only)."""
    if Ml is not None:
        # left preconditioning — the Pl of `precs(A, p) -> (Pl, Pr)` (lib/NonlinearSolveBase/src/linear_solve.jl:195-199;
        # docs/src/tutorials/large_systems.md:257,284-287 return `(Pl, I)`): GMRES on Pl⁻¹ A (Pr⁻¹) with the right-hand side
        # Pl⁻¹ b; the Arnoldi residual, the stopping test ‖Pl⁻¹(b − A x)‖ ≤ atol + rtol·‖Pl⁻¹ r₀‖ and the reported norms are the
        # PRECONDITIONED ones, as in Krylov.jl's gmres [EXT]. Composes with a right preconditioner M.
        op = (lambda v: Ml(matvec(M(v)))) if M is not None else (lambda v: Ml(matvec(v)))
        if x0 is not None and M is not None:
            raise NotImplementedError("x0 with a right preconditioner")
        z, info = gmres(op, Ml(np.asarray(b, dtype=np.float64)), x0, atol, rtol, restart, itmax, fixed_iters, ortho, allreduce)
        return (M(z) if M is not None else z), info
    if M is not None:  # right preconditioning: solve (A M⁻¹) z = b, x = M⁻¹ z  (zero initial guess only)
        assert x0 is None
        z, info = gmres(lambda v: matvec(M(v)), b, None, atol, rtol, restart, itmax, fixed_iters, ortho, allreduce)
        return M(z), info
    if isinstance(ortho, tuple) and ortho[0] == "sstep":   # ("sstep", s) monomial | ("sstep", s, "newton", (lo, hi))
        interval = ortho[3] if len(ortho) > 3 and ortho[2] == "newton" else None
        s_blk = int(ortho[1]) if int(ortho[1]) > 0 else (15 if interval is not None else 6)
        return gmres_sstep(matvec, b, atol, rtol, restart, itmax, fixed_iters, s_blk, allreduce, x0, interval=interval)
    if ortho == "sstep":
        return gmres_sstep(matvec, b, atol, rtol, restart, itmax, fixed_iters, 6, allreduce, x0)
    ar = allreduce if allreduce is not None else (lambda z: z)
    b = np.asarray(b, dtype=np.float64)
    n = b.size
    x = np.zeros(n) if x0 is None else np.array(x0, dtype=np.float64)
    info = GmresInfo()
    r = b - matvec(x) if x0 is not None and np.any(x) else b.copy()
    beta = math.sqrt(float(ar(np.dot(r, r))))
    info.rnorm0 = info.rnorm = beta
    info.residuals.append(beta)
    if not math.isfinite(beta):
        info.failed = True
        return x, info
    if fixed_iters > 0:
        eps_stop, cap = -1.0, int(fixed_iters)
    else:
        eps_stop, cap = atol + rtol * beta, int(itmax)
    if beta == 0.0 or (fixed_iters <= 0 and beta <= eps_stop):
        info.converged = True
        return x, info
    m = int(restart)
    while True:
        V = np.zeros((m + 1, n))
        R = np.zeros((m, m))          # upper-triangular factor of the Hessenberg matrix
        cs, sn, g = np.zeros(m), np.zeros(m), np.zeros(m + 1)
        V[0] = r / beta
        g[0] = beta
        k = 0
        done = False
        Hraw = np.zeros((m + 1, m))   # un-rotated Hessenberg (dcgs2 only)
        pend_r, pend_beta = None, 1.0
        while k < m and info.iters < cap:
            w = matvec(V[k])
            h = np.zeros(k + 2)
            if ortho == "dcgs2":
                # CGS2 whose second correction is applied one step late (the device's NK_ORTHO_DCGS2): V[k] holds the
                # once-projected p when k ≥ 1 and `w` above is A p. Finish v_k = (p − V r)/β, rebuild A v_k from A p
                # through the Arnoldi relation A V = V H̄, then do the first projection, the second projection's
                # coefficients (its application stays pending) and ‖w″‖ by Pythagoras.
                if k > 0:
                    c = Hraw[: k + 1, :k] @ pend_r
                    V[k] = (V[k] - V[:k].T @ pend_r) / pend_beta
                    w = (w - V[: k + 1].T @ c) / pend_beta
                hh = ar(V[: k + 1] @ w)
                w = w - V[: k + 1].T @ hh
                h2 = ar(V[: k + 1] @ w)
                ss = float(ar(np.dot(w, w)))
                h[: k + 1] = hh + h2
                Hraw[: k + 1, k] = h[: k + 1]
                pend_r, pend_beta = h2, math.sqrt(max(ss - float(h2 @ h2), 0.0))
                Hraw[k + 1, k] = pend_beta
            elif ortho == "mgs":
                for i in range(k + 1):
                    h[i] = float(ar(np.dot(V[i], w)))
                    w = w - h[i] * V[i]
            else:  # classical Gram–Schmidt with one re-orthogonalisation (CGS2)
                hh = ar(V[: k + 1] @ w)
                w = w - V[: k + 1].T @ hh
                h2 = ar(V[: k + 1] @ w)
                w = w - V[: k + 1].T @ h2
                h[: k + 1] = hh + h2
            hn = pend_beta if ortho == "dcgs2" else math.sqrt(float(ar(np.dot(w, w))))
            h[k + 1] = hn
            for i in range(k):  # apply previous rotations
                t = cs[i] * h[i] + sn[i] * h[i + 1]
                h[i + 1] = -sn[i] * h[i] + cs[i] * h[i + 1]
                h[i] = t
            d = math.hypot(h[k], h[k + 1])
            if d == 0.0:
                cs[k], sn[k] = 1.0, 0.0
            else:
                cs[k], sn[k] = h[k] / d, h[k + 1] / d
            R[: k, k] = h[:k]
            R[k, k] = d
            g[k + 1] = -sn[k] * g[k]
            g[k] = cs[k] * g[k]
            info.iters += 1
            k += 1
            info.rnorm = abs(g[k])
            info.residuals.append(info.rnorm)
            if not math.isfinite(info.rnorm):
                info.failed = True
                done = True
                break
            if fixed_iters <= 0 and info.rnorm <= eps_stop:
                info.converged = True
                done = True
                break
            if hn == 0.0:  # happy breakdown
                info.converged = True
                done = True
                break
            V[k] = w if ortho == "dcgs2" else w / hn   # dcgs2: un-normalised, un-corrected until the next step
        if k > 0 and not info.failed:
            y = np.linalg.solve(np.triu(R[:k, :k]), g[:k]) if k > 1 else np.array([g[0] / R[0, 0]])
            x = x + V[:k].T @ y
        if done or info.iters >= cap:
            return x, info
        info.restarts += 1
        r = b - matvec(x)
        beta = math.sqrt(float(ar(np.dot(r, r))))


def gmres_dcgs2_1r(matvec: Callable, b, atol=0.0, rtol=1e-8, restart=30, itmax=300, fixed_iters=0,
                   allreduce: Optional[Callable] = None):
    """Restarted GMRES(m), zero initial guess, whose Arnoldi process is CGS2 with delayed re-orthogonalisation and ONE
    reduction per step (the device's NK_ORTHO_DCGS2_1R; Bielich, Langou, Thomas, Świrydowicz et al., "Low-synch
    Gram–Schmidt with delayed reorthogonalization for Krylov solvers"). Column k holds the once-projected, un-normalised
    u_k; the operator is applied to it; one fused reduction yields r = Vᵀu (its pending second projection), ‖u‖², and
    the raw projections Vᵀz, uᵀz of z = A u; the Hessenberg column k−1 (one step late), v_k = (u − V r)/β,
    A v_k = (z − V H̄ r)/β and its first projection follow algebraically; one axpy sweep stores v_k and u_{k+1}.
    Same arithmetic as CGS2 up to rounding; the stopping test sees a column one step after CGS2 would. The pending column
    left by a cycle's last step only closes the last Hessenberg column, so it is not re-orthogonalised (its norm comes out
    of the last axpy sweep): the solutions differ from the fully re-orthogonalised form by ~1e-15 relative."""
    ar = allreduce if allreduce is not None else (lambda z: z)
    b = np.asarray(b, dtype=np.float64)
    n = b.size
    x = np.zeros(n)
    info = GmresInfo()
    r0 = b.copy()
    beta0 = math.sqrt(float(ar(np.dot(r0, r0))))
    info.rnorm0 = info.rnorm = beta0
    info.residuals.append(beta0)
    if not math.isfinite(beta0):
        info.failed = True
        return x, info
    eps_stop, cap = (-1.0, int(fixed_iters)) if fixed_iters > 0 else (atol + rtol * beta0, int(itmax))
    if beta0 == 0.0 or (fixed_iters <= 0 and beta0 <= eps_stop):
        info.converged = True
        return x, info
    m = int(restart)
    while True:
        V = np.zeros((m + 2, n))
        Hraw = np.zeros((m + 2, m + 1))
        R = np.zeros((m, m))
        cs, sn, g = np.zeros(m), np.zeros(m), np.zeros(m + 1)
        V[0] = r0 / beta0
        g[0] = beta0
        steps = min(m, cap - info.iters)
        kdone, done = 0, False

        def finish_column(j, h):  # Givens on Hessenberg column j (entries 0..j+1), as in `gmres`
            nonlocal kdone, done
            for i in range(j):
                tt = cs[i] * h[i] + sn[i] * h[i + 1]
                h[i + 1] = -sn[i] * h[i] + cs[i] * h[i + 1]
                h[i] = tt
            dd = math.hypot(h[j], h[j + 1])
            cs[j], sn[j] = (1.0, 0.0) if dd == 0.0 else (h[j] / dd, h[j + 1] / dd)
            R[:j, j] = h[:j]
            R[j, j] = dd
            g[j + 1] = -sn[j] * g[j]
            g[j] = cs[j] * g[j]
            info.iters += 1
            kdone = j + 1
            info.rnorm = abs(g[j + 1])
            info.residuals.append(info.rnorm)
            if not math.isfinite(info.rnorm):
                info.failed = True
                done = True
            elif (fixed_iters <= 0 and info.rnorm <= eps_stop) or h[j + 1] == 0.0:
                info.converged = True
                done = True

        tprev = None
        for k in range(steps + 1):
            last = (k == steps)                      # the flush: no operator, only the pending column's reduction
            u = V[k]
            z = None if last else matvec(u)
            if k == 0:
                tl = float(ar(np.dot(V[0], z)))      # v_0 is final: only the first projection of A v_0
                V[1] = z - tl * V[0]
                tprev = np.array([tl])
                continue
            if last:  # the column that closes the cycle is never a basis vector: first projection only, β = ‖u‖
                red = np.concatenate([np.zeros(k), ar(np.array([np.dot(u, u)]))])
            else:
                red = ar(np.concatenate([V[:k] @ u, [np.dot(u, u)], V[:k] @ z, [np.dot(u, z)]]))
            rr, a = red[:k], float(red[k])
            beta = math.sqrt(max(a - float(rr @ rr), 0.0))
            h = np.zeros(k + 1)
            h[:k] = tprev + rr
            h[k] = beta
            Hraw[: k + 1, k - 1] = h
            finish_column(k - 1, h.copy())
            if done or last:
                if not done and beta > 0:
                    V[k] = (u - V[:k].T @ rr) / beta
                break
            gg, d = red[k + 1: 2 * k + 1], float(red[2 * k + 1])
            c = Hraw[: k + 1, :k] @ rr
            ttop = (gg - c[:k]) / beta
            tlast = (d - float(rr @ gg) - beta * c[k]) / beta ** 2
            vk = (u - V[:k].T @ rr) / beta
            w = (z - V[:k].T @ c[:k] - vk * c[k]) / beta
            V[k] = vk
            V[k + 1] = w - V[:k].T @ ttop - vk * tlast
            tprev = np.concatenate([ttop, [tlast]])
        if kdone > 0 and not info.failed:
            y = np.linalg.solve(np.triu(R[:kdone, :kdone]), g[:kdone]) if kdone > 1 else np.array([g[0] / R[0, 0]])
            x = x + V[:kdone].T @ y
        if done or info.iters >= cap:
            return x, info
        info.restarts += 1
        r0 = b - matvec(x)
        beta0 = math.sqrt(float(ar(np.dot(r0, r0))))


class SStepBreakdown(Exception):
    """A block of the monomial basis lost rank numerically (the Cholesky factorisation of the Pythagorean Gram block failed)."""


def leja_chebyshev_nodes(s):
    """Chebyshev points of [−1, 1] in Leja order (csrc/nk_sstep.hip::nk_ss_leja_nodes, the same loop): t_0 the point of
    largest modulus, t_j the point maximising Π_{i<j} |t − t_i|; ties go to the first point of cos((2i+1)π/2s), i = 0..s−1."""
    pts = [math.cos((2.0 * i + 1.0) * math.pi / (2.0 * s)) for i in range(s)]
    used, out = [False] * s, []
    for j in range(s):
        best, bv = -1, -1.0
        for i in range(s):
            if used[i]:
                continue
            v = abs(pts[i]) if j == 0 else 1.0
            for q in range(j):
                v *= abs(pts[i] - out[q])
            if v > bv:
                bv, best = v, i
        used[best] = True
        out.append(pts[best])
    return np.array(out)


def gershgorin_interval(A):
    """[min_i (a_ii − r_i), max_i (a_ii + r_i)], r_i = Σ_{j≠i} |a_ij|: bounds of the real part of A's spectrum — where the
    device places the shifts of the s-step Newton basis for a concrete CSR operator (csrc/nk_csr.hip::k_csr_gershgorin)."""
    A = sp.csr_matrix(A)
    d = A.diagonal()
    rad = np.asarray(abs(A).sum(axis=1)).ravel() - np.abs(d)
    return float(np.min(d - rad)), float(np.max(d + rad))


# ----------------------------------------------------------------------------- preconditioner objects (csrc/nk_precond.hip)
def multicolor_permutation(A):
    """Greedy distance-1 colouring of the symmetrised pattern in natural order (smallest free colour), rows then ordered by
    (colour, original index): perm[permuted row] = original row, and the number of colours — csrc/nk_precond.hip::multicolor_perm."""
    A = sp.csr_matrix(A)
    S = (A + A.T).tocsr()
    n = A.shape[0]
    color = -np.ones(n, dtype=np.int64)
    nc = 0
    for i in range(n):
        nb = S.indices[S.indptr[i]:S.indptr[i + 1]]
        used = set(int(color[j]) for j in nb if j != i and color[j] >= 0)
        c = 0
        while c in used:
            c += 1
        color[i] = c
        nc = max(nc, c + 1)
    return np.argsort(color, kind="stable").astype(np.int64), nc


def ilu0(A, perm=None):
    """ILU(0): A[perm][:, perm] ≈ L U on the pattern of A — no fill, no pivoting, L unit lower — by the sequential IKJ
    algorithm (Saad, Iterative Methods, Alg. 10.4), each row's updates in ascending column order: the operations of
    csrc/nk_precond.hip::ilu_factor_row in the same order. Returns (L, U) as CSR. The defining property — (L U)_ij = A_ij on
    the pattern — pins it independently of any implementation."""
    A = sp.csr_matrix(A, dtype=np.float64)
    if perm is not None:
        A = A[perm][:, perm].tocsr()
    A.sort_indices()
    n = A.shape[0]
    rp, ci, lu = A.indptr, A.indices, A.data.copy()
    dg = np.empty(n, dtype=np.int64)
    for i in range(n):
        row = ci[rp[i]:rp[i + 1]]
        d = np.searchsorted(row, i)
        if d >= row.size or row[d] != i:
            raise ArithmeticError(f"ILU(0): row {i} has no stored diagonal entry")
        dg[i] = rp[i] + d
    for i in range(n):
        for p in range(rp[i], dg[i]):
            k = ci[p]
            piv = lu[dg[k]]
            l = lu[p] / piv
            lu[p] = l
            q, s_ = p + 1, dg[k] + 1
            qe, se = rp[i + 1], rp[k + 1]
            while q < qe and s_ < se:
                if ci[q] == ci[s_]:
                    lu[q] -= l * lu[s_]
                    q += 1
                    s_ += 1
                elif ci[q] < ci[s_]:
                    q += 1
                else:
                    s_ += 1
        if lu[dg[i]] == 0.0 or not math.isfinite(lu[dg[i]]):
            raise ArithmeticError("ILU(0): zero or non-finite pivot")
    Mx = sp.csr_matrix((lu, ci.copy(), rp.copy()), shape=(n, n))
    return (sp.tril(Mx, -1) + sp.identity(n)).tocsr(), sp.triu(Mx, 0).tocsr()


def ilu0_preconditioner(A, ordering="multicolor"):
    """x ↦ M⁻¹ x with M = Pᵀ L U P, the ILU(0) of A in the chosen ordering ("natural" | "multicolor") — what
    nls.ILU0Preconditioner applies on the device (two sparse triangular solves in the permuted numbering)."""
    import scipy.sparse.linalg as spla
    perm = multicolor_permutation(A)[0] if ordering == "multicolor" else None
    Lf, Uf = ilu0(A, perm)
    Lc, Uc = sp.csr_matrix(Lf), sp.csr_matrix(Uf)

    def apply(x):
        x = np.asarray(x, dtype=np.float64)
        xb = x if perm is None else x[perm]
        z = spla.spsolve_triangular(Uc, spla.spsolve_triangular(Lc, xb, lower=True, unit_diagonal=True), lower=False)
        if perm is None:
            return z
        out = np.empty_like(z)
        out[perm] = z
        return out
    return apply


def ilut(A, tau):
    """Crout ILU with a drop tolerance — the tutorial's `ilu(W, τ = 50.0)` (docs/src/tutorials/large_systems.md:252-260;
    IncompleteLU.jl [EXT — not in /root/reference; restated from Li, Saad, Chow, "Crout versions of ILU for general sparse
    matrices", SIAM J. Sci. Comput. 25 (2003), with IncompleteLU.jl's absolute drop rule as far as it can be told without its
    source: parity unpinned]). A ≈ (I + L) U; step k forms
        z = A[k, k:] − Σ_{i<k} l_ki U[i, k:]        u_kk = z_k,  u_kj = z_j kept if |z_j| ≥ τ (j > k)
        w = A[k+1:, k] − Σ_{i<k} u_ik L[k+1:, i]     l_ik = w_i / u_kk kept if |w_i| ≥ τ (i > k)
    — the test compares the entry BEFORE the division by the pivot. Sums run over ascending i (csrc/nk_precond.hip::ilut_update
    does the same). Returns (L, U) as CSR, L with its unit diagonal. Defining properties that pin it independently of any
    implementation: τ = 0 gives the complete LU (L U = A), and every kept entry equals the exact Crout recurrence on the kept
    pattern."""
    A = sp.csr_matrix(A, dtype=np.float64)
    A.sum_duplicates()
    A.sort_indices()
    n = A.shape[0]
    Ac = A.tocsc()
    Ac.sort_indices()
    Urow = [dict() for _ in range(n)]     # strictly upper entries of U's rows
    Lcol = [dict() for _ in range(n)]     # strictly lower entries of L's columns
    Lrow = [[] for _ in range(n)]         # (i, l_ki) of row k, ascending i
    Ucol = [[] for _ in range(n)]         # (i, u_ik) of column k, ascending i
    diag = np.zeros(n)
    for k in range(n):
        z = {}
        for j, v in zip(A.indices[A.indptr[k]:A.indptr[k + 1]], A.data[A.indptr[k]:A.indptr[k + 1]]):
            if j >= k:
                z[j] = z.get(j, 0.0) + v
        if k not in z:
            raise ArithmeticError(f"ILU(τ): row {k} has no stored diagonal entry")
        for i, lki in Lrow[k]:
            for j, uij in Urow[i].items():
                if j >= k:
                    z[j] = z.get(j, 0.0) - lki * uij
        piv = z[k]
        if piv == 0.0 or not math.isfinite(piv):
            raise ArithmeticError("ILU(τ): zero or non-finite pivot")
        diag[k] = piv
        for j in sorted(z):
            if j > k and abs(z[j]) >= tau and z[j] != 0.0:
                Urow[k][j] = z[j]
                Ucol[j].append((k, z[j]))
        w = {}
        for r, v in zip(Ac.indices[Ac.indptr[k]:Ac.indptr[k + 1]], Ac.data[Ac.indptr[k]:Ac.indptr[k + 1]]):
            if r > k:
                w[r] = w.get(r, 0.0) + v
        for i, uik in Ucol[k]:
            for r, lri in Lcol[i].items():
                if r > k:
                    w[r] = w.get(r, 0.0) - uik * lri
        for r in sorted(w):
            if abs(w[r]) >= tau and w[r] != 0.0:
                l = w[r] / piv
                Lcol[k][r] = l
                Lrow[r].append((k, l))
    rows, cols, vals = [], [], []
    for k in range(n):
        for r, l in Lcol[k].items():
            rows.append(r); cols.append(k); vals.append(l)
    Lm = sp.csr_matrix((vals, (rows, cols)), shape=(n, n)) + sp.identity(n, format="csr")
    rows, cols, vals = list(range(n)), list(range(n)), list(diag)
    for k in range(n):
        for j, u in Urow[k].items():
            rows.append(k); cols.append(j); vals.append(u)
    Um = sp.csr_matrix((vals, (rows, cols)), shape=(n, n))
    return Lm.tocsr(), Um.tocsr()


def ilut_preconditioner(A, tau):
    """x ↦ U⁻¹ (I + L)⁻¹ x with the Crout ILU(τ) factors of A — what nls.ILUTPreconditioner applies on the device."""
    import scipy.sparse.linalg as spla
    Lf, Uf = ilut(A, tau)
    Lc, Uc = sp.csr_matrix(Lf), sp.csr_matrix(Uf)
    return lambda x: spla.spsolve_triangular(Uc, spla.spsolve_triangular(Lc, np.asarray(x, dtype=np.float64), lower=True,
                                                                          unit_diagonal=True), lower=False)


def jacobi_preconditioner(A):
    d = sp.csr_matrix(A).diagonal()
    return lambda x: np.asarray(x, dtype=np.float64) / d


# ----------------------------------------------------------------------------- aggregation algebraic multigrid
# The slot the reference's tutorial fills with AlgebraicMultigrid.jl's ruge_stuben / smoothed_aggregation returned from
# `precs(A, p)` as Pl (docs/src/tutorials/large_systems.md:276-316) [EXT package; its arithmetic is not in the tree]: a
# multigrid preconditioner built from the sparse Jacobian ALONE. What the device builds (csrc/nk_amg.hip), restated:
#   * coarsening by PAIRWISE AGGREGATION, `passes` times per level (aggregates of ≤ 2^passes rows): rows in natural order; an
#     unmatched row i takes the unmatched neighbour j with the largest strength s_ij = −a_ij·sign(a_ii) (couplings of the sign
#     opposite to the diagonal; first in CSR order on ties) provided s_ij ≥ θ·max_k s_ik, else stays a singleton; the next
#     pass works on the Galerkin matrix of the pairs;
#   * piecewise-constant prolongation T (one 1 per row), Galerkin coarse matrices TᵀA T — every coarse entry the sum of the
#     fine entries it covers, in ascending order of their position in the fine value array;
#   * levels until ≤ coarse_max rows (or until a level shrinks by less than 20 %); the coarsest matrix is inverted (dense) —
#     if coarsening stalled above coarse_max rows it is smoothed 4ν times instead;
#   * V-cycle: ν Chebyshev steps on D⁻¹A over [λmax/ratio, λmax], λmax = max_i Σ_j |a_ij| / |a_ii| (Gershgorin), before and
#     after; the coarse correction is OVER-CORRECTED by ω (plain aggregation under-estimates smooth error by about a factor of
#     two; 1.8 is the measured optimum on the 5-point Laplacian and the Brusselator) — x += ω·T x_c.
# A fixed linear operator (fixed degree, fixed hierarchy), so it serves plain GMRES on either side. New values on the same
# pattern (a new Jacobian) refresh the numbers — Galerkin sums, D⁻¹, λmax, the coarse inverse — and keep the aggregates.
#
# Two matchings: "greedy" (the sequential pass above — the device's HOST set-up: several ranks, NK_AMG_SETUP=host) and
# "handshake" (round 5, the device-side set-up, the default on one rank): every unmatched row names its best unmatched
# neighbour — largest s_ij among s_ij ≥ θ·max_k s_ik, ties by a key that both ends of an edge compute alike — and two rows
# that name each other are paired; AMG_HS_ROUNDS rounds, what is left stays a singleton. The tie-break key of the edge
# a < b, d = b − a: [d (variant 1: far first) or 2³¹ − 1 − d (variant 0: near first)] ≫ [⌊a / d⌋ even first] ≫ [a 32-bit hash
# of (a, b)]: on a lexicographically numbered grid with equal couplings ⌊a / d⌋ is the position along the grid line, so whole
# lines pair up in ONE round and the aggregates come out as regular as the greedy pass's. A level that variant 0 does not
# coarsen by 2^passes within 2 % is coarsened with both variants and keeps the one with fewer aggregates (ties: variant 0) —
# near-first reproduces the greedy aggregates on even-sized and periodic grids, far-first keeps the pairs aligned on odd-sized ones.
AMG_HS_ROUNDS = 8


def amg_edge_hash(a, b):
    a = np.asarray(a).astype(np.uint32)
    b = np.asarray(b).astype(np.uint32)
    h = (a * np.uint32(0x9E3779B1)) ^ (b * np.uint32(0x85EBCA77))
    h = h ^ (h >> np.uint32(16))
    h = h * np.uint32(0x7FEB352D)
    h = h ^ (h >> np.uint32(15))
    h = h * np.uint32(0x846CA68B)
    h = h ^ (h >> np.uint32(16))
    return h


def amg_handshake_pass(rp, ci, v, theta, variant=0, rounds=AMG_HS_ROUNDS):
    """one pairwise pass by handshaking: (cid, nc) — aggregates numbered in the order of their smallest row"""
    n = len(rp) - 1
    rp, ci = np.asarray(rp, dtype=np.int64), np.asarray(ci, dtype=np.int64)
    row = np.repeat(np.arange(n, dtype=np.int64), np.diff(rp))
    off = ci != row
    diag = np.zeros(n)
    np.add.at(diag, row[~off], v[~off])
    sg = np.where(diag < 0, -1.0, 1.0)
    s = -v * sg[row]
    smax = np.zeros(n)
    np.maximum.at(smax, row[off], s[off])
    cand = off & (s > 0) & (s >= theta * smax[row])
    a, b = np.minimum(row, ci), np.maximum(row, ci)
    d = np.maximum(b - a, 1)
    k2 = d if variant == 1 else (np.int64(0x7FFFFFFF) - d)
    key = (k2.astype(np.uint64) << np.uint64(33)) | ((np.uint64(1) - ((a // d) & 1).astype(np.uint64)) << np.uint64(32)) \
        | amg_edge_hash(a, b).astype(np.uint64)
    match = -np.ones(n, dtype=np.int64)
    me = np.arange(n, dtype=np.int64)
    for _ in range(rounds):
        ok = cand & (match[row] < 0) & (match[ci] < 0)
        idx = np.nonzero(ok)[0]
        if len(idx) == 0:
            break
        o = idx[np.lexsort((ci[idx], key[idx], s[idx], row[idx]))]      # per row: ascending (s, key, column) — the last one is named
        last = np.ones(len(o), dtype=bool)
        last[:-1] = row[o][1:] != row[o][:-1]
        best = -np.ones(n, dtype=np.int64)
        best[row[o][last]] = ci[o][last]
        mutual = (best >= 0) & (best[np.maximum(best, 0)] == me)
        match[mutual] = best[mutual]
    rep = (match < 0) | (me < match)
    num = np.cumsum(rep) - 1
    cid = np.where(rep, num, num[np.maximum(match, 0)])
    return cid.astype(np.int64), int(rep.sum())


def amg_pairwise_pass(rp, ci, v, theta):
    n = len(rp) - 1
    cid = -np.ones(n, dtype=np.int64)
    nc = 0
    for i in range(n):
        if cid[i] >= 0:
            continue
        a, e = rp[i], rp[i + 1]
        cols, vals = ci[a:e], v[a:e]
        dg = vals[cols == i].sum() if np.any(cols == i) else 0.0
        sg = -1.0 if dg < 0 else 1.0
        smax, best, bv = 0.0, -1, 0.0
        for k in range(e - a):
            j = cols[k]
            if j == i:
                continue
            sv = -vals[k] * sg
            if sv > smax:
                smax = sv
            if cid[j] < 0 and sv > bv:
                bv, best = sv, j
        cid[i] = nc
        if best >= 0 and bv >= theta * smax:
            cid[best] = nc
        nc += 1
    return cid, nc


def amg_galerkin(rp, ci, v, cid, nc):
    """TᵀA T for the aggregate map cid: (rowptr, col, val, entry_map) — entry_map[k] = the coarse entry fine entry k is summed
    into; values summed in ascending k"""
    n = len(rp) - 1
    rows = np.repeat(np.arange(n, dtype=np.int64), np.diff(rp))
    key = cid[rows] * np.int64(nc) + cid[ci]
    order = np.argsort(key, kind="stable")
    ks = key[order]
    first = np.ones(len(ks), dtype=bool)
    first[1:] = ks[1:] != ks[:-1]
    eid = np.cumsum(first) - 1
    emap = np.empty(len(ks), dtype=np.int64)
    emap[order] = eid
    ukey = ks[first]
    crow, ccol = ukey // nc, ukey % nc
    crp = np.zeros(nc + 1, dtype=np.int64)
    np.add.at(crp, crow + 1, 1)
    crp = np.cumsum(crp)
    cv = np.zeros(len(ukey))
    np.add.at(cv, emap, v)          # sequential, in ascending k
    return crp, ccol.astype(np.int64), cv, emap


class AggregationAMG:
    def __init__(self, A, nu=2, passes=2, theta=0.25, overcorrection=1.8, cheb_ratio=4.0, coarse_max=128, max_levels=24,
                 matching="handshake"):
        A = sp.csr_matrix(A)
        A.sort_indices()
        self.nu, self.passes, self.theta, self.omega = int(nu), int(passes), float(theta), float(overcorrection)
        self.ratio, self.coarse_max = float(cheb_ratio), int(coarse_max)
        self.matching = matching
        self.levels = []      # dicts: rp, ci (pattern), agg (row → coarse row), nc, emap (fine entry → coarse entry)
        rp, ci, v = A.indptr.astype(np.int64), A.indices.astype(np.int64), A.data.astype(np.float64)

        def coarsen(passfn):
            agg = np.arange(len(rp) - 1, dtype=np.int64)
            crp, cci, cv, nc = rp, ci, v, len(rp) - 1
            for p_ in range(self.passes):
                cid, nc = passfn(crp, cci, cv, self.theta)
                if p_ + 1 < self.passes:
                    crp, cci, cv, _e = amg_galerkin(crp, cci, cv, cid, nc)
                agg = cid[agg]
            return agg, nc

        while len(rp) - 1 > self.coarse_max and len(self.levels) < max_levels:
            n = len(rp) - 1
            if matching == "greedy":
                agg, nc = coarsen(amg_pairwise_pass)
            else:
                agg, nc = coarsen(lambda *a: amg_handshake_pass(*a, variant=0))
                if nc * (1 << self.passes) > n + n // 50:      # not (nearly) a full coarsening: try the other tie-break
                    agg1, nc1 = coarsen(lambda *a: amg_handshake_pass(*a, variant=1))
                    if nc1 < nc:
                        agg, nc = agg1, nc1
            if nc > 0.8 * n:
                break
            nrp, nci, nv, emap = amg_galerkin(rp, ci, v, agg, nc)   # one-stage sums: what a value refresh recomputes
            self.levels.append(dict(rp=rp, ci=ci, agg=agg, nc=nc, emap=emap, n=n))
            rp, ci, v = nrp, nci, nv
        self.coarse = dict(rp=rp, ci=ci, n=len(rp) - 1)
        self.update(A)

    def sizes(self):
        return [L["n"] for L in self.levels] + [self.coarse["n"]]

    def _values_on_pattern(self, A):
        """A's values laid out on the creation-time pattern (SciPy drops entries that happen to be exactly zero — the device's
        pattern is structural and keeps them: a matrix whose pattern is a SUBSET of the creation pattern is padded with zeros)"""
        A = sp.csr_matrix(A)
        A.sort_indices()
        rp, ci = (self.levels[0] if self.levels else self.coarse)["rp"], (self.levels[0] if self.levels else self.coarse)["ci"]
        if A.nnz == len(ci) and np.array_equal(A.indptr, rp) and np.array_equal(A.indices, ci):
            return A.data.astype(np.float64)
        n = len(rp) - 1
        pk = np.repeat(np.arange(n, dtype=np.int64), np.diff(rp)) * n + ci
        ak = np.repeat(np.arange(n, dtype=np.int64), np.diff(A.indptr)) * n + A.indices
        pos = np.searchsorted(pk, ak)
        if np.any(pos >= len(pk)) or np.any(pk[np.minimum(pos, len(pk) - 1)] != ak):
            raise ValueError("AMG update: the matrix has entries outside the pattern the hierarchy was built on")
        v = np.zeros(len(ci))
        v[pos] = A.data
        return v

    def update(self, A):
        """new values on the creation-time pattern"""
        v = self._values_on_pattern(A)
        for L in self.levels:
            self._numbers(L, v)
            nv = np.zeros(int(L["emap"].max()) + 1 if len(L["emap"]) else 0)
            np.add.at(nv, L["emap"], v)
            v = nv
        C_ = self.coarse
        self._numbers(C_, v)
        C_["dense"] = C_["n"] <= self.coarse_max
        if C_["dense"]:
            C_["inv"] = np.linalg.inv(C_["A"].toarray()) if C_["n"] else np.zeros((0, 0))
        return self

    @staticmethod
    def _numbers(L, v):
        n = L["n"]
        A = sp.csr_matrix((v, L["ci"], L["rp"]), shape=(n, n))
        d = A.diagonal()
        if np.any(d == 0.0) or not np.all(np.isfinite(d)):
            raise ArithmeticError("AMG: zero or non-finite diagonal entry")
        L["A"], L["dinv"] = A, 1.0 / d
        L["lmax"] = float(np.max(np.asarray(abs(A).sum(axis=1)).ravel() / np.abs(d))) if n else 1.0

    def _cheb(self, L, b, x, nsteps):
        """nsteps Chebyshev steps on D⁻¹A from x (None: zero); returns (x, r_lagged, d): r is the residual BEFORE the last d"""
        lmax = L["lmax"]
        lmin = lmax / self.ratio
        th, de = 0.5 * (lmax + lmin), 0.5 * (lmax - lmin)
        s1 = th / de
        rho = 1.0 / s1
        A, dinv = L["A"], L["dinv"]
        if x is None:
            r = b.copy()
            d = dinv * r * (1.0 / th)
            x = d.copy()
        else:
            r = b - A @ x
            d = dinv * r * (1.0 / th)
            x = x + d
        for _ in range(1, nsteps):
            rho_new = 1.0 / (2.0 * s1 - rho)
            r = r - A @ d
            d = (rho_new * rho) * d + (2.0 * rho_new / de) * (dinv * r)
            x = x + d
            rho = rho_new
        return x, r, d

    def apply(self, b, level=0):
        b = np.asarray(b, dtype=np.float64)
        if level == len(self.levels):
            C_ = self.coarse
            if C_["dense"]:
                return C_["inv"] @ b
            return self._cheb(C_, b, None, 4 * self.nu)[0]
        L = self.levels[level]
        x, r, d = self._cheb(L, b, None, self.nu)
        r = r - L["A"] @ d                                   # b − A x
        bc = np.zeros(L["nc"])
        np.add.at(bc, L["agg"], r)                           # Tᵀ r, members in ascending row order
        xc = self.apply(bc, level + 1)
        x = x + self.omega * xc[L["agg"]]
        return self._cheb(L, b, x, self.nu)[0]

    __call__ = apply


@dataclass
class LinearSolveParameters:   # lib/NonlinearSolveBase/src/linear_solve.jl:1-4
    u: object
    p: object


def sstep_block_width(want):
    """widths the device's sweeps are compiled for (csrc/nk_sstep.hip::nk_ss_block_width); other blocks are cut into these"""
    if want >= 15:
        return 15
    if want >= 8:
        return 8
    if want >= 6:
        return 6
    if want >= 4:
        return 4
    return 2 if want >= 2 else 1


def gmres_sstep(matvec: Callable, b, atol=0.0, rtol=1e-8, restart=30, itmax=300, fixed_iters=0, s=6,
                allreduce: Optional[Callable] = None, x0=None, interval=None, implicit=True):
    """Restarted GMRES(m) whose Arnoldi process advances s columns at a time — the CPU restatement of csrc/nk_sstep.hip
    (the device's NK_ORTHO_SSTEP; Hoemmen, "Communication-avoiding Krylov subspace methods", and Carson, Lund, Rozložník,
    Thomas, "Block Gram–Schmidt algorithms and their stability properties", for BCGS-PIP). Per block: the monomial vectors
    X_j = A X_{j−1} (X_0 = A v_k; the device scales them by a power of two, which changes nothing in binary floating point),
    then twice [C ; G] = [V_k X]ᵀ X in ONE reduction, RᵀR = G − CᵀC, X ← (X − V_k C) R⁻¹. With C = C₁ + C₂R₁ and R = R₂R₁ the
    monomial vectors have the coordinates F_j = [C_j ; R_j] in the new basis [V_k Q], and the Arnoldi relation of the new
    columns follows from A v_k = X_0, A X_{j−1} = X_j and q_j = (X_{j−1} − V_k C_{j−1} − Σ_{i<j} q_i R_{i,j−1}) / R_{j−1,j−1}.
    Same Krylov space, same minimisation as `gmres`; the iterates differ by rounding (1e-13 relative on the path's Jacobians).
    The stopping test sees the s columns of a block together. Raises SStepBreakdown where the device falls back to DCGS2.
    interval = (lo, hi), real bounds of the operator's spectrum: NEWTON basis X_j = (A − θ_j I) X_{j−1} / σ with
    θ_j = c + h·t_j (t the Leja-ordered Chebyshev points, c / h centre / half width) and σ = h/2 rounded to a power of two
    (Bai, Hu, Reichel 1994; Hoemmen 2010 §7.3) — then A v_k = σ X_0 + θ_0 v_k and A X_{j−1} = σ X_j + θ_j X_{j−1}, and the
    block stays well conditioned up to s = 16 (the monomial basis, interval = None, breaks down near s = 10).
    implicit = True (the device's default since round 4, NK_SS_IMPLICIT): NO block is updated a second time. A block's stored
    columns S_b = Q₁ relate to the true basis by S_b = V_true[:k0] C₂ + Q_b R₂ (pass 2's factors), and everything that needs the
    true basis goes through that relation: later blocks carry their inner products with stored columns into true coordinates
    (Q_bᵀX = R₂⁻ᵀ(S_bᵀX − C₂ᵀ V_true[:k0]ᵀX), blocks in order) and their projection coefficients back onto the stored columns
    (b = R₂⁻¹W_b, W[:k0] −= C₂ b, blocks last first); the next block's powers start from the last STORED column s = V_true u,
    u = [C₂ ; R₂][:, last], so that A v_k = (σF₀ + θ₀u − Σ_{i<k} u_i A v_i)/u_k; the solution update x += S ŷ takes ŷ from y by the
    same block-by-block map. implicit = False: every block but the cycle's last gets the explicit second update (rounds 2–3).
    Mathematically the same iterates (1e-15 … 2e-13 apart on the path's Jacobians, tests/test_oracle_pins.py)."""
    ar = allreduce if allreduce is not None else (lambda z: z)
    # (the device takes the implicit form with the NEWTON basis only: monomial blocks of 6–8 columns live at pivot ratios of
    #  1e-9 … 1e-12, below the bar the implicit form needs — they keep the explicit second update and the 1e-12 bar)
    implicit = bool(implicit) and interval is not None
    fix = []   # (k0, sb, C2, R2) of the blocks of the running cycle left at their first pass

    def to_true(P):
        P = P.copy()
        for (k0_, s_, C2_, R2_) in fix:
            P[k0_:k0_ + s_] = np.linalg.solve(R2_.T, P[k0_:k0_ + s_] - C2_.T @ P[:k0_])
        return P

    def to_stored(Wc):
        Wc = Wc.copy()
        for (k0_, s_, C2_, R2_) in reversed(fix):
            bv = np.linalg.solve(R2_, Wc[k0_:k0_ + s_])
            Wc[:k0_] -= C2_ @ bv
            Wc[k0_:k0_ + s_] = bv
        return Wc
    theta, sigma = np.zeros(max(int(s), 1)), 1.0
    if interval is not None:
        lo_, hi_ = float(interval[0]), float(interval[1])
        c_, h_ = 0.5 * (lo_ + hi_), 0.5 * (hi_ - lo_)
        if h_ > 0.0 and math.isfinite(h_) and math.isfinite(c_):
            theta = c_ + h_ * leja_chebyshev_nodes(int(s))
            sigma = 2.0 ** round(math.log2(0.5 * h_))   # (rint: ties to even, as the device's rint)
    b = np.asarray(b, dtype=np.float64)
    n = b.size
    x = np.zeros(n) if x0 is None else np.array(x0, dtype=np.float64)
    info = GmresInfo()
    r0 = b - matvec(x) if x0 is not None and np.any(x) else b.copy()
    beta0 = math.sqrt(float(ar(np.dot(r0, r0))))
    info.rnorm0 = info.rnorm = beta0
    info.residuals.append(beta0)
    if not math.isfinite(beta0):
        info.failed = True
        return x, info
    eps_stop, cap = (-1.0, int(fixed_iters)) if fixed_iters > 0 else (atol + rtol * beta0, int(itmax))
    if beta0 == 0.0 or (fixed_iters <= 0 and beta0 <= eps_stop):
        info.converged = True
        return x, info
    m = int(restart)

    def pip(V, X):  # one Pythagorean block projection: the coefficients and the triangular factor, X updated (V: STORED columns)
        k, sb = V.shape[0], X.shape[0]
        red = ar(np.concatenate([(V @ X.T).ravel(), (X @ X.T).ravel()]))
        C, G = to_true(red[: k * sb].reshape(k, sb)), red[k * sb:].reshape(sb, sb)
        S = 0.5 * ((G - C.T @ C) + (G - C.T @ C).T)
        try:
            Rm = np.linalg.cholesky(S).T
        except np.linalg.LinAlgError as e:
            raise SStepBreakdown(str(e))
        if not np.all(np.isfinite(Rm)):
            raise SStepBreakdown("non-finite factor")
        # the device's verdict (ss_factor): a pivot below 1e-12 of the column's own squared norm — the block's κ is beyond 1e6
        if np.any(np.diag(Rm) ** 2 <= 1e-12 * np.diag(G)):
            raise SStepBreakdown("pivot below 1e-12 of the column's squared norm")
        Xn = np.linalg.solve(Rm.T, X - to_stored(C).T @ V)   # rows of X are vectors: Xᵀ ← (Xᵀ − V_kᵀC) R⁻¹ (C on the stored columns)
        return C, Rm, Xn

    while True:
        steps = min(m, cap - info.iters)
        V = np.zeros((m + 1, n))
        H = np.zeros((m + 2 + s, m))
        R = np.zeros((m, m))
        cs, sn, g = np.zeros(m), np.zeros(m), np.zeros(m + 1)
        V[0] = r0 / beta0
        g[0] = beta0
        k, kdone, done = 1, 0, False
        fix.clear()
        while k - 1 < steps and not done:
            sb = sstep_block_width(min(s, steps - (k - 1)))
            if k + sb > 48 and sb > 8:
                sb = 8
            X = np.zeros((sb, n))
            z = V[k - 1]
            for j in range(sb):
                X[j] = (matvec(z) - theta[j] * z) / sigma if interval is not None else matvec(z)
                z = X[j]
            C1, R1, X = pip(V[:k], X)
            C2, R2, X2 = pip(V[:k], X)
            u = np.zeros(k)
            u[k - 1] = 1.0
            if fix and fix[-1][0] + fix[-1][1] == k:          # the powers started from the previous block's last STORED column
                u = np.concatenate([fix[-1][2][:, -1], fix[-1][3][:, -1]])
            last = (k - 1 + sb >= steps)
            if implicit and not last and max(float(np.max(np.abs(C2))), float(np.max(np.abs(R2 - np.eye(sb))))) > 0.1:
                # a block may stay at its first pass only if that pass left it NEARLY orthonormal: the next block's Hessenberg
                # recovery starts from a stored column, a combination u of true basis vectors whose images carry this block's
                # recovery errors — harmless for u ≈ e_k, amplified column by column otherwise (Arnoldi residual 5e-2 against
                # 1e-7 for the explicit update at a departure of 0.75; equal up to 0.2). The device counts it as a breakdown.
                raise SStepBreakdown("first pass too far from orthonormal for the implicit second pass")
            if implicit or last:   # (the device leaves a cycle's last block at its first pass in either mode: k_backsolve adapts y)
                V[k: k + sb] = X
                fix.append((k, sb, C2, R2))
            else:
                V[k: k + sb] = X2
            F = np.vstack([C1 + C2 @ R1, R2 @ R1])        # (k + sb) × sb
            K = k + sb
            NC = np.zeros((sb, K))
            a0 = sigma * F[:, 0]
            a0[:k] += theta[0] * u
            a0[:k] -= H[:k, : k - 1] @ u[: k - 1]
            NC[0] = a0 / u[k - 1]
            for j in range(1, sb):
                a = sigma * F[:, j] + theta[j] * F[:, j - 1]
                a[: k] -= H[:k, : k - 1] @ F[: k - 1, j - 1]
                a -= NC[0] * F[k - 1, j - 1]
                for q in range(1, j):
                    a -= NC[q] * F[k + q - 1, j - 1]
                NC[j] = a / F[k + j - 1, j - 1]
            for j in range(sb):
                jc = k - 1 + j
                H[: jc + 2, jc] = NC[j][: jc + 2]
                h = NC[j][: jc + 2].copy()
                for i in range(jc):
                    tt = cs[i] * h[i] + sn[i] * h[i + 1]
                    h[i + 1] = -sn[i] * h[i] + cs[i] * h[i + 1]
                    h[i] = tt
                dd = math.hypot(h[jc], h[jc + 1])
                cs[jc], sn[jc] = (1.0, 0.0) if dd == 0.0 else (h[jc] / dd, h[jc + 1] / dd)
                R[:jc, jc] = h[:jc]
                R[jc, jc] = dd
                g[jc + 1] = -sn[jc] * g[jc]
                g[jc] = cs[jc] * g[jc]
                info.iters += 1
                kdone = jc + 1
                info.rnorm = abs(g[jc + 1])
                info.residuals.append(info.rnorm)
                if not math.isfinite(info.rnorm):
                    info.failed = True
                    done = True
                elif (fixed_iters <= 0 and info.rnorm <= eps_stop) or h[jc + 1] == 0.0:
                    info.converged = True
                    done = True
                if done:
                    break
            k += sb
        if kdone > 0 and not info.failed:
            y = np.linalg.solve(np.triu(R[:kdone, :kdone]), g[:kdone]) if kdone > 1 else np.array([g[0] / R[0, 0]])
            Kst = (fix[-1][0] + fix[-1][1]) if fix else kdone      # stored columns behind the coefficients
            Kst = max(Kst, kdone)
            w_ = np.zeros(Kst)
            w_[:kdone] = y
            w_ = to_stored(w_.reshape(-1, 1)).ravel()             # y on the true basis → coefficients on the stored columns
            x = x + V[:Kst].T @ w_
        if done or info.iters >= cap:
            return x, info
        info.restarts += 1
        r0 = b - matvec(x)
        beta0 = math.sqrt(float(ar(np.dot(r0, r0))))


# ----------------------------------------------------------------------------- algorithm descriptors
class BratuMultigrid:
    """Geometric multigrid V-cycle for the Bratu Jacobian, the CPU restatement of csrc/nk_mg.hip (what a user would plug in
    through `precs(A, p)`, docs/src/tutorials/large_systems.md:244-316, with an AMG preconditioner). Levels n_l = n_{l−1}//2
    down to ≤ coarse_max; J_l by rediscretisation with the fine level's scale and u_l = R u_{l−1}; bilinear interpolation at
    the points' physical positions (grids need not be nested), restriction = row-normalised transpose; ν Chebyshev steps
    on [λmax/4, λmax], λmax = 8·scale/h_l²; sparse LU on the coarsest grid."""

    def __init__(self, prob: "Bratu2D", u, nu=2, coarse_max=31):
        import scipy.sparse.linalg as spla
        self.nu = int(nu) if nu > 0 else 2
        coarse_max = coarse_max if coarse_max >= 3 else 31
        scale = prob.c_lap * prob.h * prob.h
        self.levels = []
        ns, ul = prob.ns, np.asarray(u, dtype=np.float64)
        while True:
            pl = Bratu2D(ns, prob.lam, scale)
            lev = dict(ns=ns, J=sp.csr_matrix(pl.jac(ul)), lmax=8.0 * scale / (pl.h * pl.h))
            self.levels.append(lev)
            if ns <= coarse_max or ns // 2 < 3:
                break
            nc = ns // 2
            P1 = self._interp1d(ns, nc)
            P = sp.kron(P1, P1).tocsr()                      # lexicographic k = j·n + i
            Rm = P.T.tocsr()
            Rm = sp.diags(1.0 / np.asarray(Rm.sum(axis=1)).ravel()) @ Rm
            lev["P"], lev["R"] = P, sp.csr_matrix(Rm)
            ul = lev["R"] @ ul
            ns = nc
        self.lu = spla.splu(sp.csc_matrix(self.levels[-1]["J"])) if len(self.levels) > 1 else None

    @staticmethod
    def _interp1d(nf, nc):
        h, H = 1.0 / (nf + 1), 1.0 / (nc + 1)
        rows, cols, vals = [], [], []
        for i in range(nf):
            t = (i + 1) * h / H
            fl = int(math.floor(t))
            w1 = t - fl
            for I, w in ((fl - 1, 1.0 - w1), (fl, w1)):      # 0-based coarse neighbours; outside = zero boundary
                if 0 <= I < nc and w != 0.0:
                    rows.append(i), cols.append(I), vals.append(w)
        return sp.csr_matrix((vals, (rows, cols)), shape=(nf, nc))

    def _smooth(self, lev, x, b, zero_guess):
        lmax, lmin = lev["lmax"], lev["lmax"] / 4.0
        theta, delta = 0.5 * (lmax + lmin), 0.5 * (lmax - lmin)
        sigma = theta / delta
        rho = 1.0 / sigma
        J = lev["J"]
        r = b.copy() if zero_guess else b - J @ x
        d = r / theta
        x = d.copy() if zero_guess else x + d
        for _ in range(1, self.nu):
            rho_new = 1.0 / (2.0 * sigma - rho)
            r = r - J @ d
            d = rho_new * rho * d + 2.0 * rho_new / delta * r
            x = x + d
            rho = rho_new
        return x

    def __call__(self, v):
        L = self.levels
        if len(L) == 1:
            return self._smooth(L[0], None, v, True)
        bs, xs = [np.asarray(v, dtype=np.float64)], []
        for l in range(len(L) - 1):
            x = self._smooth(L[l], None, bs[l], True)
            xs.append(x)
            bs.append(L[l]["R"] @ (bs[l] - L[l]["J"] @ x))
        e = self.lu.solve(bs[-1])
        for l in range(len(L) - 2, -1, -1):
            e = self._smooth(L[l], xs[l] + L[l]["P"] @ e, bs[l], False)
        return e


class BrusselatorMultigrid(BratuMultigrid):
    """The same V-cycle for the Brusselator Jacobian (two coupled species on a periodic N × N grid; what the reference's
    problem-agnostic AMG `precs` would be handed for config C5): levels N_l = N/2^l while even and > coarse_max, level
    operators by rediscretisation (spacing 2^l·dx, linearisation point restricted by full weighting), periodic bilinear
    prolongation / full-weighting restriction per species on the nested grids, ν Chebyshev steps on −[λmax/4, λmax]
    (the Jacobian is diffusion-dominated with a NEGATIVE spectrum; λmax = 8α/dx_l² + A + 1 + 2·max|u|·max|v| + max|u|²
    bounds every Gershgorin disc on every level), direct solve on the coarsest level (where the reaction block may be
    indefinite). Restates nonlinearsolve.jl_amd/csrc/nk_mg.hip (Brusselator branch)."""

    def __init__(self, prob: "Brusselator2D", u, nu=2, coarse_max=8):
        import scipy.sparse.linalg as spla
        self.nu = nu
        self.levels = []
        N, dx = prob.N, prob.dx
        alpha = prob.alpha * prob.dx * prob.dx               # the problem stores α/dx²
        ul = np.asarray(u, dtype=np.float64)
        nn = N * N
        U, V = float(np.max(np.abs(ul[:nn]))), float(np.max(np.abs(ul[nn:])))
        rb = prob.A + 1.0 + 2.0 * U * V + U * U
        while True:
            pl = Brusselator2D(N, prob.A, prob.B, alpha, dx)
            lev = dict(ns=N, J=sp.csr_matrix(pl.jac(ul)), lmax=-(8.0 * alpha / (dx * dx) + rb))
            self.levels.append(lev)
            if N <= coarse_max or N % 2 != 0 or N // 2 < 4:
                break
            P1 = self._interp1d_periodic(N)
            P2 = sp.kron(P1, P1).tocsr()                     # one species, lexicographic k = i + N·j
            P = sp.block_diag([P2, P2]).tocsr()
            lev["P"], lev["R"] = P, (0.25 * P.T).tocsr()      # full weighting = ¼ Pᵀ (rows sum to 1)
            ul = lev["R"] @ ul
            N, dx = N // 2, 2.0 * dx
        Jc = self.levels[-1]["J"]
        self.lu = spla.splu(sp.csc_matrix(Jc)) if len(self.levels) > 1 else None

    @staticmethod
    def _interp1d_periodic(nf):
        nc = nf // 2
        rows, cols, vals = [], [], []
        for i in range(nf):
            I = i // 2
            if i % 2 == 0:
                rows.append(i), cols.append(I), vals.append(1.0)
            else:
                rows += [i, i]
                cols += [I, (I + 1) % nc]
                vals += [0.5, 0.5]
        return sp.csr_matrix((vals, (rows, cols)), shape=(nf, nc))


@dataclass
class MultigridPrecs:
    nu: int = 2
    coarse_max: int = 31


@dataclass
class ChebyshevPrecs:
    degree: int = 16
    ratio: float = 100.0


@dataclass
class KrylovJL_GMRES:
    """Krylov protocol of SURVEY.md §8d (GMRES(m), restart on, x0 = 0); `precs` = optional right preconditioner."""
    gmres_restart: int = 30
    maxiters: int = 300
    ortho: str = "mgs"
    fixed_iters: int = 0
    precs: Optional[ChebyshevPrecs] = None


@dataclass
class ObjectPrecs:
    """precs through a built-in object on the concrete J (nk_options.precond_kind): kind = "jacobi" | "ilu0" (multicolour) |
    "ilu0_natural" | "amg" (AggregationAMG); side = "left" (the reference's tutorial precs return `(Pl, I)`) or "right"."""
    kind: str = "ilu0"
    side: str = "left"


@dataclass
class DirectSolve:
    """linsolve = nothing on a sparse J: LinearSolve default sparse LU (KLU/UMFPACK [EXT]) → SuperLU."""


@dataclass
class EisenstatWalkerForcing2:  # eisenstat_walker.jl:18-29
    eta0: float = 0.5
    eta_max: float = 0.9
    gamma: float = 0.9
    alpha: float = 2.0
    safeguard: bool = True
    safeguard_threshold: float = 0.1


@dataclass
class BackTracking:
    """LineSearches.jl BackTracking [EXT] (c_1 = 1e-4, ρ_hi = 0.5, ρ_lo = 0.1, order = 3, iterations = 1000)."""
    c_1: float = 1e-4
    rho_hi: float = 0.5
    rho_lo: float = 0.1
    order: int = 3
    maxiters: int = 1000


@dataclass
class LineSearchesJL:
    """LineSearch.jl's wrapper around LineSearches.jl [EXT] — `NewtonRaphson(linesearch = LineSearchesJL(; method = …))`, the
    five methods the reference's own tests run (lib/NonlinearSolveFirstOrder/test/rootfind_tests__item2.jl:40-46): `Static`,
    `BackTracking`, `StrongWolfe`, `MoreThuente`, `HagerZhang` are restated here from the published algorithms (Nocedal & Wright
    alg. 3.5/3.6; Moré & Thuente 1994 / MINPACK cvsrch + cstep; Hager & Zhang 2005) with LineSearches.jl's default parameters.
    ϕ(α) = ½‖f(u + α δu)‖², ϕ'(α) = f(u + α δu)ᵀ J(u + α δu) δu; every ϕ, ϕ' or (ϕ, ϕ') evaluation is one residual
    evaluation (`nf += 1`); α₀ = 1 on every call; a non-descent direction (ϕ'(0) ≥ 0) takes the full step and reports
    failure. Parity unpinned (no source in the tree; the reference only tests convergence)."""
    method: str = "BackTracking"


@dataclass
class NewtonRaphson:  # raphson.jl:30-43
    linsolve: object = None
    forcing: Optional[EisenstatWalkerForcing2] = None
    concrete_jac: Optional[bool] = None
    linesearch: Optional[BackTracking] = None
    name: str = "NewtonRaphson"


@dataclass
class GaussNewton:  # gauss_newton.jl:11-23 on a NonlinearLeastSquaresProblem: NewtonDescent in normal form when the
    linsolve: object = None   # linear solver needs a square A (descent/newton.jl:58-95): JᵀJ δ = Jᵀ f
    concrete_jac: Optional[bool] = None
    linesearch: Optional[BackTracking] = None
    forcing: Optional[EisenstatWalkerForcing2] = None
    name: str = "GaussNewton"


@dataclass
class LevenbergMarquardt:  # levenberg_marquardt.jl:37-64: GeodesicAcceleration(DampedNewtonDescent(LM damping)) +
    linsolve: object = None      # LevenbergMarquardtTrustRegion, concrete_jac = Val(true)
    damping_initial: float = 1.0
    alpha_geodesic: float = 0.75
    disable_geodesic: bool = False
    damping_increase_factor: float = 2.0
    damping_decrease_factor: float = 3.0
    finite_diff_step_geodesic: float = 0.1
    b_uphill: float = 1.0
    min_damping_D: float = 1e-8
    concrete_jac: Optional[bool] = True
    name: str = "LevenbergMarquardt"


@dataclass
class PseudoTransient:  # pseudo_transient.jl:37-57: DampedNewtonDescent(SwitchedEvolutionRelaxation), no globalisation
    # mass_matrix (pseudo_transient.jl:102-149): None / 1.0 → identity damping α⁻¹ (resolve_ser_mass_matrix maps I to nothing);
    # a number λ → λI; a vector → Diagonal; a 2-D array / sparse matrix → general M.  Damping term D = α⁻¹ M.
    linsolve: object = None
    alpha_initial: float = 1e-3
    concrete_jac: Optional[bool] = None
    linesearch: Optional[BackTracking] = None
    mass_matrix: object = None
    name: str = "PseudoTransient"

    def mass(self, n, prob=None):
        M = self.mass_matrix
        if M is None:   # resolve_ser_mass_matrix(::Nothing, prob): the NonlinearFunction's own mass matrix, if it has a real one
            M = getattr(prob, "mass_matrix", None)
        if M is None:
            return None
        if sp.issparse(M):
            M = sp.csr_matrix(M)
        elif np.ndim(M) == 0:
            return None if float(M) == 1.0 else sp.identity(n, format="csr") * float(M)
        elif np.ndim(M) == 1:
            M = sp.diags(np.asarray(M, dtype=float)).tocsr()
        else:
            M = sp.csr_matrix(np.asarray(M, dtype=float))
        if M.shape != (n, n):   # DimensionMismatch (pseudo_transient.jl:110-118)
            raise ValueError(f"mass matrix has size {M.shape} but the problem has {n} unknowns")
        return M


SIMPLE, NLSOLVE, NOCEDAL_WRIGHT, HEI, YUAN, BASTIN, FAN = range(7)


@dataclass
class TrustRegion:  # trust_region.jl:25-43 ; 0 means "scheme default" (trust_region.jl:320-328)
    linsolve: object = None
    radius_update_scheme: int = SIMPLE
    max_trust_radius: float = 0.0
    initial_trust_radius: float = 0.0
    step_threshold: float = 1.0 / 10000
    shrink_threshold: float = 1.0 / 4
    expand_threshold: float = 3.0 / 4
    shrink_factor: float = 1.0 / 4
    expand_factor: float = 2.0
    max_shrink_times: int = 32
    concrete_jac: Optional[bool] = None
    name: str = "TrustRegion"


@dataclass
class Stats:  # NLStats
    nf: int = 0
    njacs: int = 0
    nfactors: int = 0
    nsolve: int = 0
    nsteps: int = 0
    gmres_iters: int = 0


@dataclass
class Solution:
    u: np.ndarray
    resid: np.ndarray
    retcode: int
    stats: Stats
    trace: list


# ----------------------------------------------------------------------------- termination cache
(TM_ABSNORM_SAFEBEST, TM_NORM, TM_REL, TM_RELNORM, TM_RELNORM_SAFE, TM_RELNORM_SAFEBEST, TM_ABS, TM_ABSNORM,
 TM_ABSNORM_SAFE) = range(9)
_SAFE = (TM_ABSNORM_SAFEBEST, TM_ABSNORM_SAFE, TM_RELNORM_SAFE, TM_RELNORM_SAFEBEST)
_BEST = (TM_ABSNORM_SAFEBEST, TM_RELNORM_SAFEBEST)
_RELSAFE = (TM_RELNORM_SAFE, TM_RELNORM_SAFEBEST)


class TerminationCache:
    """NonlinearTerminationModeCache functor for the nine SciMLBase modes — termination_conditions.jl:243-376.
    Mode 0 with max_stalled_steps = 32 is the solver default (:385-389). `norm` is the mode's internalnorm:
    "inf" = Base.Fix1(maximum, abs), "l2" = Base.Fix2(norm, 2); apply_norm(f, du, u) = f(du .+ u) (utils.jl:102)."""

    def __init__(self, fu, u, abstol, reltol=DEFAULT_TOL, mode=TM_ABSNORM_SAFEBEST, norm="inf", patience_steps=100,
                 patience_objective_multiplier=3.0, min_max_factor=1.3, max_stalled_steps=32,
                 protective_threshold=None, leastsq=False):
        self.abstol, self.reltol, self.mode = abstol, reltol, mode
        self.nrm = Linf_NORM if norm == "inf" else L2_NORM
        self.patience_steps = patience_steps
        self.pom = patience_objective_multiplier
        self.min_max_factor = min_max_factor
        self.max_stalled_steps = max_stalled_steps
        self.protective_threshold = protective_threshold
        self.leastsq = leastsq
        self.reinit(fu, u)

    def _objective(self, du, u):
        if self.mode in _RELSAFE:
            return self.nrm(du) / (self.nrm(du + u) + np.finfo(float).eps * self.reltol)  # eps(reltol)
        return self.nrm(du)

    def reinit(self, fu, u):
        self.u = np.array(u, copy=True) if self.mode in _BEST else None
        self.retcode = DEFAULT
        self.nsteps = 0
        if self.mode in _SAFE:
            self.initial_objective = self._objective(fu, u)
            self.u0_norm = L2_NORM(u) if (self.mode in _RELSAFE and self.max_stalled_steps is not None) else None
        else:
            self.initial_objective = float("inf")
        self.best_objective_value = self.initial_objective
        self.objectives_trace = np.zeros(self.patience_steps)
        self.step_norm_trace = None if self.max_stalled_steps is None else np.zeros(self.max_stalled_steps)

    def _check_convergence(self, du, u):  # termination_conditions.jl:338-376
        m = self.mode
        if m == TM_REL:
            return bool(np.all(np.abs(du) <= self.reltol * np.abs(u + du)))
        if m == TM_ABS:
            return bool(np.all(np.abs(du) <= self.abstol))
        dn = self.nrm(du)
        if m == TM_NORM:
            return dn <= self.abstol or dn <= self.reltol * self.nrm(du + u)
        if m == TM_RELNORM:
            return dn <= self.reltol * self.nrm(du + u)
        return dn <= self.abstol  # TM_ABSNORM

    def __call__(self, du, u, uprev):
        if self.mode not in _SAFE:
            if self._check_convergence(du, u):
                self.retcode = SUCCESS
                return True
            return False
        objective = self._objective(du, u)
        criteria = self.reltol if self.mode in _RELSAFE else self.abstol
        if not math.isfinite(objective):
            self.retcode = UNSTABLE
            return True
        if self.protective_threshold is not None and \
                objective > self.initial_objective * self.protective_threshold * du.size:
            self.retcode = UNSTABLE
            return True
        if self.mode in _BEST and objective < self.best_objective_value:
            self.best_objective_value = objective
            self.u[...] = u
        if objective <= criteria:
            self.retcode = SUCCESS
            return True
        self.nsteps += 1
        L = len(self.objectives_trace)
        self.objectives_trace[(self.nsteps - 1) % L] = objective  # mod1
        if objective <= self.pom * criteria and self.nsteps > self.patience_steps:
            tr = self.objectives_trace[: self.nsteps] if self.nsteps < L else self.objectives_trace
            if tr.min() < self.min_max_factor * tr.max():
                self.retcode = STALLED
                return True
        if self.step_norm_trace is not None:
            du_norm = L2_NORM(u - uprev)
            self.step_norm_trace[(self.nsteps - 1) % len(self.step_norm_trace)] = du_norm
            if self.nsteps > self.max_stalled_steps:
                mx = self.step_norm_trace.max()
                stalled = (mx <= self.reltol * (mx + self.u0_norm)) if self.mode in _RELSAFE else (mx <= self.abstol)
                if stalled:
                    self.retcode = STALLED
                    return True
        self.retcode = FAILURE
        return False


# ----------------------------------------------------------------------------- the cache
class FirstOrderCache:
    """GeneralizedFirstOrderAlgorithmCache — `init`, `step!`, `solve!`, `reinit!`."""

    def __init__(self, prob: Problem, alg, abstol=None, reltol=None, maxiters=1000, u0=None,
                 termination_kwargs=None, store_trace=True, lin_x0_zero=True):
        self.prob, self.alg = prob, alg
        self.abstol = DEFAULT_TOL if abstol is None else float(abstol)
        self.reltol = DEFAULT_TOL if reltol is None else float(reltol)
        self.maxiters = int(maxiters)
        self.termination_kwargs = termination_kwargs or {}
        self.store_trace = store_trace
        self.is_tr = isinstance(alg, TrustRegion)
        self.is_lm = isinstance(alg, LevenbergMarquardt)
        ls = alg.linsolve
        self.krylov = ls if isinstance(ls, KrylovJL_GMRES) else None
        # jacobian.jl:43-47 — concrete J needed unless a Krylov method without concrete_jac
        self.concrete = (self.krylov is None) or bool(alg.concrete_jac)
        self._init(prob.u0() if u0 is None else np.asarray(u0, dtype=np.float64))

    # -- SciMLBase.__init (FirstOrder/src/solve.jl:140-301)
    def _init(self, u0):
        prob = self.prob
        self.u = np.array(u0, dtype=np.float64, copy=True)
        self.fu = prob.f(self.u)
        self.u_cache = self.u.copy()
        self.stats = Stats()
        self.nsteps = 0
        self.retcode = DEFAULT
        self.force_stop = False
        self.make_new_jacobian = True
        self.tc = TerminationCache(self.fu, self.u, self.abstol, self.reltol, **self.termination_kwargs)
        self.J = None
        if self.concrete:
            self.J = prob.jac(self.u)  # jacobian.jl:104-118 (evaluated once "to get the type")
            self.stats.njacs += 1
        self.du = np.zeros_like(self.u)  # descent/newton.jl:34-36
        kr0 = getattr(self, "krylov", None)
        if kr0 is not None and callable(kr0.precs) and not getattr(self, "_precs_inited", False):
            # construct_linear_solver → init(linprob, linsolve; …) (linear_solve.jl:74-118): LinearSolve evaluates
            # `precs(A, p)` when it builds the cache [EXT]; p = LinearSolveParameters(u, prob.p) (:1-4). reinit! does not
            # rebuild the cache (FirstOrder/src/solve.jl:108-133 → reinit!(lincache; p) only swaps p): no call there.
            self._precs_inited = True
            kr0.precs(self.J if self.concrete else self._operator_view(self.u), LinearSolveParameters(self.u, getattr(prob, "p", None)))
        self.trace = []
        self.eta = float("nan")
        forcing = getattr(self.alg, "forcing", None)
        self.forcing = forcing if (forcing is not None and self.krylov is not None) else None
        if self.forcing is not None:  # eisenstat_walker.jl:92-101
            self.ew_eta = self.forcing.eta0
            self.ew_rnorm = self.ew_rnorm_prev = L2_NORM(self.fu)
        self.lin_reltol = self.reltol  # FirstOrder/src/solve.jl:203
        self.lin_abstol = self.abstol
        if self.is_tr:
            self._tr_init(self.u, self.fu)
        if self.is_lm:
            self._lm_init(self.u)
        self.is_pt = isinstance(self.alg, PseudoTransient)
        if self.is_pt:   # SwitchedEvolutionRelaxationCache (pseudo_transient.jl:107-124)
            self.pt_ainv = 1.0 / float(self.alg.alpha_initial)
            self.pt_res = L2_NORM(self.fu)

    # -- LevenbergMarquardt: damping cache (levenberg_marquardt.jl:72-117), LevenbergMarquardtTrustRegionCache (:204-245),
    #    GeodesicAccelerationCache (geodesic_acceleration.jl:51-55); also what their reinit! methods restore
    def _lm_init(self, u):
        a = self.alg
        self.lm_lam = float(a.damping_initial)
        self.lm_lam_factor = float(a.damping_increase_factor)
        self.lm_DtD = np.full(u.size, float(a.min_damping_D))
        self.lm_Jdamped = self.lm_lam * self.lm_DtD
        self.lm_v_cache = np.array(u, copy=True)      # `@bb v = copy(u)` (:212) — the iterate, not a velocity
        self.lm_norm_v_old = float("inf")
        self.lm_loss_old = float("inf")               # never written again by the reference (:250-268)
        self.lm_tr_accepted = False
        self.lm_geo_accepted = False
        self.lm_beta = float("nan")

    # DampedNewtonDescent.solve! (descent/damped_newton.jl:224-345) in the two modes a NonlinearProblem reaches:
    # :normal_form when the linear solver needs a square A (Krylov), :least_squares otherwise (linsolve = nothing → QR of
    # [J; √(λDᵀD)]). recompute_A = (idx === Val(1)): the velocity solve refreshes DᵀD and the damped matrix, the
    # acceleration solve reuses them. Returns δu = −x, or None when the linear solve reports failure.
    def _lm_damped_solve(self, rhs, recompute_A):
        self.stats.nsolve += 1
        J = self.J
        if recompute_A:   # (J !== nothing || new_jacobian) && recompute_A — a concrete J is never `nothing`
            diag = np.asarray(J.multiply(J).sum(axis=0)).ravel() if sp.issparse(J) else np.sum(np.asarray(J) ** 2, axis=0)
            self.lm_DtD = np.maximum(self.lm_DtD, diag)   # update_levenberg_marquardt_diagonal!! (:270-293)
            self.lm_Jdamped = self.lm_lam * self.lm_DtD   # @. J_damped = λ * DᵀD
        D = self.lm_Jdamped
        if self.krylov is not None:    # :normal_form — (JᵀJ + λDᵀD) x = Jᵀ rhs
            kr = self.krylov
            b = J.T @ rhs
            op = lambda v: J.T @ (J @ v) + D * v  # noqa: E731
            x, info = gmres(op, b, None, atol=self.lin_abstol, rtol=self.lin_reltol, restart=kr.gmres_restart,
                            itmax=kr.maxiters, fixed_iters=kr.fixed_iters, ortho=kr.ortho)
            self.stats.gmres_iters += info.iters
            self.last_gmres = info
            if info.failed:
                return None
        else:                          # :least_squares — min ‖[J; √D] x − [rhs; 0]‖ (QR in the reference)
            n = J.shape[1]
            if n <= 4000:
                Jd = J.toarray() if sp.issparse(J) else np.asarray(J)
                A = np.vstack([Jd, np.diag(np.sqrt(D))])
                x = np.linalg.lstsq(A, np.concatenate([rhs, np.zeros(n)]), rcond=None)[0]
            else:                      # large sparse J: the same minimiser through the normal equations (SPQR is [EXT])
                import scipy.sparse.linalg as spla
                x = spla.spsolve(sp.csc_matrix(J.T @ J + sp.diags(D)), J.T @ rhs)
            if not np.all(np.isfinite(x)):
                return None
        return -x

    # GeodesicAcceleration.solve! (descent/geodesic_acceleration.jl:98-136); without it the damped Newton step itself
    def _lm_descent(self):
        a = self.alg
        v = self._lm_damped_solve(self.fu, True)
        if a.disable_geodesic:
            if v is None:
                return None, False, None
            return v, True, v
        if v is None:            # geodesic's solve! reads `.δu` only: an inner linear-solve failure is not propagated
            v = self.du.copy()
        h = float(a.finite_diff_step_geodesic)
        fu_c = self.prob.f(self.u + h * v)                     # Utils.evaluate_f!! — not counted in stats.nf
        Jv = self.J @ v
        fu_c = (2.0 / h) * ((fu_c - self.fu) / h - Jv)
        acc = self._lm_damped_solve(fu_c, False)
        if acc is None:
            acc = np.zeros_like(v)
        self.lm_geo_accepted = bool(2.0 * L2_NORM(acc) <= L2_NORM(v) * float(a.alpha_geodesic))
        du = v + acc / 2.0 if self.lm_geo_accepted else self.du  # a rejected step leaves δu as it was
        return du, self.lm_geo_accepted, v

    # LevenbergMarquardtTrustRegionCache solve! (levenberg_marquardt.jl:247-268)
    def _lm_tr_solve(self, du, v):
        norm_v = L2_NORM(v)
        with np.errstate(invalid="ignore", divide="ignore"):
            beta = float(np.float64(np.dot(v, self.lm_v_cache)) / (np.float64(norm_v) * np.float64(self.lm_norm_v_old)))
        self.lm_beta = beta
        u_new = self.u + du
        fu_new = self.prob.f(u_new)
        self.stats.nf += 1
        loss = L2_NORM(fu_new)
        try:
            lhs = math.pow(1.0 - beta, float(self.alg.b_uphill)) * loss
        except ValueError:   # negative base, fractional exponent: Julia throws a DomainError here
            lhs = float("nan")
        if lhs <= self.lm_loss_old:
            self.lm_tr_accepted = True
            self.lm_norm_v_old = norm_v
            self.lm_v_cache = np.array(v, copy=True)
        else:
            self.lm_tr_accepted = False
        return self.lm_tr_accepted, u_new, fu_new

    # callback_into_cache!(topcache, ::LevenbergMarquardtDampingCache) (levenberg_marquardt.jl:159-168)
    def _lm_callback(self):
        geo_ok = True if self.alg.disable_geodesic else self.lm_geo_accepted   # last_step_accepted default: true
        if self.lm_tr_accepted and geo_ok:
            self.lm_lam_factor = 1.0 / float(self.alg.damping_decrease_factor)
        self.lm_lam *= self.lm_lam_factor
        self.lm_lam_factor = float(self.alg.damping_increase_factor)

    # -- trust_region.jl:204-258 (+ defaults :320-384)
    def _tr_defaults(self):
        a, m = self.alg, self.alg.radius_update_scheme
        def pick(val, default):
            return default if val == 0 else float(val)
        self.step_threshold = pick(a.step_threshold, {HEI: 0.0, YUAN: 1e-3, BASTIN: 0.05}.get(m, 1e-4))
        self.shrink_threshold = pick(a.shrink_threshold, {HEI: 0.0, NLSOLVE: 0.05, BASTIN: 0.05}.get(m, 0.25))
        self.expand_threshold = pick(a.expand_threshold, {NLSOLVE: 0.9, HEI: 0.0, BASTIN: 0.9}.get(m, 0.75))
        self.shrink_factor = pick(a.shrink_factor, {NLSOLVE: 0.5, HEI: 0.0, BASTIN: 0.05}.get(m, 0.25))
        self.expand_factor = pick(a.expand_factor, 2.0)
        self.p1, self.p2, self.p3, self.p4 = {
            NLSOLVE: (0.5, 0.0, 0.0, 0.0), HEI: (5.0, 0.1, 0.15, 0.15), YUAN: (2.0, 1.0 / 6, 6.0, 0.0),
            FAN: (0.1, 0.25, 12.0, 1e18), BASTIN: (2.5, 0.25, 0.0, 0.0)}.get(m, (0.0, 0.0, 0.0, 0.0))

    def _tr_radii(self, u, fu):
        a, m = self.alg, self.alg.radius_update_scheme
        u0_norm, fu_norm = L2_NORM(u), L2_NORM(fu)
        if a.max_trust_radius != 0:
            mtr = float(a.max_trust_radius)
        elif m in (SIMPLE, NOCEDAL_WRIGHT):
            mtr = max(fu_norm, float(np.max(u) - np.min(u)))
        else:
            mtr = float("inf")
        if a.initial_trust_radius != 0:
            itr = float(a.initial_trust_radius)
        elif m == NLSOLVE:
            itr = u0_norm if u0_norm > 0 else 1.0
        elif m in (HEI, BASTIN):
            itr = 1.0
        elif m == FAN:
            itr = (fu_norm ** 0.99) / 10
        else:
            itr = mtr / 11
        return mtr, itr

    def _tr_init(self, u, fu):
        self._tr_defaults()
        self.max_trust_radius, self.initial_trust_radius = self._tr_radii(u, fu)
        if self.alg.radius_update_scheme == YUAN:
            self.initial_trust_radius = self.p1 * L2_NORM(self._apply_JT(fu, u))
        self.trust_region = self.initial_trust_radius
        self.rho = 0.0
        self.shrink_counter = 0
        self.last_step_accepted = False
        self.tr_du_cache = np.zeros_like(u)

    def _operator_view(self, u):
        """what `precs` receives as A on the matrix-free path: the StatefulJacobianOperator (a callable v ↦ J(u) v here)"""
        return lambda v: self.prob.jvp(v, u)

    # -- operators (jacobian.jl:237-262, SciMLJacobianOperators.jl:238-243)
    def _apply_J(self, v, u):
        self.stats_op = getattr(self, "stats_op", 0) + 1
        return (self.J @ v) if self.concrete else self.prob.jvp(v, u)

    def _apply_JT(self, v, u):
        return (self.J.T @ v) if self.concrete else self.prob.vjp(v, u)

    # -- NewtonDescent.solve! (descent/newton.jl:97-141) through LinearSolveJLCache (ext:16-32)
    def _newton_descent(self, new_jacobian):
        self.stats.nsolve += 1
        shift = 0.0
        if getattr(self, "is_pt", False):
            # SwitchedEvolutionRelaxation solve! (pseudo_transient.jl:152-164): α⁻¹ ← α⁻¹·‖f‖/‖f_prev‖, then
            # dampen_jacobian!!(J_cache, J, α⁻¹) (damped_newton.jl:283-288,352-366): (J + α⁻¹ I) δ = f
            res = L2_NORM(self.fu)
            self.pt_ainv = float(np.float64(self.pt_ainv) * (np.float64(res) / np.float64(self.pt_res)))   # 0/0 → NaN as in Julia
            self.pt_res = res
            shift = self.pt_ainv
            Mm = self.alg.mass(self.u.size, self.prob)
            Dm = (lambda v: shift * v) if Mm is None else (lambda v: shift * (Mm @ v))
        if self.krylov is not None:
            kr = self.krylov
            u_now = self.u
            M, Ml = None, None
            if isinstance(kr.precs, MultigridPrecs):  # precs(A, p) re-evaluated at the current u
                MG = BrusselatorMultigrid if isinstance(self.prob, Brusselator2D) else BratuMultigrid
                M = MG(self.prob, u_now, kr.precs.nu, kr.precs.coarse_max)
            elif isinstance(kr.precs, ObjectPrecs):   # a built-in object refactorised for the current concrete J
                assert self.concrete, "ObjectPrecs needs a concrete J"
                if kr.precs.kind == "amg":   # aggregates fixed when the object is built, numbers refreshed for every new J
                    if getattr(self, "_amg", None) is None:
                        # the device builds the hierarchy on the STRUCTURAL pattern of J (entries that are zero at this u included)
                        S = sp.csr_matrix(self.prob.jac(np.full(self.prob.n, 0.37)))
                        S.sort_indices()
                        Jf = sp.csr_matrix(self.J)
                        Jf.sort_indices()
                        if S.nnz != Jf.nnz:
                            n_ = S.shape[0]
                            pk = np.repeat(np.arange(n_, dtype=np.int64), np.diff(S.indptr)) * n_ + S.indices
                            ak = np.repeat(np.arange(n_, dtype=np.int64), np.diff(Jf.indptr)) * n_ + Jf.indices
                            vv = np.zeros(S.nnz)
                            vv[np.searchsorted(pk, ak)] = Jf.data
                            Jf = sp.csr_matrix((vv, S.indices.copy(), S.indptr.copy()), shape=S.shape)
                        self._amg = AggregationAMG(Jf)
                    elif new_jacobian:
                        self._amg.update(self.J)
                    Pm = self._amg
                else:
                    Pm = jacobi_preconditioner(self.J) if kr.precs.kind == "jacobi" else \
                        ilu0_preconditioner(self.J, "natural" if kr.precs.kind == "ilu0_natural" else "multicolor")
                M, Ml = (Pm, None) if kr.precs.side == "right" else (None, Pm)
            elif callable(kr.precs):
                # the `precs(A, p) -> (Pl, Pr)` hook: a new A marks the LinearSolve cache fresh (ext/NonlinearSolveBase
                # LinearSolveExt.jl:64-111 update_A! → set_lincache_A!), and LinearSolve re-evaluates precs for a fresh A
                # [EXT] — once per linear solve with a new Jacobian; pinned by test/Core/core_tests__item21.jl:10-37.
                if new_jacobian or getattr(self, "_precs_pair", None) is None:
                    out = kr.precs(self.J if self.concrete else self._operator_view(u_now),
                                   LinearSolveParameters(u_now, getattr(self.prob, "p", None)))
                    self._precs_pair = out if isinstance(out, tuple) else (out, None)
                Ml, M = self._precs_pair      # wrap_preconditioners (linear_solve.jl:195-199): nothing → identity
            elif kr.precs is not None:  # precs(A, p) re-evaluated for the current J (concrete J: Gershgorin bound)
                assert self.concrete, "the oracle's Chebyshev precs needs a concrete J (Gershgorin bound)"
                lmax = gershgorin_lambda(self.J)
                M = chebyshev_preconditioner(lambda v: self._apply_J(v, u_now), lmax / kr.precs.ratio, lmax,
                                             kr.precs.degree)
            if isinstance(self.alg, GaussNewton):   # normal form (descent/newton.jl:107-118): JᵀJ δ = Jᵀ fu
                rhs = self._apply_JT(self.fu, u_now)
                op = lambda v: self._apply_JT(self._apply_J(v, u_now), u_now)  # noqa: E731
            else:
                rhs, op = self.fu, ((lambda v: self._apply_J(v, u_now) + Dm(v)) if shift else (lambda v: self._apply_J(v, u_now)))
            ortho = kr.ortho
            if isinstance(ortho, tuple) and ortho[0] == "sstep" and len(ortho) == 3 and ortho[2] in ("auto", "newton"):
                # the device's choice (nk_ss_prepare): Newton basis where it can bound the spectrum — Gershgorin discs of a
                # concrete J, the closed form of the Bratu stencil — and no preconditioner / normal form / shift is in the way
                interval = None
                if M is None and Ml is None and not shift and not isinstance(self.alg, GaussNewton):
                    if self.concrete:
                        interval = gershgorin_interval(self.J)
                    elif isinstance(self.prob, Bratu2D):
                        d_ = self.prob.c_exp * np.exp(u_now)
                        interval = (-float(np.max(d_)), 8.0 * self.prob.c_lap - float(np.min(d_)))
                ortho = ("sstep", ortho[1], "newton", interval) if interval is not None else ("sstep", ortho[1])
            x, info = gmres(op, rhs, None, atol=self.lin_abstol,
                            rtol=self.lin_reltol, restart=kr.gmres_restart, itmax=kr.maxiters,
                            fixed_iters=kr.fixed_iters, ortho=ortho, M=M, Ml=Ml)
            self.stats.gmres_iters += info.iters
            self.last_gmres = info
            if info.failed:
                return None
        else:
            import scipy.sparse.linalg as spla
            if new_jacobian or getattr(self, "_lu", None) is None:
                self.stats.nfactors += 1
                try:  # a singular / non-finite factorisation is LinearSolve's ReturnCode.Failure ⇒ success = false
                    self._lu = spla.splu(sp.csc_matrix(self.J + shift * (sp.identity(self.J.shape[0]) if Mm is None else Mm))
                                         if shift else sp.csc_matrix(self.J))
                except RuntimeError:
                    self._lu = None
                    return None
            x = self._lu.solve(self.fu)
            if not np.all(np.isfinite(x)):
                return None
        return -x  # @bb @. δu *= -1  (newton.jl:138)

    # -- Dogleg.solve! (descent/dogleg.jl:86-151)
    def _dogleg(self, new_jacobian, trust_region):
        du_newton = self._newton_descent(new_jacobian)
        if du_newton is None:
            return None, float("nan")
        if L2_NORM(du_newton) <= trust_region:
            return du_newton.copy(), float("nan")
        du_cauchy = -self._apply_JT(self.fu, self.u)  # steepest.jl:75-77
        l_grad = L2_NORM(du_cauchy)
        Jdc = self._apply_J(du_cauchy, self.u)
        duJJdu = float(np.dot(Jdc, Jdc))
        d_cauchy = (l_grad ** 3) / duJJdu
        if d_cauchy >= trust_region:
            lam = trust_region / l_grad
            return lam * du_cauchy, lam * lam * duJJdu
        c1 = (d_cauchy / l_grad) * du_cauchy
        c2 = du_newton - c1
        a = float(np.dot(c2, c2))
        b = 2.0 * float(np.dot(c1, c2))
        c = d_cauchy ** 2 - trust_region ** 2
        aux = max(0.0, b * b - 4.0 * a * c)
        tau = (-b + math.sqrt(aux)) / (2.0 * a)
        return c1 + tau * c2, float("nan")

    # -- GenericTrustRegionSchemeCache solve! (trust_region.jl:396-514)
    def _tr_solve(self, du, duJJdu):
        m = self.alg.radius_update_scheme
        u_new = self.u + du
        fu_new = self.prob.f(u_new)
        self.stats.nf += 1
        if math.isnan(duJJdu):
            Jdu = self._apply_J(du, self.u)
            duJJdu = float(np.dot(Jdu, Jdu))
        JTfu = self._apply_JT(self.fu, self.u)
        num = (L2_NORM(fu_new) ** 2 - L2_NORM(self.fu) ** 2) / 2.0
        denom = float(np.dot(du, JTfu)) + duJJdu / 2.0
        self.rho = num / denom if denom != 0 else float("nan")
        rho = self.rho
        self.last_step_accepted = bool(rho > self.step_threshold)
        nd = L2_NORM(du)
        if m == SIMPLE:
            if rho < self.shrink_threshold:
                self.trust_region *= self.shrink_factor
                self.shrink_counter += 1
            else:
                self.shrink_counter = 0
                if rho > self.expand_threshold and rho > self.step_threshold:
                    self.trust_region = self.expand_factor * self.trust_region
        elif m == NLSOLVE:
            if rho < self.shrink_threshold:
                self.trust_region *= self.shrink_factor
                self.shrink_counter += 1
            else:
                self.shrink_counter = 0
                if rho >= self.expand_threshold:
                    self.trust_region = self.expand_factor * nd
                elif rho >= self.p1:
                    self.trust_region = max(self.trust_region, self.expand_factor * nd)
        elif m == NOCEDAL_WRIGHT:
            if rho < self.shrink_threshold:
                self.trust_region = self.shrink_factor * nd
                self.shrink_counter += 1
            else:
                self.shrink_counter = 0
                if rho > self.expand_threshold and abs(nd - self.trust_region) < 1e-6 * self.trust_region:
                    self.trust_region = self.expand_factor * self.trust_region
        elif m == HEI:
            r, c2, M, g1, g2, beta = rho, self.shrink_threshold, self.p1, self.p3, self.p4, self.p2
            if r >= c2:
                rf = (2 * (M - 1 - g2) * math.atan(r - c2) + (1 + g2)) / math.pi
            else:
                rf = (1 - g1 - beta) * (math.exp(r - c2) + beta / (1 - g1 - beta))
            tr_new = rf * nd
            if tr_new < self.trust_region:
                self.shrink_counter += 1
            else:
                self.shrink_counter = 0
            self.trust_region = tr_new
        elif m == YUAN:
            if rho < self.shrink_threshold:
                self.p1 = self.p2 * self.p1
                self.shrink_counter += 1
            else:
                if rho >= self.expand_threshold and 2 * nd > self.trust_region:
                    self.p1 = self.p3 * self.p1
                self.shrink_counter = 0
            JTf_new = self.prob.vjp(fu_new, u_new)
            self.trust_region = self.p1 * L2_NORM(JTf_new)
        elif m == FAN:
            if rho < self.shrink_threshold:
                self.p1 *= self.p2
                self.shrink_counter += 1
            else:
                self.shrink_counter = 0
                if rho > self.expand_threshold:
                    self.p1 = min(self.p1 * self.p3, self.p4)
            self.trust_region = self.p1 * (L2_NORM(fu_new) ** 0.99)
        elif m == BASTIN:
            if rho > self.step_threshold:
                Jdu2 = self.prob.jvp(self.tr_du_cache, u_new)
                JTf2 = self.prob.vjp(fu_new, u_new)
                denom_1 = float(np.dot(JTf2, JTf2))
                JTJdu = self.prob.vjp(Jdu2, u_new)
                denom_2 = float(np.dot(JTJdu, JTJdu))
                rho2 = num / (denom_1 + denom_2 / 2.0)
                if rho2 >= self.expand_threshold:
                    self.trust_region = self.p1 * L2_NORM(self.tr_du_cache)
                self.shrink_counter = 0
            else:
                self.trust_region *= self.p2
                self.shrink_counter += 1
        self.trust_region = min(self.trust_region, self.max_trust_radius)
        return self.last_step_accepted, u_new, fu_new

    # -- BackTracking on ϕ(α) = ½‖f(u + α δu)‖², ϕ'(0) = fuᵀ J δu  (LineSearches.jl backtracking.jl restated, [EXT])
    def _backtracking(self, ls, du):
        def phi(a):
            self.stats.nf += 1
            f = self.prob.f(self.u + a * du)
            return 0.5 * float(np.dot(f, f))
        phi0 = 0.5 * float(np.dot(self.fu, self.fu))
        dphi0 = float(np.dot(self.fu, self._apply_J(du, self.u)))
        a1 = a2 = 1.0
        phx0 = phi0
        phx1 = phi(a1)
        itf = 0
        while not math.isfinite(phx1) and itf < 1074:
            itf += 1
            a1 = a2
            a2 = a1 / 2.0
            phx1 = phi(a2)
        it = 0
        while phx1 > phi0 + ls.c_1 * a2 * dphi0:
            it += 1
            if it > ls.maxiters:
                return a2, True
            if ls.order == 2 or it == 1:
                atmp = -(dphi0 * a2 * a2) / (2.0 * (phx1 - phi0 - dphi0 * a2))
            else:
                div = 1.0 / (a1 * a1 * a2 * a2 * (a2 - a1))
                ca = (a1 * a1 * (phx1 - phi0 - dphi0 * a2) - a2 * a2 * (phx0 - phi0 - dphi0 * a1)) * div
                cb = (-a1 ** 3 * (phx1 - phi0 - dphi0 * a2) + a2 ** 3 * (phx0 - phi0 - dphi0 * a1)) * div
                if abs(ca) <= np.finfo(float).eps:
                    atmp = dphi0 / (2.0 * cb)
                else:
                    atmp = (-cb + math.sqrt(max(cb * cb - 3.0 * ca * dphi0, 0.0))) / (3.0 * ca)
            a1 = a2
            atmp = min(atmp, a2 * ls.rho_hi) if atmp == atmp else a2 * ls.rho_hi
            a2 = max(atmp, a2 * ls.rho_lo) if atmp == atmp else a2 * ls.rho_lo
            phx0, phx1 = phx1, phi(a2)
        return a2, False

    # -- LineSearchesJL(; method) [EXT]: Static / StrongWolfe / MoreThuente on ϕ(α) = ½‖f(u + α δu)‖²
    def _lsjl(self, method, du):
        prob, u = self.prob, self.u

        def phi_dphi(a, want_d=True):
            self.stats.nf += 1
            ut = u + a * du
            f = prob.f(ut)
            ph = 0.5 * float(np.dot(f, f))
            if not want_d:
                return ph
            return ph, float(np.dot(f, prob.jvp(du, ut)))

        phi = lambda a: phi_dphi(a, False)      # noqa: E731
        dphi = lambda a: phi_dphi(a)[1]         # noqa: E731
        phi0, dphi0 = phi_dphi(0.0)
        if dphi0 >= 0.0:
            return 1.0, True
        if method == "Static":
            return self._ls_static(phi, 1.0), False
        if method == "StrongWolfe":
            return self._ls_strongwolfe(phi, dphi, phi_dphi, 1.0, phi0, dphi0), False
        if method == "MoreThuente":
            return self._ls_morethuente(phi_dphi, 1.0, phi0, dphi0), False
        if method == "HagerZhang":
            return self._ls_hagerzhang(phi_dphi, 1.0, phi0, dphi0)
        raise ValueError(method)

    # LineSearches.HagerZhang (Hager & Zhang 2005, CG_DESCENT line search: bracket B0–B3, secant² S1–S4, update U0–U3 with
    # bisection θ = ½, (approximate) Wolfe tests T1/T2); δ = 0.1, σ = 0.9, α_max = ∞, ρ = 5, ε = 1e-6, γ = 0.66, at most 50
    # iterations, ψ₃ = 0.1; `mayterminate` is false (the wrapper never sets it). Returns (α, failed): the method's exceptions
    # (non-descent direction, iteration limit, lost bracket) are reported as a failed line search at the best step so far.
    def _ls_hagerzhang(self, phidphi, c, phi_0, dphi_0):
        delta, sigma, rho, eps_hz, gamma, lsmax, psi3 = 0.1, 0.9, 5.0, 1e-6, 0.66, 50, 0.1
        alphamax = float("inf")
        feps = np.finfo(float).eps
        if not (math.isfinite(phi_0) and math.isfinite(dphi_0)) or dphi_0 >= feps * abs(phi_0):
            return 0.0, True
        alphas, values, slopes = [0.0], [phi_0], [dphi_0]
        phi_lim = phi_0 + eps_hz * abs(phi_0)

        def ev(a):
            pa, da = phidphi(a)
            alphas.append(a), values.append(pa), slopes.append(da)
            return pa, da

        def wolfe(cc, pc, dc):
            w1 = delta * dphi_0 >= (pc - phi_0) / cc and dc >= sigma * dphi_0
            w2 = (2.0 * delta - 1.0) * dphi_0 >= dc >= sigma * dphi_0 and pc <= phi_lim
            return w1 or w2

        def bisect(ia, ib):                       # U3 with θ = ½
            a, b = alphas[ia], alphas[ib]
            while b - a > np.spacing(b):
                d = (a + b) / 2.0
                pd, gd = ev(d)
                idd = len(alphas) - 1
                if gd >= 0.0:
                    return ia, idd
                if pd <= phi_lim:
                    a, ia = d, idd
                else:
                    b, ib = d, idd
            return ia, ib

        def update(ia, ib, ic):                   # U0–U3
            a, b, cc = alphas[ia], alphas[ib], alphas[ic]
            if cc < a or cc > b:
                return ia, ib
            if slopes[ic] >= 0.0:
                return ia, ic
            if values[ic] <= phi_lim:
                return ic, ib
            return bisect(ia, ic)

        def secant(a, b, da, db):
            return (a * db - b * da) / (db - da)

        def secant2(ia, ib):                      # S1–S4
            a, b, da, db = alphas[ia], alphas[ib], slopes[ia], slopes[ib]
            if not (da < 0.0 and db >= 0.0):
                raise ArithmeticError("bracket lost")
            cc = secant(a, b, da, db)
            pc, dc = ev(cc)
            ic = len(alphas) - 1
            if wolfe(cc, pc, dc):
                return True, ic, ic
            iA, iB = update(ia, ib, ic)
            a, b = alphas[iA], alphas[iB]
            if iB == ic:
                cc = secant(alphas[ib], alphas[iB], slopes[ib], slopes[iB])
            elif iA == ic:
                cc = secant(alphas[ia], alphas[iA], slopes[ia], slopes[iA])
            if (iA == ic or iB == ic) and a <= cc <= b:
                pc, dc = ev(cc)
                ic = len(alphas) - 1
                if wolfe(cc, pc, dc):
                    return True, ic, ic
                iA, iB = update(iA, iB, ic)
            return False, iA, iB

        if c <= feps:
            return 0.0, False
        phi_c, dphi_c = phidphi(c)
        itf = 1
        while not (math.isfinite(phi_c) and math.isfinite(dphi_c)) and itf < 53:
            itf += 1
            c *= psi3
            phi_c, dphi_c = phidphi(c)
        if not (math.isfinite(phi_c) and math.isfinite(dphi_c)):
            return 0.0, False
        alphas.append(c), values.append(phi_c), slopes.append(dphi_c)
        bracketed, ia, ib, it = False, 0, 1, 1
        try:
            while not bracketed and it < lsmax:   # B0–B3
                if dphi_c >= 0.0:
                    ib = len(alphas) - 1
                    for i in range(ib - 1, -1, -1):
                        if values[i] <= phi_lim:
                            ia = i
                            break
                    bracketed = True
                elif values[-1] > phi_lim:
                    ib, ia = len(alphas) - 1, 0
                    ia, ib = bisect(ia, ib)
                    bracketed = True
                else:
                    cold, phi_cold = c, phi_c
                    if np.nextafter(cold, np.inf) >= alphamax:
                        return cold, False
                    c = min(c * rho, alphamax)
                    phi_c, dphi_c = phidphi(c)
                    itf = 1
                    while not (math.isfinite(phi_c) and math.isfinite(dphi_c)) and c > np.nextafter(cold, np.inf) and itf < 53:
                        alphamax = c
                        itf += 1
                        c = (cold + c) / 2.0
                        phi_c, dphi_c = phidphi(c)
                    if not (math.isfinite(phi_c) and math.isfinite(dphi_c)):
                        return cold, False
                    if dphi_c < 0.0 and c == alphamax:
                        return c, False
                    alphas.append(c), values.append(phi_c), slopes.append(dphi_c)
                it += 1
            while it < lsmax:                     # L1–L3
                a, b = alphas[ia], alphas[ib]
                if b - a <= np.spacing(b):
                    return a, False
                isw, iA, iB = secant2(ia, ib)
                if isw:
                    return alphas[iA], False
                A, B = alphas[iA], alphas[iB]
                if B - A < gamma * (b - a):
                    if np.nextafter(values[ia], np.inf) >= values[ib] and np.nextafter(values[iA], np.inf) >= values[iB]:
                        return A, False
                    ia, ib = iA, iB
                else:
                    cc = (A + B) / 2.0
                    ev(cc)
                    ia, ib = update(iA, iB, len(alphas) - 1)
                it += 1
        except ArithmeticError:
            return alphas[ia], True
        return alphas[ia], True                   # iteration limit: LineSearchException in LineSearches.jl

    @staticmethod
    def _ls_static(phi, a):           # LineSearches.jl static.jl: the proposed step, halved while ϕ is not finite
        pa = phi(a)
        it = 0
        while not math.isfinite(pa) and it < 52:   # -log2(eps(Float64))
            it += 1
            a = a / 2.0
            pa = phi(a)
        return a

    @staticmethod
    def _ls_sw_interp(a1, a2, p1, p2, d1, d2):   # cubic interpolation (Nocedal & Wright eq. 3.59)
        q1 = d1 + d2 - 3.0 * (p1 - p2) / (a1 - a2)
        rad = q1 * q1 - d1 * d2
        q2 = math.sqrt(rad) if rad >= 0.0 else float("nan")
        return a2 - (a2 - a1) * ((d2 + q2 - q1) / (d2 - d1 + 2.0 * q2))

    def _ls_strongwolfe(self, phi, dphi, phidphi, a0, phi0, dphi0, c1=1e-4, c2=0.9, rho=2.0):
        def zoom(alo, ahi):           # alg. 3.6
            aj = float("nan")
            for _ in range(10):
                plo, dlo = phidphi(alo)
                phi_, dhi = phidphi(ahi)
                aj = self._ls_sw_interp(alo, ahi, plo, phi_, dlo, dhi) if alo < ahi else \
                    self._ls_sw_interp(ahi, alo, phi_, plo, dhi, dlo)
                pj = phi(aj)
                if pj > phi0 + c1 * aj * dphi0 or pj > plo:
                    ahi = aj
                else:
                    dj = dphi(aj)
                    if abs(dj) <= -c2 * dphi0:
                        return aj
                    if dj * (ahi - alo) >= 0.0:
                        ahi = alo
                    alo = aj
            return aj
        a_prev, a_i, a_max = 0.0, a0, 65536.0
        p_prev = phi0
        i = 1
        while a_i < a_max:            # alg. 3.5
            p_i = phi(a_i)
            if p_i > phi0 + c1 * a_i * dphi0 or (p_i >= p_prev and i > 1):
                a = zoom(a_prev, a_i)
                phi(a)                # the method returns (α*, ϕ(α*)): one more evaluation
                return a
            d_i = dphi(a_i)
            if abs(d_i) <= -c2 * dphi0:
                return a_i
            if d_i >= 0.0:
                a = zoom(a_i, a_prev)
                phi(a)
                return a
            a_prev, p_prev = a_i, p_i
            a_i *= rho
            i += 1
        phi(a_max)
        return a_max

    @staticmethod
    def _ls_cstep(stx, fx, dgx, sty, fy, dgy, alpha, f, dg, bracketed, amin, amax):
        """MINPACK cstep (Moré & Thuente): safeguarded cubic / quadratic step and the update of the interval of uncertainty."""
        info = 0
        if (bracketed and (alpha <= min(stx, sty) or alpha >= max(stx, sty))) or dgx * (alpha - stx) >= 0.0 or amax < amin:
            return stx, fx, dgx, sty, fy, dgy, alpha, f, dg, bracketed, 0
        sgnd = dg * (dgx / abs(dgx))
        if f > fx:                       # case 1: higher function value — bracketed
            info, bound = 1, True
            theta = 3.0 * (fx - f) / (alpha - stx) + dgx + dg
            sc = max(abs(theta), abs(dgx), abs(dg))
            gamma = sc * math.sqrt((theta / sc) ** 2 - (dgx / sc) * (dg / sc))
            if alpha < stx:
                gamma = -gamma
            pp = gamma - dgx + theta
            q = gamma - dgx + gamma + dg
            r = pp / q
            ac = stx + r * (alpha - stx)
            aq = stx + ((dgx / ((fx - f) / (alpha - stx) + dgx)) / 2.0) * (alpha - stx)
            af = ac if abs(ac - stx) < abs(aq - stx) else (ac + aq) / 2.0
            bracketed = True
        elif sgnd < 0.0:                 # case 2: lower value, derivatives of opposite sign — bracketed
            info, bound = 2, False
            theta = 3.0 * (fx - f) / (alpha - stx) + dgx + dg
            sc = max(abs(theta), abs(dgx), abs(dg))
            gamma = sc * math.sqrt((theta / sc) ** 2 - (dgx / sc) * (dg / sc))
            if alpha > stx:
                gamma = -gamma
            pp = gamma - dg + theta
            q = gamma - dg + gamma + dgx
            r = pp / q
            ac = alpha + r * (stx - alpha)
            aq = alpha + (dg / (dg - dgx)) * (stx - alpha)
            af = ac if abs(ac - alpha) > abs(aq - alpha) else aq
            bracketed = True
        elif abs(dg) < abs(dgx):         # case 3: lower value, same sign, derivative magnitude decreases
            info, bound = 3, True
            theta = 3.0 * (fx - f) / (alpha - stx) + dgx + dg
            sc = max(abs(theta), abs(dgx), abs(dg))
            gamma = sc * math.sqrt(max(0.0, (theta / sc) ** 2 - (dgx / sc) * (dg / sc)))
            if alpha > stx:
                gamma = -gamma
            pp = gamma - dg + theta
            q = gamma + dgx - dg + gamma
            r = pp / q
            if r < 0.0 and gamma != 0.0:
                ac = alpha + r * (stx - alpha)
            elif alpha > stx:
                ac = amax
            else:
                ac = amin
            aq = alpha + (dg / (dg - dgx)) * (stx - alpha)
            if bracketed:
                af = ac if abs(alpha - ac) < abs(alpha - aq) else aq
            else:
                af = ac if abs(alpha - ac) > abs(alpha - aq) else aq
        else:                            # case 4: lower value, same sign, derivative magnitude does not decrease
            info, bound = 4, False
            if bracketed:
                theta = 3.0 * (f - fy) / (sty - alpha) + dgy + dg
                sc = max(abs(theta), abs(dgy), abs(dg))
                gamma = sc * math.sqrt((theta / sc) ** 2 - (dgy / sc) * (dg / sc))
                if alpha > sty:
                    gamma = -gamma
                pp = gamma - dg + theta
                q = gamma - dg + gamma + dgy
                r = pp / q
                af = alpha + r * (sty - alpha)
            elif alpha > stx:
                af = amax
            else:
                af = amin
        if f > fx:
            sty, fy, dgy = alpha, f, dg
        else:
            if sgnd < 0.0:
                sty, fy, dgy = stx, fx, dgx
            stx, fx, dgx = alpha, f, dg
        af = max(amin, min(amax, af))
        alpha = af
        if bracketed and bound:
            if sty > stx:
                alpha = min(stx + (2.0 / 3.0) * (sty - stx), alpha)
            else:
                alpha = max(stx + (2.0 / 3.0) * (sty - stx), alpha)
        return stx, fx, dgx, sty, fy, dgy, alpha, f, dg, bracketed, info

    def _ls_morethuente(self, phidphi, alpha, phi0, dphi0, f_tol=1e-4, gtol=0.9, x_tol=1e-8, amin=1e-16, amax=65536.0,
                        maxfev=100):
        info, info_cstep = 0, 1
        bracketed, stage1, nfev = False, True, 0
        finit, dgtest = phi0, f_tol * dphi0
        width = amax - amin
        width1 = 2.0 * width
        stx, fx, dgx = 0.0, finit, dphi0
        sty, fy, dgy = 0.0, finit, dphi0
        stmin, stmax = 0.0, alpha + 4.0 * (alpha - stx)
        alpha = min(max(alpha, amin), amax)
        f, dg = phidphi(alpha)
        nfev += 1
        itf = 0
        while (not math.isfinite(f) or not math.isfinite(dg)) and itf < 52:
            itf += 1
            alpha = alpha / 2.0
            f, dg = phidphi(alpha)
            nfev += 1
            stx = 0.875 * alpha
        while True:
            if bracketed:
                stmin, stmax = min(stx, sty), max(stx, sty)
            else:
                stmin, stmax = stx, alpha + 4.0 * (alpha - stx)
            stmin, stmax = max(amin, stmin), min(amax, stmax)
            alpha = min(max(alpha, amin), amax)
            if (bracketed and (alpha <= stmin or alpha >= stmax)) or nfev >= maxfev - 1 or info_cstep == 0 or \
                    (bracketed and stmax - stmin <= x_tol * stmax):
                alpha = stx
            f, dg = phidphi(alpha)      # (the first pass evaluates the initial step a second time, as LineSearches.jl does)
            nfev += 1
            ftest1 = finit + alpha * dgtest
            if (bracketed and (alpha <= stmin or alpha >= stmax)) or info_cstep == 0:
                info = 6
            if alpha == amax and f <= ftest1 and dg <= dgtest:
                info = 5
            if alpha == amin and (f > ftest1 or dg >= dgtest):
                info = 4
            if nfev >= maxfev:
                info = 3
            if bracketed and stmax - stmin <= x_tol * stmax:
                info = 2
            if f <= ftest1 and abs(dg) <= -gtol * dphi0:
                info = 1
            if info != 0:
                break
            if stage1 and f <= ftest1 and dg >= min(f_tol, gtol) * dphi0:
                stage1 = False
            if stage1 and f <= fx and f > ftest1:
                fm, fxm, fym = f - alpha * dgtest, fx - stx * dgtest, fy - sty * dgtest
                dgm, dgxm, dgym = dg - dgtest, dgx - dgtest, dgy - dgtest
                stx, fxm, dgxm, sty, fym, dgym, alpha, fm, dgm, bracketed, info_cstep = self._ls_cstep(
                    stx, fxm, dgxm, sty, fym, dgym, alpha, fm, dgm, bracketed, stmin, stmax)
                fx, fy = fxm + stx * dgtest, fym + sty * dgtest
                dgx, dgy = dgxm + dgtest, dgym + dgtest
            else:
                stx, fx, dgx, sty, fy, dgy, alpha, f, dg, bracketed, info_cstep = self._ls_cstep(
                    stx, fx, dgx, sty, fy, dgy, alpha, f, dg, bracketed, stmin, stmax)
            if bracketed:
                if abs(sty - stx) >= (2.0 / 3.0) * width1:
                    alpha = stx + (sty - stx) / 2.0
                width1 = width
                width = abs(sty - stx)
        return alpha

    # -- pre/post_step_forcing! (eisenstat_walker.jl:42-89)
    def _pre_step_forcing(self, it):
        p = self.forcing
        if it == 0:
            self.ew_eta = p.eta0
            self.ew_rnorm = self.ew_rnorm_prev = L2_NORM(self.fu)
        else:
            eta_prev = self.ew_eta
            self.ew_eta = p.gamma * (self.ew_rnorm / self.ew_rnorm_prev) ** p.alpha
            if p.safeguard:
                eta_sg = p.gamma * eta_prev ** p.alpha
                if eta_sg > p.safeguard_threshold and eta_sg > self.ew_eta:
                    self.ew_eta = eta_sg
            self.ew_eta = min(max(self.ew_eta, 0.0), p.eta_max)
        self.lin_reltol = self.ew_eta  # LinearSolve.update_tolerances!(…; reltol = η)

    def _post_step_forcing(self):
        self.ew_rnorm_prev = self.ew_rnorm
        self.ew_rnorm = L2_NORM(self.fu)

    # -- InternalAPI.step! (FirstOrder/src/solve.jl:325-465) + CommonSolve.step! (Base/src/solve.jl:835-859)
    def step(self, recompute_jacobian=None, evaluate_residual=True):
        if self.force_stop or self.nsteps >= self.maxiters:
            return
        self._internal_step(recompute_jacobian, evaluate_residual)
        self.stats.nsteps += 1
        self.nsteps += 1

    # supports_deferred_residual (FirstOrder/src/solve.jl:303-316; residual_only_termination_mode,
    # termination_conditions.jl:43-45): unglobalised step, AbsTerminationMode / AbsNormTerminationMode, no trace
    def supports_deferred_residual(self):
        if self.is_tr or self.is_lm or getattr(self.alg, "linesearch", None) is not None:
            return False
        if self.tc.mode not in (TM_ABS, TM_ABSNORM):
            return False
        return not self.store_trace

    # refresh_residual! (FirstOrder/src/solve.jl:318-324)
    def refresh_residual(self):
        if not getattr(self, "fu_deferred", False):
            return None
        self.fu_deferred = False
        self.fu = self.prob.f(self.u)
        self.stats.nf += 1
        if self.tc(self.fu, self.u, self.u_cache):   # check_and_update!
            self.retcode = self.tc.retcode
            self._rollback_to_best()
            self.force_stop = True
        return None

    def _internal_step(self, recompute_jacobian=None, evaluate_residual=True):
        self.refresh_residual()                      # solve.jl:333
        defer_residual = (not evaluate_residual) and self.supports_deferred_residual()   # :336-337
        if (recompute_jacobian is None or recompute_jacobian) and self.make_new_jacobian:
            if self.concrete:
                self.J = self.prob.jac(self.u)
                self.stats.njacs += 1
            new_jacobian = True
        else:
            new_jacobian = False
        if self.forcing is not None:
            self._pre_step_forcing(self.nsteps)
        if self.is_lm:
            return self._lm_step(new_jacobian, recompute_jacobian, evaluate_residual)
        if self.is_tr:
            du, duJJdu = self._dogleg(new_jacobian, self.trust_region)
        else:
            du, duJJdu = self._newton_descent(new_jacobian), float("nan")
        if du is None:  # linear solve failed
            if new_jacobian:
                self.retcode = LINSOLVE_FAILED
                self.force_stop = True
                return
            self.make_new_jacobian = True
            return self._internal_step(True, evaluate_residual)
        self.du = du
        if self.forcing is not None:
            self._post_step_forcing()
        self.make_new_jacobian = True
        accepted = True
        if self.is_tr:
            self.tr_du_cache = du
            accepted, u_new, fu_new = self._tr_solve(du, duJJdu)
            if accepted:
                self.u = u_new.copy()
                self.fu = fu_new.copy()
            else:
                self.make_new_jacobian = False
            if self.shrink_counter > self.alg.max_shrink_times:
                self.retcode = SHRINK_EXCEEDED
                self.force_stop = True
        else:
            ls = getattr(self.alg, "linesearch", None)
            if ls is not None:  # Val(:LineSearch), solve.jl:392-408
                if isinstance(ls, LineSearchesJL) and ls.method != "BackTracking":
                    alpha, ls_failed = self._lsjl(ls.method, du)
                else:
                    alpha, ls_failed = self._backtracking(ls if isinstance(ls, BackTracking) else BackTracking(), du)
                if ls_failed:
                    self.retcode = LINESEARCH_FAILED
                    self.force_stop = True
                du = alpha * du
                self.du = du
            self.u = self.u + du         # @bb axpy!(α, δu, cache.u)
            if not defer_residual:
                self.fu = self.prob.f(self.u)  # Utils.evaluate_f!
                self.stats.nf += 1
        # check_and_update! (termination_conditions.jl:414-426) — or the deferral (solve.jl:448-452)
        if defer_residual:
            self.fu_deferred = True
        elif self.tc(self.fu, self.u, self.u_cache):
            self.retcode = self.tc.retcode
            self._rollback_to_best()
            self.force_stop = True
        if self.store_trace:
            self.trace.append(dict(iter=self.nsteps + 1, fnorm_inf=Linf_NORM(self.fu), step_norm2=L2_NORM(du),
                                   eta=self.lin_reltol if self.krylov is not None else float("nan"),
                                   gmres_iters=self.last_gmres.iters if self.krylov is not None else 0,
                                   accepted=bool(accepted),
                                   trust_region=self.trust_region if self.is_tr else float("nan"),
                                   rho=self.rho if self.is_tr else float("nan")))
        self.u_cache = self.u.copy()

    # the rest of step! (FirstOrder/src/solve.jl:365-462) for LevenbergMarquardt: Val(:TrustRegion) globalisation with a
    # cache that has neither `trust_region` nor `shrink_counter`, and a descent that can report success = false
    def _lm_step(self, new_jacobian, recompute_jacobian, evaluate_residual):
        du, success, v = self._lm_descent()
        if du is None:  # DampedNewtonDescent alone: linsolve_success = false
            if new_jacobian:
                self.retcode = LINSOLVE_FAILED
                self.force_stop = True
                return
            self.make_new_jacobian = True
            return self._internal_step(True, evaluate_residual)
        accepted = False
        if success:
            self.du = du
            self.make_new_jacobian = True
            accepted, u_new, fu_new = self._lm_tr_solve(du, v)
            if accepted:
                self.u = u_new.copy()
                self.fu = fu_new.copy()
            else:
                self.make_new_jacobian = False
            if self.tc(self.fu, self.u, self.u_cache):
                self.retcode = self.tc.retcode
                self._rollback_to_best()
                self.force_stop = True
        else:
            self.make_new_jacobian = False
        if self.store_trace:
            self.trace.append(dict(iter=self.nsteps + 1, fnorm_inf=Linf_NORM(self.fu), step_norm2=L2_NORM(self.du),
                                   eta=self.lin_reltol if self.krylov is not None else float("nan"),
                                   gmres_iters=self.last_gmres.iters if self.krylov is not None else 0,
                                   accepted=bool(accepted), trust_region=self.lm_lam, rho=self.lm_beta))
        self.u_cache = self.u.copy()
        self._lm_callback()

    def _rollback_to_best(self):  # update_from_termination_cache! (termination_conditions.jl:440-453)
        if self.tc.u is None or np.array_equal(self.u, self.tc.u):
            return
        self.u = self.tc.u.copy()
        self.fu = self.prob.f(self.u)
        self.stats.nf += 1

    # -- _run_cache_to_completion! (Base/src/solve.jl:360-387)
    def solve(self):
        while (not self.force_stop) and self.nsteps < self.maxiters:
            self.step()
        if self.retcode == DEFAULT:
            self.retcode = MAXITERS if self.nsteps >= self.maxiters else SUCCESS
        self.refresh_residual()   # Base/src/solve.jl:376-378
        self._rollback_to_best()
        return Solution(self.u.copy(), self.fu.copy(), self.retcode, self.stats, self.trace)

    # -- reinit! (FirstOrder/src/solve.jl:108-133, trust_region.jl:292-317, eisenstat_walker.jl:104-107)
    def reinit(self, u0=None, p=None):
        if p is not None:
            self.prob.p = p
        u0 = self.u if u0 is None else np.asarray(u0, dtype=np.float64)
        self.u = np.array(u0, copy=True)
        self.fu = self.prob.f(self.u)
        self.u_cache = self.u.copy()
        self.stats = Stats()
        self.nsteps = 0
        self.force_stop = False
        self.retcode = DEFAULT
        self.make_new_jacobian = True
        self.trace = []
        self.tc.reinit(self.fu, self.u)
        if self.forcing is not None:
            self.ew_eta = self.forcing.eta0
        self.lin_reltol = self.reltol
        if self.is_tr:
            self.max_trust_radius, self.initial_trust_radius = self._tr_radii(self.u, self.fu)
            if self.alg.radius_update_scheme == YUAN:
                self.initial_trust_radius = self.p1 * L2_NORM(self._apply_JT(self.fu, self.u))
            self.last_step_accepted = False
            self.trust_region = self.initial_trust_radius
            self.shrink_counter = 0
        if self.is_lm:
            self._lm_init(self.u)
        if self.is_pt:   # reinit!: α⁻¹ back to its initial value, the residual norm re-seeded (needs_reset)
            self.pt_ainv = 1.0 / float(self.alg.alpha_initial)
            self.pt_res = L2_NORM(self.fu)


def init(prob, alg, **kw):
    return FirstOrderCache(prob, alg, **kw)


def simple_newton_raphson(f, jac, u0, p, abstol=None, maxiters=1000):
    """SimpleNewtonRaphson for one small system — lib/SimpleNonlinearSolve/src/raphson.jl:39-83 restated:
    iszero(fx) short cut; per iteration δx = J \\ fx, x −= δx, THEN the AbsNorm(maximum∘abs) test on the residual of the
    previous iterate (check_termination precedes evaluate_f!!, utils.jl:64-66), then f and J at the new x.
    Returns (x, fx, retcode, iterations)."""
    if abstol is None:
        abstol = float(np.finfo(float).eps) ** 0.8
    x = np.array(u0, dtype=float)
    fx = np.asarray(f(x, p), dtype=float)
    if not np.any(fx):
        return x, fx, SUCCESS, 0
    J = np.asarray(jac(x, p), dtype=float)
    for it in range(1, maxiters + 1):
        with np.errstate(all="ignore"):
            try:
                dx = np.linalg.solve(J, fx)
            except np.linalg.LinAlgError:        # singular J: StaticArrays' LU yields Inf/NaN, the loop runs on
                dx = np.full_like(x, np.nan)
            x = x - dx
            nrm = np.max(np.abs(fx)) if not np.any(np.isnan(fx)) else np.nan
            if nrm <= abstol:
                return x, fx, SUCCESS, it
            fx = np.asarray(f(x, p), dtype=float)
            J = np.asarray(jac(x, p), dtype=float)
    return x, fx, MAXITERS, maxiters


def simple_trust_region(f, jac, u0, p, abstol=None, maxiters=1000, step_threshold=1e-4, shrink_threshold=0.25,
                        expand_threshold=0.75, shrink_factor=0.25, expand_factor=2.0, max_shrink_times=32):
    """SimpleTrustRegion for one small system — lib/SimpleNonlinearSolve/src/trust_region.jl:57-229 restated (default
    radius-update rule), including its quirks: δsd = −g (no Cauchy step length), and after a rejected trial `fx` keeps the
    trial point's residual while J, g, f_k stay at the accepted point. Returns (x, fx, retcode, iterations)."""
    if abstol is None:
        abstol = float(np.finfo(float).eps) ** 0.8
    x = np.array(u0, dtype=float)
    xo = x.copy()
    fx = np.asarray(f(x, p), dtype=float)
    norm_fx = float(np.linalg.norm(fx))
    J = np.asarray(jac(x, p), dtype=float)
    dmax = max(norm_fx, float(np.max(x) - np.min(x)))
    delta = dmax / 11.0
    fk = 0.5 * norm_fx ** 2
    g = J.T @ fx
    shrink = 0

    def done(fv):
        return (not np.any(np.isnan(fv))) and np.max(np.abs(fv)) <= abstol

    if done(fx):
        return x, fx, SUCCESS, 0
    with np.errstate(all="ignore"):
        for it in range(1, maxiters + 1):
            try:
                dN = -np.linalg.solve(J, fx)
            except np.linalg.LinAlgError:
                dN = np.full_like(x, np.nan)
            if np.linalg.norm(dN) <= delta:
                dl = dN
            else:
                dsd = -g
                nsd = np.linalg.norm(dsd)
                if nsd >= delta:
                    dl = dsd * (delta / nsd)
                else:
                    q = dN - dsd
                    dNN, dSN, dSS = float(q @ q), float(dsd @ q), float(dsd @ dsd)
                    tau = (-dSN + math.sqrt(dSN * dSN - dNN * (dSS - delta * delta))) / dNN
                    dl = dsd + tau * q
            x = xo + dl
            fx = np.asarray(f(x, p), dtype=float)
            fk1 = float(np.linalg.norm(fx)) ** 2 / 2.0
            r = (fk1 - fk) / (float(dl @ g) + float(dl @ (J.T @ (J @ dl))) / 2.0)
            if r >= shrink_threshold:
                shrink = 0
            else:
                delta = shrink_factor * delta
                shrink += 1
                if shrink > max_shrink_times:
                    return x, fx, SHRINK_EXCEEDED, it
            if r >= step_threshold:
                if done(fx):
                    return x, fx, SUCCESS, it
                xo = x.copy()
                J = np.asarray(jac(x, p), dtype=float)
                if r > expand_threshold:
                    delta = min(expand_factor * delta, dmax)
                fk = fk1
                g = J.T @ fx
    return x, fx, MAXITERS, maxiters


# ----------------------------------------------------------------------------- polyalgorithms
@dataclass
class NonlinearSolvePolyAlgorithm:   # lib/NonlinearSolveBase/src/polyalg.jl:62-73: tried in order until one succeeds
    algs: tuple
    start_index: int = 1             # 1-based, as in the reference

    def __post_init__(self):
        self.algs = tuple(self.algs)
        assert 0 < self.start_index <= len(self.algs)


def RobustMultiNewton(concrete_jac=None, linsolve=None):   # lib/NonlinearSolveFirstOrder/src/poly_algs.jl:21-45
    kw = dict(concrete_jac=concrete_jac, linsolve=linsolve)
    return NonlinearSolvePolyAlgorithm((
        TrustRegion(**kw), TrustRegion(radius_update_scheme=BASTIN, **kw), NewtonRaphson(**kw),
        NewtonRaphson(linesearch=BackTracking(), **kw), TrustRegion(radius_update_scheme=NLSOLVE, **kw),
        TrustRegion(radius_update_scheme=FAN, **kw)))


def FastShortcutNLLSPolyalg(concrete_jac=None, linsolve=None):   # poly_algs.jl:62-88 (LevenbergMarquardt keeps its own concrete_jac)
    return NonlinearSolvePolyAlgorithm((
        GaussNewton(linsolve=linsolve, concrete_jac=concrete_jac), LevenbergMarquardt(linsolve=linsolve, disable_geodesic=True),
        TrustRegion(linsolve=linsolve, concrete_jac=concrete_jac),
        GaussNewton(linsolve=linsolve, linesearch=BackTracking(), concrete_jac=concrete_jac),
        TrustRegion(linsolve=linsolve, radius_update_scheme=FAN, concrete_jac=concrete_jac), LevenbergMarquardt(linsolve=linsolve)))


RETAIN_REPROBE_INTERVAL = 8   # polyalg.jl:186


def findmin_resids(resids, least_squares=False):
    """polyalg.jl:412-430: index of the smallest ‖fu‖ (∞-norm; 2-norm for least squares); NaN counts as Inf; `None` entries
    (sub-algorithms never attempted) are skipped; the earliest of equal minima wins."""
    nrm = (lambda r: float(np.linalg.norm(r, 2))) if least_squares else (lambda r: float(np.max(np.abs(r))) if r.size else 0.0)
    idx = next(i for i, r in enumerate(resids) if r is not None)
    f0 = nrm(resids[idx])
    best, bi = np.inf, -1
    for j in range(idx + 1, len(resids)):
        fx = np.inf if resids[j] is None else nrm(resids[j])
        fx = np.inf if np.isnan(fx) else fx
        if fx < best:
            best, bi = fx, j
    return (best, bi) if (bi >= 0 and best < f0) else (f0, idx)


def _sum_stats(parts):
    out = Stats()
    for st in parts:
        for k in ("nf", "njacs", "nfactors", "nsolve", "nsteps", "gmres_iters"):
            setattr(out, k, getattr(out, k) + getattr(st, k))
    return out


class PolyAlgorithmCache:
    """NonlinearSolvePolyAlgorithmCache (polyalg.jl:79-121) — `init` builds every sub-cache (:266-311), `solve!` runs them in
    order from `current` (Base/src/solve.jl:465-614), `reinit!` with best-sub-algorithm retention, wrap-around, periodic
    re-probe and lazy sub-cache reinitialisation (polyalg.jl:188-244). The reference hands ONE NLStats object to all
    sub-caches, and every sub-cache `reinit!` zeroes it; restated as: the statistics are summed over the sub-caches that ran
    since the last reinitialisation of any of them."""

    def __init__(self, prob, alg, least_squares=False, **kw):
        self.prob, self.alg, self.least_squares = prob, alg, least_squares
        self.caches = [FirstOrderCache(prob, a, **kw) for a in alg.algs]
        self.N = len(self.caches)
        self.best, self.current = -1, alg.start_index
        self.retain_best, self.start_current, self.wrapped, self.retain_count = False, alg.start_index, False, 0
        self.deferred = (None, None)
        self.retcode, self.force_stop, self.nsteps = DEFAULT, False, 0
        self._ran = []         # sub-caches whose statistics the shared NLStats currently holds
        self.u0 = np.array(prob.u0() if kw.get("u0") is None else kw["u0"], dtype=np.float64)

    @property
    def stats(self):
        return _sum_stats(c.stats for c in self._ran)

    def _note(self, c):
        if all(c is not r for r in self._ran):
            self._ran.append(c)

    def _deferred_reinit(self, i):   # polyalg.jl:246-253
        self.caches[i - 1].reinit(self.deferred[0], self.deferred[1])
        self._ran = []               # the sub-cache reinit! zeroes the shared statistics

    def reinit(self, u0=None, p=None, retain_best=False):
        if u0 is None:
            u0 = self.u0
        self.u0 = np.array(u0, dtype=np.float64)
        self.retain_best = retain_best
        self.retain_count = self.retain_count + 1 if retain_best else 0
        retained = retain_best and 1 <= self.best <= self.N
        reprobe = retained and self.best > self.alg.start_index and self.retain_count % RETAIN_REPROBE_INTERVAL == 0
        self.current = self.best if (retained and not reprobe) else self.alg.start_index
        self.start_current, self.wrapped = self.current, False
        if retain_best:
            self.deferred = (self.u0.copy(), p)
            self.caches[self.current - 1].reinit(self.u0, p)
        else:
            for c in self.caches:
                c.reinit(self.u0, p)
        self._ran = []
        self.nsteps, self.force_stop, self.retcode = 0, False, DEFAULT
        return self

    def _attempt(self, i):
        c = self.caches[i - 1]
        if self.retain_best and i != self.start_current:
            self._deferred_reinit(i)
        self._note(c)
        sol = c.solve()
        if sol.retcode == SUCCESS:
            self.best = i
            self.retcode = sol.retcode
            return Solution(sol.u, c.fu.copy(), sol.retcode, self.stats, sol.trace)
        self.current = i + 1
        return None

    def solve(self):
        attempted = [False] * self.N
        for i in range(1, self.N + 1):
            if i == self.current:
                attempted[i - 1] = True
                out = self._attempt(i)
                if out is not None:
                    return out
        if self.retain_best and not self.wrapped and self.start_current > self.alg.start_index:
            self.wrapped, self.current = True, self.alg.start_index
        for i in range(1, self.N):
            if self.wrapped and i == self.current and i < self.start_current:
                attempted[i - 1] = True
                out = self._attempt(i)
                if out is not None:
                    return out
        # every rung failed: the lowest residual among the sub-caches wins (:576-612) — `@isdefined(cache_i)` is true for
        # every i (assigned unconditionally in the first pass), so un-attempted sub-caches compete with their init residual
        fus = [c.fu for c in self.caches]
        _m, idx = findmin_resids(fus, self.least_squares)
        c = self.caches[idx]
        self.retcode = c.retcode
        return Solution(c.u.copy(), c.fu.copy(), c.retcode, self.stats, c.trace)

    # -- InternalAPI.step! (polyalg.jl:313-371): one step of the current sub-cache; on its termination record the success or
    # move up the ladder (wrapping once under retention); past the last rung pick the lowest residual
    def step(self):
        if not (1 <= self.current <= self.N):
            if self.retain_best and not self.wrapped and self.start_current > self.alg.start_index:
                self.wrapped, self.current = True, self.alg.start_index
                self._deferred_reinit(self.current)
                return
            _m, idx = findmin_resids([c.fu for c in self.caches], self.least_squares)
            self.best, self.retcode, self.force_stop = idx + 1, self.caches[idx].retcode, True
            return
        i = self.current
        c = self.caches[i - 1]
        self._note(c)
        c.step()
        self.nsteps += 1
        if c.force_stop or c.nsteps >= c.maxiters:   # !not_terminated(sub-cache)
            rc = c.retcode if c.retcode != DEFAULT else (MAXITERS if c.nsteps >= c.maxiters else SUCCESS)
            if rc == SUCCESS:
                self.best, self.force_stop, self.retcode = i, True, rc
            elif self.wrapped and i + 1 >= self.start_current:
                _m, idx = findmin_resids([cc.fu for cc in self.caches], self.least_squares)
                self.best, self.retcode, self.force_stop = idx + 1, self.caches[idx].retcode, True
            else:
                self.current = i + 1
                if i != self.N and self.retain_best:
                    self._deferred_reinit(i + 1)

    @property
    def u(self):
        return self.caches[min(max(self.current, 1), self.N) - 1].u

    @property
    def fu(self):
        return self.caches[min(max(self.current, 1), self.N) - 1].fu


def polysolve(prob, alg, least_squares=False, **kw):
    """`__generated_polysolve` (Base/src/solve.jl:657-790): the one-shot path — each sub-algorithm gets a fresh `__solve` (its
    cache is built only when the ladder reaches it), all sharing one NLStats; first success wins, else the lowest residual
    among the sub-algorithms that ran."""
    sols, ran = [None] * len(alg.algs), []
    for i in range(alg.start_index, len(alg.algs) + 1):
        c = FirstOrderCache(prob, alg.algs[i - 1], **kw)
        sol = c.solve()
        ran.append(c)
        sols[i - 1] = sol
        if sol.retcode == SUCCESS:
            return Solution(sol.u, sol.resid, sol.retcode, _sum_stats(x.stats for x in ran), sol.trace)
    _m, idx = findmin_resids([None if s_ is None else s_.resid for s_ in sols], least_squares)
    sol = sols[idx]
    return Solution(sol.u, sol.resid, sol.retcode, _sum_stats(x.stats for x in ran), sol.trace)


def init(prob, alg, **kw):  # noqa: F811
    if isinstance(alg, NonlinearSolvePolyAlgorithm):
        return PolyAlgorithmCache(prob, alg, **kw)
    return FirstOrderCache(prob, alg, **kw)


def solve(prob, alg, **kw):
    if isinstance(alg, NonlinearSolvePolyAlgorithm):
        return polysolve(prob, alg, **kw)
    return FirstOrderCache(prob, alg, **kw).solve()


# ----------------------------------------------------------------------------- Jacobian operators (L2)
class JacobianOperator:
    """JacobianOperator / StatefulJacobianOperator / normal form — SciMLJacobianOperators.jl:86-291."""

    def __init__(self, prob: Problem, mode="jvp"):
        self.prob, self.mode = prob, mode

    @property
    def T(self):
        return JacobianOperator(self.prob, "vjp" if self.mode == "jvp" else "jvp")

    def __call__(self, v, u):
        return self.prob.jvp(v, u) if self.mode == "jvp" else self.prob.vjp(v, u)


class StatefulJacobianOperator:
    def __init__(self, op: JacobianOperator, u):
        self.op, self.u = op, np.asarray(u, dtype=np.float64)

    @property
    def T(self):
        return StatefulJacobianOperator(self.op.T, self.u)

    def __matmul__(self, v):
        if isinstance(v, StatefulJacobianOperator):  # Jᵀ * J → normal form
            left, right = self, v
            return _NormalForm(left, right)
        return self.op(np.asarray(v, dtype=np.float64), self.u)


class _NormalForm:
    def __init__(self, vjp_op, jvp_op):
        self.vjp_op, self.jvp_op = vjp_op, jvp_op

    def __matmul__(self, x):
        return self.vjp_op @ (self.jvp_op @ x)
