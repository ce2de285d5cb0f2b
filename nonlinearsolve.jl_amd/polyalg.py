"""Polyalgorithms over the device caches: NonlinearSolvePolyAlgorithm (lib/NonlinearSolveBase/src/polyalg.jl:62-121,188-371,
solve! in lib/NonlinearSolveBase/src/solve.jl:465-614, one-shot `__generated_polysolve` :657-790), RobustMultiNewton and
FastShortcutNLLSPolyalg (lib/NonlinearSolveFirstOrder/src/poly_algs.jl:21-88). Host control logic only — every rung of the
ladder is an nk_solver on the device; no arithmetic on vectors happens here (residual norms come from the solver's own
reduced scalars).  From Julia the reference's own NonlinearSolvePolyAlgorithm wraps the plugin algorithm unchanged (it only
needs `__solve` / `__init` of the sub-algorithms), see INTEGRATION.md."""
from __future__ import annotations

import math
from dataclasses import dataclass

from . import core as K

RETAIN_REPROBE_INTERVAL = 8   # polyalg.jl:186


@dataclass
class NonlinearSolvePolyAlgorithm:
    algs: tuple
    start_index: int = 1      # 1-based, as in the reference

    def __post_init__(self):
        self.algs = tuple(self.algs)
        if not (0 < self.start_index <= len(self.algs)):
            raise ValueError("start_index must lie in 1..length(algs)")


def RobustMultiNewton(concrete_jac=None, linsolve=None):
    kw = dict(concrete_jac=concrete_jac, linsolve=linsolve)
    RUS = K.RadiusUpdateSchemes
    return NonlinearSolvePolyAlgorithm((
        K.TrustRegion(**kw), K.TrustRegion(radius_update_scheme=RUS.Bastin, **kw), K.NewtonRaphson(**kw),
        K.NewtonRaphson(linesearch=K.BackTracking(), **kw), K.TrustRegion(radius_update_scheme=RUS.NLsolve, **kw),
        K.TrustRegion(radius_update_scheme=RUS.Fan, **kw)))


def FastShortcutNLLSPolyalg(concrete_jac=None, linsolve=None):
    RUS = K.RadiusUpdateSchemes
    return NonlinearSolvePolyAlgorithm((
        K.GaussNewton(linsolve=linsolve, concrete_jac=concrete_jac), K.LevenbergMarquardt(linsolve=linsolve, disable_geodesic=True),
        K.TrustRegion(linsolve=linsolve, concrete_jac=concrete_jac),
        K.GaussNewton(linsolve=linsolve, linesearch=K.BackTracking(), concrete_jac=concrete_jac),
        K.TrustRegion(linsolve=linsolve, radius_update_scheme=RUS.Fan, concrete_jac=concrete_jac),
        K.LevenbergMarquardt(linsolve=linsolve)))


def _sum_stats(parts):
    out = K.NLStats()
    for st in parts:
        for k in vars(out):
            setattr(out, k, getattr(out, k) + getattr(st, k))
    return out


def _resnorm(cache, least_squares):
    """‖fu‖ of a sub-cache as findmin_resids takes it (∞-norm; 2-norm for least squares), NaN → Inf. The ∞-norm is the
    solver's own globally reduced scalar; the 2-norm is reduced over the ranks of a partitioned problem."""
    if not least_squares:
        v = cache.fnorm_inf
    else:
        fu = cache.fu
        s = float((fu * fu).sum())
        if K.torch.distributed.is_available() and K.torch.distributed.is_initialized() and \
                cache.prob.device_problem.n_local != cache.prob.device_problem.n:
            t = K.torch.tensor([s], dtype=K.torch.float64)
            if K.torch.distributed.get_backend() == "nccl":
                t = t.cuda()
            K.torch.distributed.all_reduce(t)
            s = float(t[0])
        v = math.sqrt(s)
    return math.inf if math.isnan(v) else v


def _findmin(norms):
    """polyalg.jl:412-430 on precomputed norms (None = never attempted): the earliest of equal minima wins."""
    idx = next(i for i, r in enumerate(norms) if r is not None)
    best, bi = math.inf, -1
    for j in range(idx + 1, len(norms)):
        fx = math.inf if norms[j] is None else norms[j]
        if fx < best:
            best, bi = fx, j
    return bi if (bi >= 0 and best < norms[idx]) else idx


class PolyAlgorithmCache:
    """NonlinearSolvePolyAlgorithmCache: `init` builds every sub-cache, `solve!` runs them in order from `current`, `reinit!`
    offers best-sub-algorithm retention (sticky start, wrap-around floored at start_index, a re-probe every 8th retained
    reinit!, lazy sub-cache reinitialisation). Statistics: the reference shares ONE NLStats among the sub-caches and every
    sub-cache reinit! zeroes it — the sum over the sub-caches that ran since the last reinitialisation of any of them."""

    def __init__(self, prob, alg: NonlinearSolvePolyAlgorithm, least_squares=False, **kw):
        self.prob, self.alg, self.least_squares = prob, alg, least_squares
        self.caches = [K.FirstOrderCache(prob, a, **kw) for a in alg.algs]
        self.N = len(self.caches)
        self.best, self.current = -1, alg.start_index
        self.retain_best, self.start_current, self.wrapped, self.retain_count = False, alg.start_index, False, 0
        self.deferred = (None, None)
        self.retcode, self.force_stop, self.nsteps = "Default", False, 0
        self._ran = []
        self.u0 = prob.u0

    @property
    def stats(self):
        return _sum_stats(c.stats for c in self._ran)

    def _note(self, c):
        if all(c is not r for r in self._ran):
            self._ran.append(c)

    def _deferred_reinit(self, i):
        self.caches[i - 1].reinit(self.deferred[0], self.deferred[1])
        self._ran = []

    def reinit(self, u0=None, p=None, retain_best=False):
        if u0 is not None:
            self.u0 = u0
        self.retain_best = retain_best
        self.retain_count = self.retain_count + 1 if retain_best else 0
        retained = retain_best and 1 <= self.best <= self.N
        reprobe = retained and self.best > self.alg.start_index and self.retain_count % RETAIN_REPROBE_INTERVAL == 0
        self.current = self.best if (retained and not reprobe) else self.alg.start_index
        self.start_current, self.wrapped = self.current, False
        if retain_best:
            self.deferred = (self.u0, p)
            self.caches[self.current - 1].reinit(self.u0, p)
        else:
            for c in self.caches:
                c.reinit(self.u0, p)
        self._ran = []
        self.nsteps, self.force_stop, self.retcode = 0, False, "Default"
        return self

    def _attempt(self, i):
        c = self.caches[i - 1]
        if self.retain_best and i != self.start_current:
            self._deferred_reinit(i)
        self._note(c)
        sol = c.solve()
        if sol.retcode == "Success":
            self.best, self.retcode = i, sol.retcode
            return K.NonlinearSolution(sol.u, sol.resid, sol.retcode, self.stats, sol.trace, self.alg)
        self.current = i + 1
        return None

    def solve(self):
        for i in range(1, self.N + 1):
            if i == self.current:
                out = self._attempt(i)
                if out is not None:
                    return out
        if self.retain_best and not self.wrapped and self.start_current > self.alg.start_index:
            self.wrapped, self.current = True, self.alg.start_index
        for i in range(1, self.N):
            if self.wrapped and i == self.current and i < self.start_current:
                out = self._attempt(i)
                if out is not None:
                    return out
        idx = _findmin([_resnorm(c, self.least_squares) for c in self.caches])
        c = self.caches[idx]
        self.retcode = c.retcode
        return K.NonlinearSolution(c.u, c.fu, c.retcode, self.stats, c.trace if c._opts.store_trace else [], self.alg)

    def step(self):
        if not (1 <= self.current <= self.N):
            if self.retain_best and not self.wrapped and self.start_current > self.alg.start_index:
                self.wrapped, self.current = True, self.alg.start_index
                self._deferred_reinit(self.current)
                return
            idx = _findmin([_resnorm(c, self.least_squares) for c in self.caches])
            self.best, self.retcode, self.force_stop = idx + 1, self.caches[idx].retcode, True
            return
        i = self.current
        c = self.caches[i - 1]
        self._note(c)
        c.step()
        self.nsteps += 1
        rc, nst, stop = c._ret()
        if stop or nst >= c._opts.maxiters:
            name = K.L.RET_NAMES[rc] if K.L.RET_NAMES[rc] != "Default" else ("MaxIters" if nst >= c._opts.maxiters else "Success")
            if name == "Success":
                self.best, self.force_stop, self.retcode = i, True, name
            elif self.wrapped and i + 1 >= self.start_current:
                idx = _findmin([_resnorm(cc, self.least_squares) for cc in self.caches])
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

    def close(self):
        for c in self.caches:
            c.close()


def polysolve(prob, alg: NonlinearSolvePolyAlgorithm, least_squares=False, **kw):
    """The one-shot path (`__generated_polysolve`): each sub-algorithm's cache is built only when the ladder reaches it and
    released before the next one is built; first success wins, else the lowest residual among the sub-algorithms that ran."""
    sols, norms, stats = [None] * len(alg.algs), [None] * len(alg.algs), []
    for i in range(alg.start_index, len(alg.algs) + 1):
        c = K.FirstOrderCache(prob, alg.algs[i - 1], **kw)
        try:
            sol = c.solve()
            stats.append(sol.stats)
            sols[i - 1], norms[i - 1] = sol, _resnorm(c, least_squares)
        finally:
            c.close()
        if sol.retcode == "Success":
            break
    else:
        sol = sols[_findmin(norms)]
    return K.NonlinearSolution(sol.u, sol.resid, sol.retcode, _sum_stats(stats), sol.trace, alg)
