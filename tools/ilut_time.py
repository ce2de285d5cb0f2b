#!/usr/bin/env python
"""ILU(τ) behind `precs` (csrc/nk_precond.hip::ilut_update): what the HOST factorisation of every new Jacobian costs next to the
device work of a Newton step — the tutorial's `incompletelu(W, p) = (ilu(W, τ = 50.0), I)` on the Brusselator
(docs/src/tutorials/large_systems.md:252-260). Per size: factorisation (host, per `update()`), fill, levels, application (device),
and a NewtonRaphson solve with it as Pl — wall time, steps, Krylov iterations, the share of the factorisations.
    python tools/ilut_time.py [N …]     (default 32 128 256 512)"""
import os
import sys
import time

import ctypes as C

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nonlinearsolve_jl_amd as nls  # noqa: E402


def sync():
    torch.cuda.synchronize()


def timeit(fn, reps):
    fn(); sync()
    t = time.perf_counter()
    for _ in range(reps):
        fn()
    sync()
    return (time.perf_counter() - t) / reps


def run(N, tau=50.0):
    PB = nls.Brusselator2D(N)
    u = PB.initial_guess(device=True)
    J = PB.jac_csr()
    PB.jac_values(u, J)
    n = u.numel()
    b = torch.ones(n, dtype=torch.float64, device="cuda")
    t0 = time.perf_counter()
    M = nls.ILUTPreconditioner(J, tau)
    sync()
    t_create = time.perf_counter() - t0
    reps = 3 if n > 100000 else 10
    t_upd = timeit(M.update, reps)
    t_app = timeit(lambda: M.apply(b), 20)
    inf = M.info()
    nnzA = J.info()["nnz"]
    from nonlinearsolve_jl_amd import _lib as L
    nnzf = C.c_int64(0)
    L.lib().nk_precond_ilu0_factors(M._h, C.byref(nnzf), None, None, None, None)
    inf["nnz_factors"] = nnzf.value
    state = {"t": 0.0, "calls": 0}

    def incompletelu(W, p=None):
        t = time.perf_counter()
        if "M" not in state:
            state["M"] = nls.ILUTPreconditioner(W, tau)
        else:
            state["M"].update()
        state["t"] += time.perf_counter() - t
        state["calls"] += 1
        return state["M"], None

    alg = nls.NewtonRaphson(linsolve=nls.KrylovJL_GMRES(precs=incompletelu, reltol=1e-8, abstol=0.0, gmres_restart=30, maxiters=3000), concrete_jac=True)
    sync()
    t0 = time.perf_counter()
    sol = nls.solve(nls.NonlinearProblem(PB, u0=PB.initial_guess(device=True)), alg, abstol=1e-8, maxiters=50)
    sync()
    t_solve = time.perf_counter() - t0
    st = sol.stats
    print(f"| Brusselator {N}² | {n} | {nnzA} | {inf.get('nnz_factors', inf.get('nnz', '—'))} | {inf.get('levels_lower', '—')} / {inf.get('levels_upper', '—')} | "
          f"{1e3 * t_create:.1f} | {1e3 * t_upd:.1f} | {1e6 * t_app:.0f} | {sol.retcode} | {st.nsteps} | {st.gmres_iters} | {1e3 * t_solve:.1f} | "
          f"{1e3 * state['t']:.1f} ({state['calls']} calls) | {100.0 * state['t'] / t_solve:.0f} % |", flush=True)


def main():
    sizes = [int(a) for a in sys.argv[1:]] or [32, 128, 256, 512]
    print("| problem | n | nnz(J) | nnz(L + U) | levels L / U | create ms (first factorisation + schedules) | refactorisation ms (host, per new J) | apply µs (device) | retcode | Newton steps | GMRES iterations | solve ms | of which inside precs (factorisations) | share |")
    print("|---|---:|---:|---:|---:|---:|---:|---:|---|---:|---:|---:|---:|---:|")
    for N in sizes:
        try:
            run(N)
        except Exception as ex:  # noqa: BLE001
            print(f"| Brusselator {N}² | failed: {str(ex)[:200]} |", flush=True)


if __name__ == "__main__":
    main()
