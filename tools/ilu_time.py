"""Cost and effect of the device preconditioner objects behind `precs` (csrc/nk_precond.hip) on the path's matrices:
set-up / refactorisation / application time and GMRES(30) iterations to rtol 1e-8 with the object as Pl, for Jacobi, ILU(0)
multicolour and ILU(0) natural ordering.   python tools/ilu_time.py [sizes …]   (default: Bratu 256 1024, Brusselator 32 128)"""
import os
import sys
import time

import numpy as np
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


def run(name, P):
    dev = "cuda:0"
    u = P.initial_guess(device=True)
    u += 0.1 * torch.sin(torch.arange(u.numel(), device=dev, dtype=torch.float64) * 0.37)
    J = P.jac_csr()
    P.jac_values(u, J)
    n = u.numel()
    b = torch.ones(n, dtype=torch.float64, device=dev)
    rows = []
    for kind in ("none", "jacobi", "ilu0_multicolor", "ilu0_natural"):
        if kind == "ilu0_natural" and n > 300000:
            continue   # (a dependency chain of 2n − 1 levels: measured at the smaller sizes)
        G = nls.GMRES(n, restart=30).set_operator(J)
        t0 = time.perf_counter()
        M = None
        if kind == "jacobi":
            M = nls.JacobiPreconditioner(J)
        elif kind.startswith("ilu0"):
            M = nls.ILU0Preconditioner(J, ordering=kind.split("_")[1])
        sync()
        t_setup = time.perf_counter() - t0
        t_upd = timeit(M.update, 5) if M else 0.0
        t_app = timeit(lambda: M.apply(b), 20) if M else 0.0
        if M:
            G.set_preconditioner(M, "left")
        t0 = time.perf_counter()
        x, info = G.solve(b, reltol=1e-8, abstol=0.0, maxiters=3000)
        sync()
        t_solve = time.perf_counter() - t0
        r = b - J.matvec(x)
        rows.append(f"| {name} | {n} | {kind} | {M.info()['levels_lower'] if M and kind != 'jacobi' else '—'} | {1e3 * t_setup:.1f} | "
                    f"{1e3 * t_upd:.2f} | {1e6 * t_app:.0f} | {info['iters']} | {1e3 * t_solve:.1f} | {float(r.norm() / b.norm()):.1e} |")
        G.close()
    return rows


def main():
    print("| matrix | n | Pl | levels | set-up ms (symbolic + first factorisation) | refactorisation ms | apply µs | GMRES(30) iterations to rtol 1e-8 | solve ms | true ‖b − A x‖/‖b‖ |")
    print("|---|---:|---|---:|---:|---:|---:|---:|---:|---:|")
    for name, P in (("Bratu 256²", nls.Bratu2D(256)), ("Bratu 1024²", nls.Bratu2D(1024)),
                    ("Brusselator 32²", nls.Brusselator2D(32)), ("Brusselator 128²", nls.Brusselator2D(128))):
        for r in run(name, P):
            print(r, flush=True)


if __name__ == "__main__":
    main()
