#!/usr/bin/env python
"""Runs the BASELINE.json configs that fit one GPU and prints one JSON line each (results table of BASELINE.md)."""
import json, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import nonlinearsolve_jl_amd as nls


def timed(fn):
    torch.cuda.synchronize(); t = time.perf_counter(); r = fn(); torch.cuda.synchronize()
    return r, time.perf_counter() - t


def report(name, sol, dt, **extra):
    print(json.dumps(dict(config=name, retcode=sol.retcode, nsteps=sol.stats.nsteps, gmres_iters=sol.stats.gmres_iters,
                          nf=sol.stats.nf, njacs=sol.stats.njacs, nfactors=sol.stats.nfactors, seconds=round(dt, 4),
                          steps_per_s=round(sol.stats.nsteps / dt, 2),
                          fnorm_inf=float(np.max(np.abs(sol.resid.cpu().numpy() if hasattr(sol.resid, "cpu") else sol.resid))),
                          **extra)))


dev = "cuda"
# C1: quadratic, u0 = ones(1000), default tolerance, direct solve (plumbing)
sol, dt = timed(lambda: nls.solve(nls.NonlinearProblem(nls.Quadratic(1000, 2.0)), nls.NewtonRaphson()))
report("C1 quadratic N=1000 NewtonRaphson() direct", sol, dt, err=float(np.max(np.abs(sol.u - np.sqrt(2)))))
# C2: Bratu 256², sparse J (closed form) + direct solve; and coloured assembly check
P = nls.Bratu2D(256, 6.0)
sol, dt = timed(lambda: nls.solve(nls.NonlinearProblem(P), nls.NewtonRaphson(), abstol=1e-8, maxiters=50))
J1, J2 = P.jac_csr(), P.jac_csr()
u = torch.tensor(sol.u, device=dev)
P.jac_values(u, J1); nc = P.jac_values(u, J2, colored=True)
report("C2 bratu 256^2 NewtonRaphson() sparse J + banded LU", sol, dt, umax=float(np.max(sol.u)), ncolors=nc,
       colored_vs_closed_form=float(np.max(np.abs(J1.values() - J2.values()))))
# C3: Bratu 1024², matrix-free JVP, GMRES(30), Eisenstat–Walker, inner cap 300, maxiters 50
for concrete in (False, True):
    prob = nls.NonlinearProblem(nls.Bratu2D(1024, 6.0), u0=torch.zeros(1024 * 1024, dtype=torch.float64, device=dev))
    alg = nls.NewtonRaphson(linsolve=nls.KrylovJL_GMRES(gmres_restart=30, maxiters=300), forcing=nls.EisenstatWalkerForcing2(),
                            concrete_jac=concrete)
    sol, dt = timed(lambda: nls.solve(prob, alg, abstol=1e-8, maxiters=50))
    report(f"C3 bratu 1024^2 NR + GMRES(30) + EW ({'CSR' if concrete else 'matrix-free'}), cap 300/step, maxiters 50", sol, dt)
# C5 (one GPU): Brusselator 512², TrustRegion + GMRES(30) rtol 1e-4, matrix-free JVP/VJP, maxiters 20
PB = nls.Brusselator2D(512)
prob = nls.NonlinearProblem(PB, u0=PB.initial_guess(device=True))
alg = nls.TrustRegion(linsolve=nls.KrylovJL_GMRES(gmres_restart=30, maxiters=300, reltol=1e-4, abstol=0.0))
sol, dt = timed(lambda: nls.solve(prob, alg, abstol=1e-8, maxiters=20))
report("C5 brusselator 512^2 TrustRegion + GMRES(30) rtol 1e-4 (1 GPU), maxiters 20", sol, dt)
