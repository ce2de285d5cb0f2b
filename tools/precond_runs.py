#!/usr/bin/env python
"""Time-to-tolerance with the built-in Chebyshev `precs` on the configs that stagnate unpreconditioned."""
import json, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import nonlinearsolve_jl_amd as nls


def run(name, prob, alg, **kw):
    torch.cuda.synchronize(); t = time.perf_counter()
    sol = nls.solve(prob, alg, **kw)
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    r = sol.resid.cpu().numpy() if hasattr(sol.resid, "cpu") else sol.resid
    print(json.dumps(dict(config=name, retcode=sol.retcode, nsteps=sol.stats.nsteps, gmres_iters=sol.stats.gmres_iters,
                          op_applies=sol.stats.op_applies, seconds=round(dt, 4), fnorm_inf=float(np.max(np.abs(r))))))
    return sol


dev = "cuda"
for ns in (1024, 4096):
    for deg, ratio in ((32, 300.0), (64, 1000.0)):
        for concrete in (True, False):
            u0 = torch.zeros(ns * ns, dtype=torch.float64, device=dev)
            prob = nls.NonlinearProblem(nls.Bratu2D(ns, 6.0), u0=u0)
            alg = nls.NewtonRaphson(linsolve=nls.KrylovJL_GMRES(gmres_restart=30, maxiters=300,
                                                                precs=nls.ChebyshevPrecs(deg, ratio)),
                                    forcing=nls.EisenstatWalkerForcing2(), concrete_jac=concrete)
            run(f"bratu {ns}^2 NR+GMRES(30)+EW+Chebyshev({deg},{ratio:g}) {'CSR' if concrete else 'matfree'}", prob, alg,
                abstol=1e-8, maxiters=50)
for ns in (1024, 4096):
    for nu in (1, 2):
        for concrete in (True, False):
            for rep in range(2):  # second run: hierarchy and buffers warm
                u0 = torch.zeros(ns * ns, dtype=torch.float64, device=dev)
                prob = nls.NonlinearProblem(nls.Bratu2D(ns, 6.0), u0=u0)
                alg = nls.NewtonRaphson(linsolve=nls.KrylovJL_GMRES(gmres_restart=30, maxiters=300, precs=nls.MultigridPrecs(nu, 31)),
                                        forcing=nls.EisenstatWalkerForcing2(), concrete_jac=concrete)
                sol = run(f"bratu {ns}^2 NR+GMRES(30)+EW+Multigrid(nu={nu}) {'CSR' if concrete else 'matfree'} run{rep}", prob, alg,
                          abstol=1e-8, maxiters=50)
PB = nls.Brusselator2D(512)
for deg, ratio in ((64, 1000.0), (128, 10000.0)):
    prob = nls.NonlinearProblem(PB, u0=PB.initial_guess(device=True))
    alg = nls.TrustRegion(linsolve=nls.KrylovJL_GMRES(gmres_restart=30, maxiters=600, reltol=1e-4, abstol=0.0,
                                                      precs=nls.ChebyshevPrecs(deg, ratio)))
    run(f"brusselator 512^2 TR+GMRES(30)+Chebyshev({deg},{ratio:g}) matfree", prob, alg, abstol=1e-8, maxiters=30)
