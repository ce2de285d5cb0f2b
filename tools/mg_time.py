import sys, time
sys.path.insert(0, "/root/repo")
import torch, numpy as np
import nonlinearsolve_jl_amd as nls
for ns in (1024, 4096):
    for nu in (1, 2):
        for coarse in (63, 31, 15, 7):
            prob = nls.NonlinearProblem(nls.Bratu2D(ns, 6.0), u0=torch.zeros(ns * ns, dtype=torch.float64, device="cuda"))
            alg = nls.NewtonRaphson(linsolve=nls.KrylovJL_GMRES(precs=nls.MultigridPrecs(nu, coarse)), forcing=nls.EisenstatWalkerForcing2(), concrete_jac=False)
            best = 1e9
            for rep in range(3):
                torch.cuda.synchronize(); t = time.perf_counter()
                sol = nls.solve(prob, alg, abstol=1e-8, maxiters=50)
                torch.cuda.synchronize(); best = min(best, time.perf_counter() - t)
            print(ns, "nu", nu, "coarse", coarse, sol.retcode, sol.stats.nsteps, sol.stats.gmres_iters, round(best * 1e3, 1), "ms", flush=True)
