import sys, time
sys.path.insert(0, "/root/repo")
import torch
import nonlinearsolve_jl_amd as nls
for ns in (1024, 4096):
    prob = nls.NonlinearProblem(nls.Bratu2D(ns, 6.0), u0=torch.zeros(ns * ns, dtype=torch.float64, device="cuda"))
    alg = nls.NewtonRaphson(linsolve=nls.KrylovJL_GMRES(precs=nls.MultigridPrecs(2, 31)), forcing=nls.EisenstatWalkerForcing2())
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        cache = nls.init(prob, alg, abstol=1e-8, maxiters=50)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        sol = nls.solve_(cache)
        torch.cuda.synchronize(); t2 = time.perf_counter()
        nls.reinit_(cache, torch.zeros(ns * ns, dtype=torch.float64, device="cuda"))
        torch.cuda.synchronize(); t3 = time.perf_counter()
        sol2 = nls.solve_(cache)
        torch.cuda.synchronize(); t4 = time.perf_counter()
        cache.close()
        print(ns, "init %.1f ms  first solve %.1f ms  reinit %.1f ms  second solve (hierarchy warm) %.1f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3), sol.retcode, sol.stats.nsteps, sol.stats.gmres_iters, flush=True)
