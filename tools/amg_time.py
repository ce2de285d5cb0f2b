import sys, time; sys.path.insert(0, '/root/repo')
import numpy as np, torch
import nonlinearsolve_jl_amd as nls
for precs, name in ((nls.ObjectPrecs("amg", "left"), "amg_left"), (nls.ObjectPrecs("amg", "right"), "amg_right"), (nls.MultigridPrecs(2, 31), "geometric")):
    alg = nls.NewtonRaphson(linsolve=nls.KrylovJL_GMRES(precs=precs, gmres_restart=30, maxiters=300), forcing=nls.EisenstatWalkerForcing2(), concrete_jac=True)
    for rep in range(2):
        prob = nls.NonlinearProblem(nls.Bratu2D(1024, 6.0))
        torch.cuda.synchronize(); t0 = time.perf_counter()
        sol = nls.solve(prob, alg, abstol=1e-8, maxiters=50)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(name, rep, sol.retcode, 'steps', sol.stats.nsteps, 'gmres', sol.stats.gmres_iters, 'seconds %.4f' % dt, flush=True)
    # warm cache: init once, time solve only
    prob = nls.NonlinearProblem(nls.Bratu2D(1024, 6.0))
    cache = nls.init(prob, alg, abstol=1e-8, maxiters=50)
    cache.solve()
    cache.reinit(torch.zeros(1024*1024, dtype=torch.float64, device='cuda'))
    torch.cuda.synchronize(); t0 = time.perf_counter(); s2 = cache.solve(); torch.cuda.synchronize()
    print(name, 'initialised cache: solve %.4f s' % (time.perf_counter()-t0), s2.retcode, s2.stats.nsteps, s2.stats.gmres_iters, flush=True)
