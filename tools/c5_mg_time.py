"""Config C5 (Brusselator 512², TrustRegion + GMRES(30)) with the multigrid V-cycle behind `precs`: init vs solve split.
  python tools/c5_mg_time.py [N] [matfree]"""
import sys, time
sys.path.insert(0, "/root/repo")
import torch
import nonlinearsolve_jl_amd as nls
N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
concrete = not (len(sys.argv) > 2 and sys.argv[2] == "matfree")
P = nls.Brusselator2D(N)
alg = nls.TrustRegion(linsolve=nls.KrylovJL_GMRES(gmres_restart=30, maxiters=300, reltol=1e-9, abstol=0.0,
                                                  precs=nls.MultigridPrecs(2, 16)), concrete_jac=concrete)
for rep in range(3):
    u0 = P.initial_guess(device=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    cache = nls.init(nls.NonlinearProblem(P, u0=u0), alg, abstol=1e-7, maxiters=30)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    sol = nls.solve_(cache)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    nls.reinit_(cache, P.initial_guess(device=True))
    torch.cuda.synchronize(); t3 = time.perf_counter()
    sol2 = nls.solve_(cache)
    torch.cuda.synchronize(); t4 = time.perf_counter()
    cache.close()
    print(N, "concrete" if concrete else "matfree", "init %.1f ms  first solve %.1f ms  reinit %.1f ms  second solve (hierarchy warm) %.1f ms"
          % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3), sol2.retcode, sol2.stats.nsteps,
          sol2.stats.gmres_iters, flush=True)
