timeout 300 python -m pytest tests/test_gpu_solvers.py tests/test_gpu_multirank.py -m gpu -q -x --timeout 300 -k "chebyshev or two_ranks or precond" 2>&1 | tail -2
python - <<PY
import time, numpy as np, torch
import nonlinearsolve_jl_amd as nls
for concrete in (True, False):
    for rep in range(2):
        prob = nls.NonlinearProblem(nls.Bratu2D(1024, 6.0), u0=torch.zeros(1024*1024, dtype=torch.float64, device="cuda"))
        alg = nls.NewtonRaphson(linsolve=nls.KrylovJL_GMRES(gmres_restart=30, maxiters=300, precs=nls.ChebyshevPrecs(32, 300.0)), forcing=nls.EisenstatWalkerForcing2(), concrete_jac=concrete)
        torch.cuda.synchronize(); t=time.perf_counter(); sol = nls.solve(prob, alg, abstol=1e-8, maxiters=50); torch.cuda.synchronize()
        print("concrete" if concrete else "matfree", sol.retcode, sol.stats.nsteps, sol.stats.gmres_iters, round(time.perf_counter()-t,4))
PY
