"""Config C2: Bratu 256², NewtonRaphson + concrete sparse J + direct (banded LU) solve on one GPU."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import nonlinearsolve_jl_amd as nls
ns = int(sys.argv[1]) if len(sys.argv) > 1 else 256
prob = nls.NonlinearProblem(nls.Bratu2D(ns, 6.0))
for rep in range(2):
    torch.cuda.synchronize(); t = time.perf_counter()
    sol = nls.solve(prob, nls.NewtonRaphson(), abstol=1e-8, maxiters=50)
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    print(f"C2 direct n={ns}^2: steps {sol.stats.nsteps} nfactors {sol.stats.nfactors} time {dt:.3f} s {sol.retcode} "
          f"|F|inf={float(np.max(np.abs(sol.resid))):.2e}")
