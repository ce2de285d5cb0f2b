"""Set-up time of the aggregation AMG hierarchy on the 1024² Bratu Jacobian as a plain CSR matrix (NK_AMG_TIMING=1 prints the phases)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import nonlinearsolve_jl_amd as nls
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import reference_restatement as R
ns = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
pb = R.Bratu2D(ns)
A = pb.jac(np.zeros(pb.n)).tocsr()
A.sort_indices()
M = nls.CSRMatrix.from_scipy(A)
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    P = nls.AMGPreconditioner(M)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("AMG set-up %d^2: %.2f ms, levels %s" % (ns, 1e3 * dt, [x[0] for x in P.hierarchy()]), flush=True)
    t0 = time.perf_counter(); P.update(); torch.cuda.synchronize()
    print("   numeric refresh %.2f ms" % (1e3 * (time.perf_counter() - t0)), flush=True)
    P.close() if hasattr(P, "close") else None
