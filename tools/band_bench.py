"""Banded LU micro-benchmark (config C2's linear solve): factor and solve time of the Bratu n×n Jacobian on one GPU,
with the residual ‖A x − b‖∞ / ‖b‖∞ as the check and SciPy's SuperLU on the host cores beside it.

    python tools/band_bench.py [n_side=256] [reps=3] [--cpu]
"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch

import nonlinearsolve_jl_amd as nls

ns = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 256
reps = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 3
P = nls.Bratu2D(ns, 6.0)
J = P.jac_csr()
u = np.zeros(ns * ns)
P.jac_values(u, J)
lu = nls.BandedLU(J)
info = lu.info()
n = ns * ns
flops = 2.0 * n * info["kl"] * info["ku"]
rng = np.random.default_rng(0)
b = torch.tensor(rng.standard_normal(n), device="cuda")
tf, tsv = [], []
for _ in range(reps):
    torch.cuda.synchronize(); t = time.perf_counter()
    lu.factor()
    torch.cuda.synchronize(); tf.append(time.perf_counter() - t)
    t = time.perf_counter()
    x = lu.solve(b)
    torch.cuda.synchronize(); tsv.append(time.perf_counter() - t)
r = J @ x - b
res = float(r.abs().max() / b.abs().max())
print(f"band LU n={n} kl={info['kl']} ku={info['ku']} band={info['band_bytes'] / 1e6:.0f} MB: "
      f"factor {min(tf) * 1e3:.2f} ms ({flops / min(tf) / 1e9:.1f} GFLOP/s), solve {min(tsv) * 1e3:.2f} ms, "
      f"resid {res:.2e}")
if "--cpu" in sys.argv:
    import scipy.sparse as sp
    import scipy.sparse.linalg as spl
    from oracle import reference_restatement as R
    A = sp.csc_matrix(R.Bratu2D(ns, 6.0).jac(u))
    t = time.perf_counter(); F = spl.splu(A); tc = time.perf_counter() - t
    t = time.perf_counter(); xc = F.solve(b.cpu().numpy()); ts = time.perf_counter() - t
    print(f"scipy splu (host): factor {tc * 1e3:.1f} ms, solve {ts * 1e3:.1f} ms, "
          f"|x_gpu - x_cpu|inf = {np.max(np.abs(xc - x.cpu().numpy())):.2e}")
