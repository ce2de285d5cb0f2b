#!/usr/bin/env python
"""Generates tests/golden/c5_brusselator512_tr_direct.npz — config C5 at FULL size on the CPU oracle.

Brusselator 2-D steady state, N_g = 512 (524 288 unknowns), TrustRegion() with the concrete sparse Jacobian and a DIRECT
linear solve (SciPy SuperLU): the reference's own `TrustRegion()` default path (linsolve = nothing). The run takes a few
minutes and several GB, which is why its outcome travels as a fixture instead of being recomputed in the GPU tests:
step count, accept/reject sequence, trust radii, ‖f‖∞ per step, norms of the solution and the solution itself at every
64th unknown (8 192 samples). NOT an output of the reference (Julia is unavailable): an output of the CPU oracle."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from oracle import reference_restatement as R  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
pb = R.Brusselator2D(N)
t = time.time()
# abstol 1e-7: at N_g = 512 the residual itself carries ≈ 1.4e-8 of rounding (α/dx² = 2.6e6 times eps·|u|) — the direct-solve
# oracle stagnates at ‖f‖∞ = 1.36e-8 with abstol = 1e-8, so the full-size comparison asks for 1e-7
s = R.solve(pb, R.TrustRegion(), abstol=1e-7, maxiters=30)
dt = time.time() - t
print("retcode", R.RETCODE_NAMES[s.retcode], "nsteps", s.stats.nsteps, "fnorm", [r["fnorm_inf"] for r in s.trace],
      "accepted", [int(r["accepted"]) for r in s.trace], "seconds", round(dt, 1), flush=True)
assert s.retcode == R.SUCCESS and np.max(np.abs(pb.f(s.u))) <= 1e-7
out = dict(N=N, nsteps=s.stats.nsteps, accepted=np.array([int(r["accepted"]) for r in s.trace]),
           trust_region=np.array([r["trust_region"] for r in s.trace]), fnorm_inf=np.array([r["fnorm_inf"] for r in s.trace]),
           u_l2=np.linalg.norm(s.u), u_inf=np.max(np.abs(s.u)), stride=64, u_samples=s.u[::64].copy(), seconds=dt)
name = "c5_brusselator%d_tr_direct.npz" % N
np.savez(os.path.join(os.path.dirname(os.path.abspath(__file__)), name), **out)
print("wrote", name, {k: (v if np.ndim(v) == 0 else np.shape(v)) for k, v in out.items()})
