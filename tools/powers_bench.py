#!/usr/bin/env python
"""Resident matrix-powers kernel vs s streaming SpMV launches on the Bratu Jacobian (the kernel's own begin→end device
timestamps through the library's profile hooks). One JSON line per case; NK_PW_VARIANT / NK_SPMV_POWERS are read at load time,
so variants are separate processes:

    python tools/powers_bench.py [grid=1024] [s=15] [reps=100]
"""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch

import nonlinearsolve_jl_amd as nls

ns = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
s = int(sys.argv[2]) if len(sys.argv) > 2 else 15
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 100
ctx = nls.default_context()
P = nls.Bratu2D(ns, 6.0)
n = ns * ns
u = torch.zeros(n, dtype=torch.float64, device="cuda")
J = P.jac_csr()
P.jac_values(u, J)
x = torch.randn(n, dtype=torch.float64, device="cuda")
lam = 8.0 * (ns + 1) ** 2
theta = (0.5 + 0.4 * np.cos(np.arange(s))) * lam
ts = []
resident = None
for i in range(10 + reps):
    ctx.profile_enable(True)
    Y, resident = J.powers(x, s, theta=theta, scale=2.0 / lam)
    rep = ctx.profile_report()
    key = "spmv_powers" if resident else "spmv"
    r = rep[key]
    if i >= 10:
        ts.append(r["avg_us"] * r["launches"])
ctx.profile_enable(False)
t = np.array(ts)
by = s * (12.0 * J.info()["nnz"] + 4.0 * (n + 1) + 16.0 * n)
med = float(np.median(t))
print(json.dumps(dict(grid=ns, s=s, resident=bool(resident), variant=os.environ.get("NK_PW_VARIANT", "0"), reps=reps,
                      median_us=round(med, 2), p10_us=round(float(np.percentile(t, 10)), 2),
                      p90_us=round(float(np.percentile(t, 90)), 2), us_per_power=round(med / s, 2),
                      algorithmic_MB=round(by / 1e6, 1), effective_GBps=round(by / med / 1e3, 1),
                      checksum=float(Y[-1].abs().sum().item()))), flush=True)
