#!/usr/bin/env python
"""SpMV / JVP micro-benchmark with the measurement hygiene of SURVEY.md §8(d): 20 warm-ups, 200 individually timed launches
(the kernel's own begin→end device timestamps via the library's profile hooks), median + p10/p90; *warm* = launches
back-to-back on the same operands (the 63 MB matrix of Bratu 1024² then lives in the 256 MiB Infinity Cache), *cold* =
a 768 MB memset between launches flushes it; the 4096² matrix (1.07 GB) never fits.

    python tools/spmv_bench.py [reps=200]
"""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch

import nonlinearsolve_jl_amd as nls

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
ctx = nls.default_context()
flush = torch.empty(768 * 1024 * 1024, dtype=torch.uint8, device="cuda")


def sample(fn, key, cold):
    ts = []
    for i in range(20 + reps):
        if cold:
            flush.fill_(i & 255)
        ctx.profile_enable(True)
        fn()
        r = ctx.profile_report()[key]
        if i >= 20:
            ts.append(r["avg_us"])
    ctx.profile_enable(False)
    return np.array(ts), r["bytes"] / r["launches"]


rows = []
for ns in (1024, 4096):
    P = nls.Bratu2D(ns, 6.0)
    n = ns * ns
    u = torch.zeros(n, dtype=torch.float64, device="cuda")
    v = torch.randn(n, dtype=torch.float64, device="cuda")
    J = P.jac_csr()
    P.jac_values(u, J)
    y = torch.empty_like(v)
    for name, fn, key in (("csr_spmv", lambda: J.matvec(v, out=y), "spmv"), ("matfree_jvp", lambda: P.jvp(v, u), "jvp")):
        for cold in (False, True):
            if ns == 4096 and not cold and name == "csr_spmv":
                pass  # "warm" at 4096² is still HBM: the matrix is 4× the Infinity Cache
            t, by = sample(fn, key, cold)
            med, p10, p90 = np.median(t), np.percentile(t, 10), np.percentile(t, 90)
            rows.append(dict(grid=ns, op=name, state="cold" if cold else "warm", reps=reps, algorithmic_MB=round(by / 1e6, 2),
                             median_us=round(float(med), 2), p10_us=round(float(p10), 2), p90_us=round(float(p90), 2),
                             median_GBps=round(by / med / 1e3, 1), frac_of_8TBs=round(by / med / 1e3 / 8000.0, 3),
                             frac_of_6p29TBs=round(by / med / 1e3 / 6290.0, 3)))
            print(json.dumps(rows[-1]), flush=True)
