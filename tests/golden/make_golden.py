#!/usr/bin/env python
"""Generates tests/golden/oracle_golden.npz.

These are NOT outputs of the reference (Julia is unavailable in the build container and the reference ships no
golden vectors): they are outputs of the CPU oracle (oracle/reference_restatement.py), cross-checked here
against SciPy before being written, and committed so that (a) the oracle cannot drift silently and (b) the
GPU parity tests have fixed vectors that travel to the GPU box."""
import os
import sys

import numpy as np
import scipy.optimize

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from oracle import reference_restatement as R  # noqa: E402

out = {}
p = R.Bratu2D(16)
s = R.solve(p, R.NewtonRaphson(), abstol=1e-10, maxiters=50)
chk = scipy.optimize.root(p.f, np.zeros(p.n), method="krylov", options=dict(fatol=1e-12))
assert np.max(np.abs(chk.x - s.u)) < 1e-8, "oracle disagrees with scipy.optimize.root"
out["bratu16_u"] = s.u
out["v256"] = np.random.default_rng(256).standard_normal(256)
out["bratu16_Jv"] = p.jac(s.u) @ out["v256"]
b = R.Brusselator2D(8)
sb = R.solve(b, R.NewtonRaphson(), abstol=1e-10)
chk = scipy.optimize.root(b.f, b.u0(), method="krylov", options=dict(fatol=1e-10))
assert np.max(np.abs(chk.x - sb.u)) < 1e-6
out["brus8_u"] = sb.u
out["brus8_u0"] = b.u0()
out["brus8_f0"] = b.f(b.u0())
np.savez(os.path.join(os.path.dirname(os.path.abspath(__file__)), "oracle_golden.npz"), **out)
print("wrote", sorted(out))
