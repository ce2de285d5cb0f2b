#!/usr/bin/env python
"""Development: what the first synchronisation behind N Newton steps costs, by what the caller does between the steps.
A step never synchronises (its scalars arrive through pinned memory); the runtime keeps a record per launch until it learns
that the launches have completed. One line per variant: loop time per step, time of the closing torch.cuda.synchronize()."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import nonlinearsolve_jl_amd as nls

ns = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
prob = nls.NonlinearProblem(nls.Bratu2D(ns, 6.0))
prob.u0 = torch.zeros(ns * ns, dtype=torch.float64, device="cuda")
alg = nls.NewtonRaphson(linsolve=nls.KrylovJL_GMRES(fixed_iters=30, maxiters=30), concrete_jac=True)
cache = nls.init(prob, alg, abstol=1e-300, maxiters=10**9)
for _ in range(20):
    cache.step()
torch.cuda.synchronize()
two = [torch.cuda.Event(enable_timing=False) for _ in range(2)]
stream = torch.cuda.current_stream()


def run(label, between):
    fresh = [torch.cuda.Event(enable_timing=True) for _ in range(steps)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        cache.step()
        between(i, fresh)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(json.dumps(dict(variant=label, steps=steps, loop_ms_per_step=round((t1 - t0) / steps * 1e3, 4), closing_sync_ms=round((t2 - t1) * 1e3, 3),
                          contract_ms_per_step=round((t2 - t0) / steps * 1e3, 4))), flush=True)


for rep in range(2):
    run("nothing between the steps", lambda i, ev: None)
    run("a NEW timing event recorded per step", lambda i, ev: ev[i].record())
    run("one of TWO no-timing events re-recorded per step", lambda i, ev: two[i & 1].record())
    run("stream.query() per step", lambda i, ev: stream.query())
    run("a new event every 16th step", lambda i, ev: ev[i].record() if i % 16 == 15 else None)
    run("stream.synchronize() every 64th step", lambda i, ev: stream.synchronize() if i % 64 == 63 else None)
