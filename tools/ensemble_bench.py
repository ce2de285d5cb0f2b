"""Ensemble (kernel-generation) throughput: SimpleNewtonRaphson on the tutorial's p2_f and on quadratic systems,
one system per GPU thread, device-resident inputs; the oracle's per-system Python loop is NOT a baseline — the C
restatement in oracle/nk_oracle.c (OpenMP over systems) is.

    python tools/ensemble_bench.py [nbatch=1048576]
"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
import torch

import ensemble_sources as E
import nonlinearsolve_jl_amd as nls

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
rng = np.random.default_rng(0)
for name, src, n, u0, P, maxit in [
    ("quadratic n=4 (u.*u .- p)", E.QUADRATIC, 4, np.ones(4), rng.uniform(1.0, 100.0, (nb, 4)), 1000),
    ("quadratic n=8", E.QUADRATIC, 8, np.ones(8), rng.uniform(1.0, 100.0, (nb, 8)), 1000),
    ("tutorial p2_f n=4, maxiters 100", E.P2, 4, np.array([1.0, 2.0, 3.0, 4.0]), rng.random((nb, 4)) + 0.05, 100),
]:
    prob = nls.ImmutableNonlinearProblem(src, torch.tensor(u0, device="cuda"), torch.tensor(P, device="cuda"))
    sol = nls.vectorized_solve(prob, maxiters=maxit)   # compile + warm-up
    torch.cuda.synchronize()
    # device time of the C-ABI call itself (events on the ctx's = torch's default stream; everything device resident)
    import ctypes as C
    from nonlinearsolve_jl_amd import _lib as L
    from nonlinearsolve_jl_amd.core import _BatchKernel
    h = _BatchKernel.get(prob.ctx, src, n, P.shape[1], 0)
    du0, dp = prob.u0.contiguous(), prob.p.contiguous()
    du, dr = torch.empty((nb, n), dtype=torch.float64, device="cuda"), torch.empty((nb, n), dtype=torch.float64, device="cuda")
    drc, dit = torch.empty(nb, dtype=torch.int32, device="cuda"), torch.empty(nb, dtype=torch.int32, device="cuda")
    ptr = lambda x: C.c_void_p(x.data_ptr())
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        st = L.lib().nk_batch_solve(h, nb, ptr(du0), 0, ptr(dp), L.DEVICE, 0.0, maxit, ptr(du), ptr(dr), ptr(drc), ptr(dit))
        e1.record()
        torch.cuda.synchronize()
        assert st == 0
        ts.append(e0.elapsed_time(e1) * 1e-3)
    dt = min(ts)
    iters = sol.iters.astype(np.int64)
    # flops per Newton iteration: residual + dual-number Jacobian (~(1 + 2n)·c_f) is problem specific; count the LU only
    lu_flops = (2.0 / 3.0) * n ** 3 + 2.0 * n ** 2
    print(f"{name}: {nb} systems in {dt * 1e3:.2f} ms = {nb / dt / 1e6:.1f} M systems/s, "
          f"mean {iters.mean():.1f} Newton iterations (max {iters.max()}), "
          f"{(sol.retcode_raw == 1).mean() * 100:.1f} % Success, LU alone {iters.sum() * lu_flops / dt / 1e12:.2f} TFLOP/s")
    if "--cpu" in sys.argv and not name.startswith("quadratic n=8"):
        from oracle import c_oracle as CO
        CO.ensemble_newton(1 if "p2" in name else 0, u0, P[:1000], maxiters=maxit)   # build / warm up
        t = time.perf_counter()
        uc, rcpu, rcc, itc = CO.ensemble_newton(1 if "p2" in name else 0, u0, P, maxiters=maxit)
        tc = time.perf_counter() - t
        ug = du.cpu().numpy()
        same = (itc == dit.cpu().numpy()) & (rcc == drc.cpu().numpy())
        print(f"   C oracle, {CO.num_threads()} host threads: {nb / tc / 1e6:.1f} M systems/s ({tc * 1e3:.1f} ms); GPU/CPU = "
              f"{tc / dt:.0f}x; identical (retcode, iterations) on {same.mean() * 100:.2f} % of the systems, "
              f"max |u_gpu - u_cpu| there = {np.nanmax(np.abs(ug[same] - uc[same])):.2e}")
