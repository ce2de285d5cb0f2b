#!/usr/bin/env python
"""Kernel micro-benchmarks on one MI355X (HIP-event timing through the library's profile hooks).
usage: python tools/microbench.py [--n 1048576] [--reps 30]"""
import argparse
import json
import os
import subprocess
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def blas(args):
    import numpy as np
    import torch
    import nonlinearsolve_jl_amd as nls
    ctx = nls.default_context()
    n = args.n
    ldv = (n + 31) // 32 * 32 + args.ldv_pad
    V = torch.randn(32, ldv, dtype=torch.float64, device="cuda")
    w = torch.randn(n, dtype=torch.float64, device="cuda")
    out = {}
    for nv in (1, 4, 8, 16, 24, 31):
        h = np.random.default_rng(0).standard_normal(nv) * 1e-3
        s = np.ones(nv)
        for name, fn in (("multidot", lambda: ctx.multidot(V[:nv], w)),
                         ("multiaxpy", lambda: ctx.multiaxpy(V[:nv], h, w, want_norm2=True)),
                         ("fused", lambda: ctx.fused_axpy_dot(V[:nv], h, s, w))):
            for _ in range(3):
                fn()
            ctx.profile_enable(True)
            for _ in range(args.reps):
                fn()
            rep = ctx.profile_report()
            ctx.profile_enable(False)
            key = "multidot" if name == "multidot" else "multiaxpy"
            r = rep[key]
            out[f"{name}_nv{nv}"] = dict(avg_us=round(r["avg_us"], 2), GBps=round(r["gbps"], 1))
    print(json.dumps({"ldv_pad": args.ldv_pad, "blas": out}))


def spmv(args):
    import numpy as np
    import torch
    import nonlinearsolve_jl_amd as nls
    ctx = nls.default_context()
    P = nls.Bratu2D(args.ns, 6.0)
    J = P.jac_csr()
    u = torch.zeros(P.n_local, dtype=torch.float64, device="cuda")
    P.jac_values(u, J)
    x = torch.randn(P.n_local, dtype=torch.float64, device="cuda")
    y = torch.empty_like(x)
    for _ in range(5):
        J.matvec(x, out=y)
    ctx.profile_enable(True)
    for _ in range(args.reps):
        J.matvec(x, out=y)
    r = ctx.profile_report()["spmv"]
    ctx.profile_enable(False)
    # JVP too
    v = torch.randn_like(x)
    for _ in range(3):
        P.jvp(v, u)
    ctx.profile_enable(True)
    for _ in range(args.reps):
        P.jvp(v, u)
    rj = ctx.profile_report()["jvp"]
    print(json.dumps({"spmv": dict(tile=os.environ.get("NK_SPMV_TILE", "2048"), variant=os.environ.get("NK_SPMV_VARIANT", "0"),
                                   ns=args.ns, avg_us=round(r["avg_us"], 2), GBps=round(r["gbps"], 1)),
                      "jvp": dict(avg_us=round(rj["avg_us"], 2), GBps=round(rj["gbps"], 1))}))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=1 << 20)
    ap.add_argument("--ns", type=int, default=1024)
    ap.add_argument("--reps", type=int, default=30)
    ap.add_argument("--mode", default="all")
    ap.add_argument("--ldv-pad", type=int, default=0)
    a = ap.parse_args()
    if a.mode == "blas":
        blas(a)
    elif a.mode == "spmv":
        spmv(a)
    else:
        subprocess.call([sys.executable, __file__, "--mode", "blas", "--n", str(a.n), "--reps", str(a.reps)])
        for tile in (512, 1024, 2048, 4096):
            for var in (0, 1, 2):
                env = dict(os.environ, NK_SPMV_TILE=str(tile), NK_SPMV_VARIANT=str(var))
                subprocess.call([sys.executable, __file__, "--mode", "spmv", "--ns", str(a.ns), "--reps", str(a.reps)], env=env)
        env = dict(os.environ, NK_SPMV_VARIANT="3")
        subprocess.call([sys.executable, __file__, "--mode", "spmv", "--ns", str(a.ns), "--reps", str(a.reps)], env=env)
