#!/usr/bin/env python
"""bench.py — Newton steps/s + SpMV GB/s vs the HBM roofline on 2-D Bratu (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

Workload (config C3 of BASELINE.md, the configuration the metric is quoted on): 2-D Bratu n = 1024²
(N = 1 048 576 unknowns, nnz = 5 238 784, λ = 6, u0 = 0), NewtonRaphson with the *fixed-work* Krylov protocol
of SURVEY.md §8d — exactly 30 Arnoldi steps of GMRES(30) per Newton step, zero initial guess, CGS2
orthogonalisation (in its delayed form `dcgs2`: same arithmetic to rounding, 2 sweeps over the basis per step) — on the assembled CSR Jacobian (values refilled every step, SpMV as the operator).
A "step" is one such Newton step: Jacobian value fill + 30×(SpMV + CGS2 passes) + solution update + u += δu +
residual + ‖·‖∞ + termination bookkeeping, everything resident in HBM.
N > 1: weak scaling — every rank owns ≈1024² unknowns of a (1024·√N)² grid (row-range partition by grid
lines, halo lines by RCCL send/recv, Krylov inner products by RCCL all-reduce). `value` is the whole-job
aggregate: Newton steps/s × (global unknowns / 1024²), i.e. 1024²-unknown step equivalents per second, which
is plain Newton steps/s at N = 1; the raw rate of the global problem is config.global_newton_steps_per_sec.

Extra objects on the JSON line: `roofline` (CSR SpMV kernel, HIP-event timed on the launch stream in a second,
instrumented pass of the same K steps), `kernels` (every kernel family of the step), `cpu_baseline` (the oracle's
C/OpenMP restatement timed on this box's host cores on a bounded sample of the same workload).
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: 8.0 TB/s spec
HBM_ACHIEVABLE_GBS = 6290.0  # measured float4 copy


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--grid", dest="n", type=int, default=1024, help="grid side per GPU-equivalent (1024 ⇒ N = 1e6)")
    ap.add_argument("--ortho", default="dcgs2", choices=["cgs2", "dcgs2", "dcgs2_1r", "cgs", "mgs"])
    ap.add_argument("--arnoldi", type=int, default=30)
    ap.add_argument("--matfree", action="store_true", help="bench the matrix-free JVP operator instead of CSR")
    ap.add_argument("--cpu-steps", type=int, default=12, help="Newton steps of the CPU baseline sample (0 = skip)")
    ap.add_argument("--no-profile-pass", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback)")
    ndev = torch.cuda.device_count()
    dev_index = local_rank % ndev  # BENCH_BACKEND=gloo lets two ranks share one GPU (code-path dry run only)
    torch.cuda.set_device(dev_index)
    backend = os.environ.get("BENCH_BACKEND", "nccl")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend)

    import nonlinearsolve_jl_amd as nls

    ctx = nls.Context(device=dev_index)
    nls.set_default_context(ctx)
    comm = "none"
    if world > 1:
        # rank 0 creates a ncclUniqueId, torch.distributed broadcasts its 128 bytes, and the library then owns
        # its own RCCL communicator on the compute stream (nonlinearsolve.jl_amd/dist.py). If that bootstrap
        # fails the collectives are routed through torch.distributed's RCCL process group instead — same
        # wire, Python in the loop — and the JSON line says so.
        try:
            comm = nls.dist.init_comm(ctx, os.environ.get("NK_COMM", "rccl" if backend == "nccl" else "torch"))
        except Exception as ex:  # noqa: BLE001
            print(f"[bench] direct RCCL bootstrap failed on rank {rank}: {ex}; using torch.distributed callbacks",
                  file=sys.stderr)
            comm = nls.dist.init_comm(ctx, "torch") + "(fallback)"
    # weak scaling: grid side so that every rank owns ≈ n² unknowns; side must be ≥ world lines
    ns = args.n if world == 1 else int(round(args.n * math.sqrt(world)))
    prob = nls.NonlinearProblem(nls.Bratu2D(ns, 6.0))
    n_local = prob.device_problem.n_local
    n_global = prob.device_problem.n_global
    alg = nls.NewtonRaphson(
        linsolve=nls.KrylovJL_GMRES(gmres_restart=args.arnoldi, maxiters=args.arnoldi, ortho=args.ortho,
                                    fixed_iters=args.arnoldi),
        concrete_jac=not args.matfree)
    u0 = torch.zeros(n_local, dtype=torch.float64, device="cuda")
    prob.u0 = u0
    # abstol tiny and maxiters huge: every step does the full fixed work, nothing terminates early
    cache = nls.init(prob, alg, abstol=1e-300, maxiters=10 ** 9)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def run_steps(k):
        for _ in range(k):
            cache.step()

    run_steps(args.warmup)
    barrier()
    t0 = time.perf_counter()
    run_steps(args.steps)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    steps_per_s = args.steps / dt
    # whole-job aggregate: one unit = one Newton step on 1024² unknowns (exactly the N = 1 workload). Under weak
    # scaling the global problem has world×1024² unknowns, so a global Newton step is `world` units.
    units_per_step = n_global / float(1024 * 1024)
    value = steps_per_s * units_per_step
    stats = cache.stats
    fnorm = cache.fnorm_inf

    # ---- second, instrumented pass of the same K steps: HIP events on the launch stream around every
    # kernel family (not part of `value`)
    kernels = {}
    if not args.no_profile_pass:
        ctx.profile_enable(True)
        run_steps(args.steps)
        barrier()
        kernels = ctx.profile_report()
        ctx.profile_enable(False)
    dom = "spmv" if not args.matfree else "jvp"
    roof = None
    if dom in kernels:
        k = kernels[dom]
        kname = "k_spmv_stream" if dom == "spmv" else "k_bratu_jvp"
        # HBM traffic per launch from the rocprofv3 PMC passes committed under profiles/ (FETCH_SIZE/WRITE_SIZE,
        # separate --pmc runs, gfx950 ×2 correction on FETCH_SIZE — tools/pmc_summary.py). bench.py cannot collect
        # PMC counters itself; the same kernel at the same size is a per-launch constant. null if no profile.
        traffic, tsrc = None, None
        try:
            import glob
            cand = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc.json")))
            if cand and world == 1 and ns == 1024:
                pm = json.load(open(cand[-1]))
                for key, v in pm.items():
                    if key.startswith(kname):
                        traffic, tsrc = int(v["hbm_bytes_per_launch"]), os.path.basename(cand[-1])
        except Exception:  # noqa: BLE001
            pass
        roof = {"kernel": kname, "bound": "hbm",
                "achieved": round(k["gbps"], 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(k["gbps"] / HBM_PEAK_GBS, 4),
                "frac_of_achievable_6.29TBs": round(k["gbps"] / HBM_ACHIEVABLE_GBS, 4),
                "traffic": traffic, "traffic_source": tsrc, "launches": k["launches"],
                "avg_us": round(k["avg_us"], 2),
                "timing": "hipExtLaunchKernelGGL start/stop events (kernel begin→end on the launch stream)",
                "algorithmic_bytes_per_launch": int(k["bytes"] / k["launches"])}
    ksum = {name: {"launches": v["launches"], "avg_us": round(v["avg_us"], 2), "GB/s": round(v["gbps"], 1),
                   "frac_of_8TBs": round(v["gbps"] / HBM_PEAK_GBS, 4), "share_of_step_time": None}
            for name, v in kernels.items()}
    tot = sum(v["total_ms"] for v in kernels.values()) or 1.0
    for name, v in kernels.items():
        ksum[name]["share_of_step_time"] = round(v["total_ms"] / tot, 4)

    # ---- CPU baseline: the oracle's C/OpenMP restatement on this box's host cores, bounded sample
    cpu = None
    if rank == 0 and world == 1 and args.cpu_steps > 0:
        import numpy as np
        from oracle import c_oracle as CO
        CO.build()
        cores = CO.num_threads()
        tc = time.perf_counter()
        CO.bratu_newton(ns, 6.0, 0.0, np.zeros(ns * ns), args.cpu_steps, use_csr=not args.matfree, m=args.arnoldi,
                        itmax=args.arnoldi, fixed_iters=args.arnoldi, forcing=False)
        tcpu = time.perf_counter() - tc
        cpu = {"value": round(args.cpu_steps / tcpu, 4), "unit": "newton_steps/s", "cores": cores, "kind": "port",
               "sample": f"{args.cpu_steps} fixed-work Newton steps (30 MGS-GMRES Arnoldi steps each) of the same "
                         f"Bratu {ns}x{ns} workload, oracle/nk_oracle.c with OpenMP on {cores} threads, {tcpu:.1f} s"}

    # ---- time to tolerance (extra, not `value`): the §3 protocol (NewtonRaphson + GMRES(30) + Eisenstat–Walker,
    # inner cap 300, ‖h²F‖∞ ≤ 1e-8) with the Chebyshev(32, ratio 300) right preconditioner, device vs the oracle's
    # C/OpenMP restatement of the same algorithm on the host cores
    ttt = None
    if rank == 0 and world == 1 and args.cpu_steps > 0:
        import numpy as np
        from oracle import c_oracle as CO
        prob2 = nls.NonlinearProblem(nls.Bratu2D(ns, 6.0), u0=torch.zeros(n_local, dtype=torch.float64, device="cuda"))
        alg2 = nls.NewtonRaphson(linsolve=nls.KrylovJL_GMRES(gmres_restart=30, maxiters=300,
                                                             precs=nls.ChebyshevPrecs(32, 300.0)),
                                 forcing=nls.EisenstatWalkerForcing2(), concrete_jac=not args.matfree)
        nls.solve(prob2, alg2, abstol=1e-8, maxiters=50)  # warm-up (allocations, first-touch)
        torch.cuda.synchronize()
        tg = time.perf_counter()
        sol2 = nls.solve(prob2, alg2, abstol=1e-8, maxiters=50)
        torch.cuda.synchronize()
        tg = time.perf_counter() - tg
        tc2 = time.perf_counter()
        uC, fnC, giC = CO.bratu_newton_cheb(ns, 6.0, 0.0, np.zeros(ns * ns), 50, not args.matfree, 30, 300, 32, 300.0, 1e-8)
        tc2 = time.perf_counter() - tc2
        ttt = {"protocol": "NewtonRaphson+GMRES(30)+EisenstatWalkerForcing2+Chebyshev(32,300) to |h^2 F|inf<=1e-8",
               "gpu_seconds": round(tg, 4), "gpu_newton_steps": sol2.stats.nsteps, "gpu_gmres_iters": sol2.stats.gmres_iters,
               "gpu_retcode": sol2.retcode, "cpu_seconds": round(tc2, 3), "cpu_newton_steps": int(len(fnC)),
               "cpu_gmres_iters": int(giC.sum()), "cpu_cores": CO.num_threads(),
               "u_maxdiff_gpu_vs_cpu": float(np.max(np.abs(sol2.u.cpu().numpy() - uC))),
               "speedup": round(tc2 / tg, 1)}
        # the same solve with the built-in geometric multigrid V-cycle behind the `precs` hook (GPU only: the C oracle has
        # no multigrid; its NumPy restatement pins iteration counts at small sizes in the tests)
        alg3 = nls.NewtonRaphson(linsolve=nls.KrylovJL_GMRES(gmres_restart=30, maxiters=300, precs=nls.MultigridPrecs(2, 31)),
                                 forcing=nls.EisenstatWalkerForcing2(), concrete_jac=not args.matfree)
        nls.solve(prob2, alg3, abstol=1e-8, maxiters=50)
        torch.cuda.synchronize()
        tm = time.perf_counter()
        sol3 = nls.solve(prob2, alg3, abstol=1e-8, maxiters=50)
        torch.cuda.synchronize()
        tm = time.perf_counter() - tm
        ttt["multigrid_precs"] = {"gpu_seconds": round(tm, 4), "gpu_newton_steps": sol3.stats.nsteps,
                                  "gpu_gmres_iters": sol3.stats.gmres_iters, "gpu_retcode": sol3.retcode,
                                  "u_maxdiff_vs_chebyshev_run": float((sol3.u - sol2.u).abs().max())}

    line = None
    if rank == 0:
        line = {
            "metric": "newton_steps_per_sec", "value": round(value, 3),
            "unit": "newton_steps/s" if world == 1 else "newton_steps/s x (unknowns / 1024^2)",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * dt / args.steps, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"bratu2d_{ns}x{ns}_newtonraphson_gmres{args.arnoldi}_fixedwork_"
                                   f"{'matfree_jvp' if args.matfree else 'csr_spmv'}",
                       "global_newton_steps_per_sec": round(steps_per_s, 3), "units_per_global_step": round(units_per_step, 4),
                       "unknowns_global": n_global, "unknowns_per_gpu": n_local, "lambda": 6.0,
                       "arnoldi_steps_per_newton_step": args.arnoldi, "ortho": args.ortho,
                       "parallelism": f"row-range x{world}", "comm": comm},
            "roofline": roof, "kernels": ksum, "cpu_baseline": cpu, "time_to_tolerance": ttt,
            "gpu_vs_cpu": round(steps_per_s / cpu["value"], 1) if cpu else None,
            "check": {"fnorm_inf_after_timed_steps": fnorm, "gmres_iters": stats.gmres_iters,
                      "nsteps": stats.nsteps, "allreduces": stats.allreduces},
        }
    # ---- N > 1 on the CSR operator: try the SpMV with its halo exchange overlapped with the interior row blocks
    # (second stream + events, off by default in the library). The measurement above is complete and stays the
    # result unless the overlapped variant, measured by the same protocol, is faster; a watchdog prints the result
    # above and leaves if the attempt does not come back.
    overlap = {"tried": False}
    if world > 1 and not args.matfree and os.environ.get("NK_BENCH_OVERLAP", "auto") != "off":
        import threading
        finished = threading.Event()

        def watchdog():
            if not finished.wait(timeout=float(os.environ.get("NK_BENCH_OVERLAP_TIMEOUT", "90"))):
                if rank == 0:
                    line["config"]["halo_overlap"] = "attempt timed out; serial exchange reported"
                    print(json.dumps(line))
                    sys.stdout.flush()
                os._exit(0)

        threading.Thread(target=watchdog, daemon=True).start()
        try:
            ctx.set_halo_overlap(True)
            run_steps(args.warmup)
            barrier()
            t0 = time.perf_counter()
            run_steps(args.steps)
            barrier()
            dt2 = time.perf_counter() - t0
            t2 = torch.tensor([dt2], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
            dist.all_reduce(t2, op=dist.ReduceOp.MAX)
            dt2 = float(t2.item())
            ok = math.isfinite(cache.fnorm_inf)
            overlap = {"tried": True, "ms_per_step": round(1e3 * dt2 / args.steps, 4), "ok": bool(ok)}
            if rank == 0:
                line["config"]["halo_overlap"] = {"serial_ms_per_step": line["ms_per_step"],
                                                  "overlapped_ms_per_step": overlap["ms_per_step"],
                                                  "reported": "overlapped" if (ok and dt2 < dt) else "serial"}
                if ok and dt2 < dt:
                    sps = args.steps / dt2
                    line["value"] = round(sps * units_per_step, 3)
                    line["ms_per_step"] = overlap["ms_per_step"]
                    line["config"]["global_newton_steps_per_sec"] = round(sps, 3)
        except Exception as ex:  # noqa: BLE001
            if rank == 0:
                line["config"]["halo_overlap"] = f"attempt failed ({ex}); serial exchange reported"
        finished.set()
    if rank == 0:
        print(json.dumps(line))
    cache.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
