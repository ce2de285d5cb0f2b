#!/usr/bin/env python
"""bench.py — Newton steps/s + SpMV GB/s vs the HBM roofline on 2-D Bratu (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W            # N > 1: spawns its own N ranks (one per GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W            # …or is launched as one of them

Workload `c3` (default; config C3 of BASELINE.md, the configuration the metric is quoted on): 2-D Bratu n = 1024²
(N = 1 048 576 unknowns, nnz = 5 238 784, λ = 6, u0 = 0), NewtonRaphson with the *fixed-work* Krylov protocol of
SURVEY.md §8d — exactly 30 Arnoldi steps of GMRES(30) per Newton step (a 30-column orthonormal Krylov basis, the
30-dimensional least-squares problem solved through Givens rotations), zero initial guess — on the assembled CSR Jacobian
(values refilled every step, SpMV as the operator). The Arnoldi process is the LIBRARY DEFAULT: the s-step form with the
automatic block size and basis (`--ortho sstep --sstep 0 --sstep-basis auto`) — two blocks of 15 Newton-basis columns per cycle
(shifts = Leja-ordered Chebyshev points of the Jacobian's Gershgorin interval), 30 SpMVs, per block three sweeps over the basis
with the Gram blocks on the FP64 matrix cores, the last block left at its first pass (csrc/nk_sstep.hip) — the same Krylov space
and the same minimisation as column-by-column CGS2, iterates equal to 1e-10 over 20 Newton steps (tests/test_gpu_sstep.py: this
very protocol at full size against the C oracle); `--ortho dcgs2` runs the column-by-column form of rounds 1–2 (delayed CGS2:
two sweeps and one reduction per column), `--sstep 6 --sstep-basis monomial` round 2's blocks. A "step" is one such Newton step:
Jacobian value fill + 30 SpMVs + the orthogonalisation sweeps + solution update + u += δu + residual + ‖·‖∞ + termination
bookkeeping, everything resident in HBM. Defaults: 300 timed steps after 20 warm-up steps (≈ 0.3 s of GPU time; the CPU leg's
bounded sample and the time-to-tolerance extras dominate the wall clock of the default run).

N > 1 is STRONG scaling on the metric's configuration: the same 1024² problem row-partitioned by grid lines over N
ranks (halo lines + Krylov inner products over xGMI: peer-mapped buffers, RCCL as fallback); `value` is the plain
global Newton steps/s. `--workload c4` is Bratu 4096² (config C4), `--workload c5` the Brusselator 512² TrustRegion
step (config C5) — both row-partitioned over N ranks as well. A weak-scaling run (every rank owns ≈1024² unknowns of a
(1024·√N)² grid) is reported under the extra key `weak_scaling` when N > 1.

Extra objects on the JSON line: `roofline` (the time-dominant kernel family of the step — the CSR SpMV — HIP-event timed on
the launch stream in a second, instrumented pass of the same K steps; `traffic` from the newest PMC summary under profiles/),
`roofline_step` (bytes every kernel of the step must move ÷ ms_per_step), `step_time_stats` (the host's per-step loop times of the
timed region, the time of its closing barrier, and median / p10 / p90 of the per-step device times from one event per step in a
pass of its own — round 6 took the events, which are marker packets between the steps' kernels, and the interpreter's garbage
collector out of the timed region), `kernels` (every kernel family of the step), `cpu_baseline` (the oracle's
tuned C/OpenMP restatements of the same step — delayed CGS2 and the Newton-basis s-step form, median of 5 sustained samples
each at a thread count chosen by sustained samples; value = the faster — next to the box's STREAM triad, its CSR SpMV rate and
a single-thread figure), `config.comm_selfcheck` for N > 1 (tools/multi_gpu_selfcheck.py's verdict on the negotiated transport).
"""
import argparse
import json
import math
import os
import re
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import step_model  # noqa: E402  (tools/step_model.py: the launches of a fixed-work Newton step and their bytes)

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: 8.0 TB/s spec
HBM_ACHIEVABLE_GBS = 6290.0  # measured float4 copy


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="c3", choices=["c3", "c4", "c5"])
    ap.add_argument("--grid", dest="n", type=int, default=0, help="grid side (default: 1024 for c3, 4096 for c4, 512 for c5)")
    ap.add_argument("--ortho", default="sstep", choices=["cgs2", "dcgs2", "dcgs2_1r", "cgs", "mgs", "sstep"])
    ap.add_argument("--sstep", type=int, default=0, help="--ortho sstep: basis columns per block (0 = the library's choice: 15 Newton basis, 6 monomial)")
    ap.add_argument("--sstep-basis", default="auto", choices=["auto", "monomial", "newton"])
    ap.add_argument("--arnoldi", type=int, default=30)
    ap.add_argument("--matfree", action="store_true", help="bench the matrix-free JVP operator instead of CSR")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the CPU baseline sample (0 = skip)")
    ap.add_argument("--no-profile-pass", action="store_true")
    ap.add_argument("--no-weak", action="store_true", help="N > 1: skip the extra weak-scaling run")
    ap.add_argument("--no-ttt", action="store_true", help="skip the time-to-tolerance extras")
    ap.add_argument("--no-spmv-hbm", action="store_true", help="skip the HBM-resident SpMV measurement (Bratu 4096², ≈ 10 s)")
    ap.add_argument("--pmc", default="auto", choices=["auto", "off"],
                    help="auto: at N = 1 measure roofline.traffic in this run (two short rocprofv3 --pmc child passes of this "
                         "script: FETCH_SIZE, WRITE_SIZE); falls back to the newest PMC summary under profiles/")
    return ap.parse_args()


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: become the launcher (one rank per GPU on this node)."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


STEP_STATS = {}


def live_pmc_traffic(args, kname):
    """HBM bytes per launch of kernel `kname` from the PMC counters, measured in THIS run: two short child passes of this script
    under `rocprofv3 --kernel-trace --pmc <counter>` (FETCH_SIZE and WRITE_SIZE in passes of their own, kernel trace only — the
    collection MI355X_MICROARCH.md prescribes), traffic = (2·FETCH_SIZE + WRITE_SIZE)·1024: both counters are in KiB and on gfx950
    FETCH_SIZE tallies 64 B per 128-B request of a coalesced stream. Returns (None, None) when rocprofv3 is missing or a pass
    fails or times out — the caller then reads the newest committed summary."""
    import csv
    import glob
    import shutil
    import tempfile
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if exe is None:
        return None, None
    child = [sys.executable, os.path.abspath(__file__), "--steps", "6", "--warmup", "2", "--cpu-seconds", "0", "--no-profile-pass",
             "--no-ttt", "--no-spmv-hbm", "--pmc", "off", "--workload", args.workload, "--ortho", args.ortho, "--sstep", str(args.sstep),
             "--sstep-basis", args.sstep_basis, "--arnoldi", str(args.arnoldi)]
    if args.n:
        child += ["--grid", str(args.n)]
    if args.matfree:
        child += ["--matfree"]
    env = dict(os.environ, TMPDIR="/tmp", BENCH_PMC_CHILD="1")
    vals = {}
    t0 = time.perf_counter()
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = tempfile.mkdtemp(prefix="nk_pmc_", dir="/tmp")
            try:
                subprocess.run([exe, "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", d, "-o", "p", "--"] + child,
                               cwd="/tmp", env=env, timeout=120, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=False)
                acc = []
                for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                    for r in csv.DictReader(open(f)):
                        if r.get("Counter_Name") == counter and re.match(kname, r.get("Kernel_Name", "").replace("void ", "")):
                            acc.append(float(r["Counter_Value"]))
                if not acc:
                    return None, None
                vals[counter] = sum(acc) / len(acc)
            finally:
                shutil.rmtree(d, ignore_errors=True)
    except Exception:  # noqa: BLE001
        return None, None
    traffic = int((2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024)
    return traffic, (f"measured in this run: rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes of 8 steps, "
                     f"{time.perf_counter() - t0:.0f} s), (2·F + W)·1024")


def timed_steps(cache, steps, barrier, dist, world, backend, torch):
    """EXACTLY `steps` steps between two barriers + synchronisations (the contract's clock), max over the ranks. Nothing but the
    steps is inside the timed region — no event records (marker packets between the steps' kernels: they moved to
    step_time_distribution in round 6), and not the interpreter's cyclic garbage collector either: a full collection of the
    torch / scipy object graph takes ≈ 46 ms and, once the events were gone, fired inside the closing barrier of every run
    (profiles/r06_c_host_gaps_head_and_powers_handoff.md §3)."""
    import gc
    host = []
    gc.collect()
    gc.disable()
    barrier()
    t0 = time.perf_counter()
    for i in range(steps):
        cache.step()
        host.append(time.perf_counter())
    t_loop = time.perf_counter()
    barrier()
    dt = time.perf_counter() - t0
    gc.enable()
    STEP_STATS.update(closing_barrier_ms=round((t0 + dt - t_loop) * 1e3, 3))
    hs = sorted(b - a for a, b in zip([t0] + host[:-1], host))
    if hs:
        STEP_STATS.update(host_median_ms=round(hs[len(hs) // 2] * 1e3, 4), host_max_ms=round(hs[-1] * 1e3, 4),
                          host_p99_ms=round(hs[min(len(hs) - 1, int(0.99 * len(hs)))] * 1e3, 4), host_mean_ms=round(sum(hs) / len(hs) * 1e3, 4))
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt


def step_time_distribution(cache, steps, barrier, torch):
    """A pass of its own, NOT the contract's clock: an event recorded on the library's stream after every step (no
    synchronisation — but every record is a marker packet between two steps' kernels, a few µs of idle device each, which is why
    round 6 took the events out of the timed region): median, p10, p90 of the per-step times."""
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    barrier()
    evs[0].record()
    for i in range(steps):
        cache.step()
        evs[i + 1].record()
    barrier()
    per = sorted(evs[i].elapsed_time(evs[i + 1]) for i in range(steps))
    if per:
        q = lambda f: per[min(len(per) - 1, int(f * len(per)))]  # noqa: E731
        STEP_STATS.update(n=steps, mean_ms=round(sum(per) / len(per), 4), median_ms=round(q(0.5), 4), p10_ms=round(q(0.1), 4), p90_ms=round(q(0.9), 4),
                          min_ms=round(per[0], 4), max_ms=round(per[-1], 4), measured="in a pass of its own (one event per step)")


def bytes_per_step(args, n, nnz, newton_basis, resident_powers):
    """(HBM bytes the launched kernels of one fixed-work Newton step must move, the same with every operator application
    charged at SURVEY.md §8(d)'s CSR-SpMV figure) — tools/step_model.py lists the launches one by one; the column-by-column
    forms keep the closed formula of DESIGN.md §5"""
    if args.ortho == "sstep":
        s = args.sstep or (15 if newton_basis else 6)
        return step_model.step_bytes(n, nnz, arnoldi=args.arnoldi, s=s, matfree=args.matfree, resident_powers=resident_powers,
                                     newton_basis=newton_basis, implicit=newton_basis and os.environ.get("NK_SS_IMPLICIT", "1") != "0",
                                     fused_tail=os.environ.get("NK_FUSED_UPDATE", "1") != "0" and
                                     os.environ.get("NK_FUSED_RESIDUAL_NORMS", "1") != "0" and
                                     os.environ.get("NK_PRELOADED_RHS", "1") != "0")
    b_op = 24.0 * n if args.matfree else step_model.spmv_bytes(n, nnz)
    m = args.arnoldi
    krylov = sum(b_op + 8.0 * n * (k + 2) + 8.0 * n * (k + 4) for k in range(m))    # delayed CGS2: two sweeps per column
    once = 8.0 * n * (m + 2) + (0.0 if args.matfree else 8.0 * nnz + 8.0 * n) + 16.0 * n + 24.0 * n + 16.0 * n + 8.0 * n
    return krylov + once, krylov + once


def spmv_hbm_resident(nls, torch, ctx, reps=30):
    """The streaming CSR SpMV where NOTHING is cache resident: Bratu 4096² (1.07 GB of matrix against the 256 MiB Infinity
    Cache), the kernel's own begin→end timestamps, median of `reps` launches — the HBM-level figure next to the 1024² ones."""
    ns = 4096
    P = nls.Bratu2D(ns, 6.0)
    n = ns * ns
    u = torch.zeros(n, dtype=torch.float64, device="cuda")
    v = torch.randn(n, dtype=torch.float64, device="cuda")
    J = P.jac_csr()
    P.jac_values(u, J)
    y = torch.empty_like(v)
    ts = []
    by = 0.0
    for i in range(5 + reps):
        ctx.profile_enable(True)
        J.matvec(v, out=y)
        r = ctx.profile_report()["spmv"]
        by = r["bytes"] / r["launches"]
        if i >= 5:
            ts.append(r["avg_us"])
    ctx.profile_enable(False)
    ts.sort()
    med = ts[len(ts) // 2]
    del J, P, u, v, y
    torch.cuda.empty_cache()
    return {"kernel": "k_spmv_stream", "grid": ns, "algorithmic_bytes_per_launch": int(by), "median_us": round(med, 2),
            "achieved": round(by / med / 1e3, 1), "unit": "GB/s", "frac": round(by / med / 1e3 / HBM_PEAK_GBS, 4),
            "frac_of_achievable_6.29TBs": round(by / med / 1e3 / HBM_ACHIEVABLE_GBS, 4), "launches": reps,
            "note": "Bratu 4096² Jacobian (1.07 GB): nothing fits the Infinity Cache — the streaming kernel against HBM itself"}


def cpu_baseline(ns, arnoldi, matfree, budget_s):
    """The oracle's tuned CPU leg on this box's host cores (rank 0, N = 1 only): bounded sample of the same workload, in a
    child process so that the OpenMP runtime starts bound to the cores and waits actively (oracle/cpu_leg.py)."""
    env = dict(os.environ)
    # threads spread over the cores of the affinity mask; the leg itself picks the thread count (all hardware threads, half,
    # a quarter) by the STREAM triad it measures — SMT siblings and container CPU quotas make "all of them" a bad default
    env.pop("OMP_NUM_THREADS", None)
    env.update(OMP_PLACES="cores", OMP_PROC_BIND="spread")
    env.pop("OMP_WAIT_POLICY", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "cpu_leg.py"), str(ns), str(arnoldi),
                          str(int(bool(matfree))), str(budget_s)], env=env, capture_output=True, text=True,
                         timeout=20 * budget_s + 120)
    if out.returncode != 0:
        raise RuntimeError(out.stderr[-400:])
    return json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])


def main():
    args = parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args))

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback)")
    ndev = torch.cuda.device_count()
    backend = os.environ.get("BENCH_BACKEND", "nccl")
    if world > ndev and backend == "nccl":
        raise SystemExit(f"--gpus {world} needs {world} GPUs, this node has {ndev} "
                         "(BENCH_BACKEND=gloo lets ranks share a GPU for a code-path dry run)")
    dev_index = local_rank % ndev
    torch.cuda.set_device(dev_index)
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
    selfchecks = None
    if world > 1:
        # Transport of the library's small collectives, in order of preference: peer-mapped buffers over xGMI
        # (hipIpc; "peer"), the library's own RCCL communicator ("rccl"), torch.distributed callbacks ("torch").
        want = os.environ.get("NK_COMM", "peer" if backend == "nccl" else "torch")
        order = [want] + [t for t in ("rccl", "torch") if t != want] if backend == "nccl" else [want, "torch"]
        for tr in order:
            try:
                comm = nls.dist.init_comm(ctx, tr)
                break
            except Exception as ex:  # noqa: BLE001
                print(f"[bench] transport {tr!r} failed on rank {rank}: {ex}", file=sys.stderr)
        if comm == "none":
            raise SystemExit("no communicator could be initialised")
        # known-answer self-check of that transport before anything is timed (tools/multi_gpu_selfcheck.py): all-reduces, the
        # halo exchange inside the SpMV, three Newton steps against one rank. A failing peer path is switched off on ALL ranks
        # (the base transport then serves every collective) and checked again; a failing base transport ends the run.
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        from multi_gpu_selfcheck import selfcheck
        selfchecks = []
        try:
            sc = selfcheck(nls, ctx, torch, dist, comm)
        except Exception as ex:  # noqa: BLE001
            sc = {"transport": comm, "ok": False, "error": str(ex)}
        selfchecks.append(sc)
        if not sc["ok"] and comm.startswith("peer+"):
            ctx.comm_peer_disable()
            comm = comm[len("peer+"):] + "(peer path failed its self-check)"
            sc = selfcheck(nls, ctx, torch, dist, comm)
            selfchecks.append(sc)
        if not selfchecks[-1]["ok"]:
            raise SystemExit(f"multi-GPU self-check failed on transport {comm!r}: {selfchecks}")

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def make_cache(kind, ns):
        if kind == "c5":
            PB = nls.Brusselator2D(ns)
            prob = nls.NonlinearProblem(PB, u0=PB.initial_guess(device=True))
            alg = nls.TrustRegion(linsolve=nls.KrylovJL_GMRES(gmres_restart=args.arnoldi, maxiters=args.arnoldi, ortho=args.ortho, sstep=args.sstep, sstep_basis=args.sstep_basis,
                                                              fixed_iters=args.arnoldi), concrete_jac=not args.matfree)
        else:
            prob = nls.NonlinearProblem(nls.Bratu2D(ns, 6.0))
            prob.u0 = torch.zeros(prob.device_problem.n_local, dtype=torch.float64, device="cuda")
            alg = nls.NewtonRaphson(linsolve=nls.KrylovJL_GMRES(gmres_restart=args.arnoldi, maxiters=args.arnoldi, ortho=args.ortho, sstep=args.sstep, sstep_basis=args.sstep_basis,
                                                                fixed_iters=args.arnoldi), concrete_jac=not args.matfree)
        # abstol tiny and maxiters huge: every step does the full fixed work, nothing terminates early
        return prob, nls.init(prob, alg, abstol=1e-300, maxiters=10 ** 9)

    ns = args.n or {"c3": 1024, "c4": 4096, "c5": 512}[args.workload]
    prob, cache = make_cache(args.workload, ns)
    n_local, n_global = prob.device_problem.n_local, prob.device_problem.n_global
    for _ in range(args.warmup):
        cache.step()
    dt = timed_steps(cache, args.steps, barrier, dist, world, backend, torch)
    steps_per_s = args.steps / dt
    stats = cache.stats
    fnorm = cache.fnorm_inf
    step_time_distribution(cache, min(args.steps, 200), barrier, torch)

    # ---- second, instrumented pass of the same K steps: HIP events on the launch stream around every
    # kernel family (not part of `value`)
    kernels = {}
    if not args.no_profile_pass:
        ctx.profile_enable(True)
        for _ in range(args.steps):
            cache.step()
        barrier()
        kernels = ctx.profile_report()
        ctx.profile_enable(False)
    step_stats = dict(STEP_STATS)
    # ---- `roofline`: the TIME-DOMINANT kernel family of the step (with the matrix-powers kernel holding the matrix on the chip
    # that is no longer the SpMV but the s-step process's Gram sweeps). achieved = algorithmic bytes ÷ HIP-event time of the
    # family's launches; traffic = HBM bytes per launch from the PMC counters, measured in this run where possible.
    FAMILY = {"spmv": ("k_spmv_stream", r"k_spmv_stream<"),
              "spmv_powers": ("k_spmv_powers", r"k_spmv_powers<"),
              "jvp": ("k_bratu_jvp", r"k_bratu_jvp"),
              "multidot": ("k_ss_block<S, false, true, …> + k_ss_block_mm<S, k> / k_ss_block_ro<S, k> (s-step sweeps A and B: Gram blocks — and B's update — on the FP64 matrix cores)"
                           if args.ortho == "sstep" else "k_dcgs2r_dots", r"(k_ss_block<\d+, (true|false), true|k_ss_block_mm<|k_ss_block_ro<)" if args.ortho == "sstep" else r"k_dcgs2r_dots"),
              "multiaxpy": ("k_ss_block<S, true, false, …> + k_multiaxpy (sweep C, x = V y)" if args.ortho == "sstep" else "k_dcgs2r_axpy_tail",
                            r"(k_ss_block<\d+, true, false|k_multiaxpy)" if args.ortho == "sstep" else r"k_dcgs2r_axpy")}
    roof = None
    heavy = {k: v for k, v in kernels.items() if k in FAMILY}
    dom = max(heavy, key=lambda k: heavy[k]["total_ms"]) if heavy else None
    under_profiler = "rocprofiler" in os.environ.get("LD_PRELOAD", "") or any(k.startswith("ROCPROF") for k in os.environ)
    can_pmc = args.pmc == "auto" and world == 1 and not os.environ.get("BENCH_PMC_CHILD") and not under_profiler

    def family_roofline(fam):
        k = kernels[fam]
        kname, pattern = FAMILY[fam]
        traffic, tsrc = (live_pmc_traffic(args, pattern) if can_pmc else (None, None))
        return {"kernel": kname, "family": fam, "bound": "hbm",
                "achieved": round(k["gbps"], 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(k["gbps"] / HBM_PEAK_GBS, 4),
                "frac_of_achievable_6.29TBs": round(k["gbps"] / HBM_ACHIEVABLE_GBS, 4),
                "traffic": traffic, "traffic_source": tsrc, "launches": k["launches"],
                "avg_us": round(k["avg_us"], 2), "share_of_step_time": None,
                "timing": "hipExtLaunchKernelGGL start/stop events (kernel begin→end on the launch stream)",
                "algorithmic_bytes_per_launch": int(k["bytes"] / k["launches"])}

    if dom:
        roof = family_roofline(dom)
    # ---- the operator itself (the metric's second half: SpMV GB/s against the HBM roofline), whatever dominates the step
    roof_spmv = None
    if rank == 0 and world == 1 and args.workload == "c3" and not args.matfree:
        roof_spmv = {}
        if "spmv_powers" in kernels:
            k = kernels["spmv_powers"]
            per = int(round(k["bytes"] / k["launches"] / step_model.spmv_bytes(n_global, 5 * n_global - 4 * ns)))
            roof_spmv["resident_matrix_powers"] = {
                "kernel": "k_spmv_powers", "applications_per_launch": per, "avg_us": round(k["avg_us"], 2),
                "us_per_application": round(k["avg_us"] / max(per, 1), 2),
                "achieved": round(k["gbps"], 1), "unit": "GB/s",
                # NOT a roofline fraction: algorithmic CSR bytes (80 N per application) ÷ time is above the HBM
                # peak because the matrix is read ONCE per launch and held in the vector registers (csrc/nk_powers.hip)
                "frac_of_algorithmic": round(k["gbps"] / HBM_PEAK_GBS, 4),
                # … the bytes the launch must move (matrix once + 1 column in + `per` columns out) ÷ time ÷ 8 TB/s
                "hbm_bytes_per_launch_model": int(12.0 * (5 * n_global - 4 * ns) + 4.0 * (n_global + 1) + 8.0 * n_global * (per + 1)),
                "frac_hbm_model": round((12.0 * (5 * n_global - 4 * ns) + 4.0 * (n_global + 1) + 8.0 * n_global * (per + 1))
                                        / (k["avg_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                "note": "latency-bound (one band hand-off per application); the HBM-level figure of the streaming kernel is "
                        "`streaming_hbm_resident`"}
        if "spmv" in kernels:
            roof_spmv["streaming_in_solver"] = {k_: v_ for k_, v_ in family_roofline("spmv").items() if k_ != "share_of_step_time"}
        if not args.no_spmv_hbm and not os.environ.get("BENCH_PMC_CHILD"):
            try:
                roof_spmv["streaming_hbm_resident"] = spmv_hbm_resident(nls, torch, ctx)
            except Exception as ex:  # noqa: BLE001
                roof_spmv["streaming_hbm_resident"] = {"error": str(ex)}
    ksum = {name: {"launches": v["launches"], "avg_us": round(v["avg_us"], 2), "GB/s": round(v["gbps"], 1),
                   "frac_of_8TBs": round(v["gbps"] / HBM_PEAK_GBS, 4), "share_of_step_time": None}
            for name, v in kernels.items()}
    tot = sum(v["total_ms"] for v in kernels.values()) or 1.0
    for name, v in kernels.items():
        ksum[name]["share_of_step_time"] = round(v["total_ms"] / tot, 4)
    if roof is not None:
        roof["share_of_step_time"] = ksum[roof["family"]]["share_of_step_time"]

    # ---- N > 1 on the CSR operator with the RCCL transport: try the SpMV with its halo exchange overlapped with the
    # interior row blocks (second stream + events). Reported only if faster, under a watchdog.
    overlap = None
    if world > 1 and not args.matfree and comm.startswith("rccl") and os.environ.get("NK_BENCH_OVERLAP", "auto") != "off":
        import threading
        finished = threading.Event()
        partial = {"dt": dt}

        def watchdog():
            if not finished.wait(timeout=float(os.environ.get("NK_BENCH_OVERLAP_TIMEOUT", "90"))):
                print(f"[bench] rank {rank}: halo-overlap attempt timed out; leaving", file=sys.stderr)
                os._exit(3)

        threading.Thread(target=watchdog, daemon=True).start()
        try:
            ctx.set_halo_overlap(True)
            for _ in range(args.warmup):
                cache.step()
            dt2 = timed_steps(cache, args.steps, barrier, dist, world, backend, torch)
            ok = math.isfinite(cache.fnorm_inf)
            overlap = {"serial_ms_per_step": round(1e3 * dt / args.steps, 4), "overlapped_ms_per_step": round(1e3 * dt2 / args.steps, 4),
                       "reported": "overlapped" if (ok and dt2 < dt) else "serial"}
            if ok and dt2 < dt:
                dt, steps_per_s = dt2, args.steps / dt2
            else:
                ctx.set_halo_overlap(False)
        except Exception as ex:  # noqa: BLE001
            overlap = f"attempt failed ({ex}); serial exchange reported"
        finished.set()
        del partial

    # ---- N > 1: the weak-scaling run as an extra key (every rank owns ≈ ns² unknowns of a (ns·√N)² grid)
    weak = None
    if world > 1 and not args.no_weak and args.workload == "c3":
        nsw = int(round(ns * math.sqrt(world)))
        cache.close()
        probw, cachew = make_cache("c3", nsw)
        for _ in range(args.warmup):
            cachew.step()
        dtw = timed_steps(cachew, args.steps, barrier, dist, world, backend, torch)
        weak = {"grid": nsw, "unknowns_global": probw.device_problem.n_global, "unknowns_per_gpu": probw.device_problem.n_local,
                "global_newton_steps_per_sec": round(args.steps / dtw, 3), "ms_per_step": round(1e3 * dtw / args.steps, 4),
                "value_in_1024sq_step_equivalents": round(args.steps / dtw * probw.device_problem.n_global / float(1024 * 1024), 3)}
        cachew.close()
        cache = None

    # ---- CPU baseline (rank 0, N = 1): the oracle's tuned C/OpenMP leg on this box's host cores, bounded sample
    cpu = None
    if rank == 0 and world == 1 and args.cpu_seconds > 0 and args.workload != "c5":
        try:
            cpu = cpu_baseline(ns, args.arnoldi, args.matfree, args.cpu_seconds if ns <= 1024 else 2.5 * args.cpu_seconds)
        except Exception as ex:  # noqa: BLE001
            cpu = {"error": str(ex)}

    # ---- time to tolerance (extra, not `value`): the §3 protocol (NewtonRaphson + GMRES(30) + Eisenstat–Walker,
    # inner cap 300, ‖h²F‖∞ ≤ 1e-8) with the built-in right preconditioners behind the `precs` hook
    ttt = None
    if rank == 0 and world == 1 and not args.no_ttt and args.workload == "c3":
        prob2 = nls.NonlinearProblem(nls.Bratu2D(ns, 6.0), u0=torch.zeros(n_local, dtype=torch.float64, device="cuda"))
        ttt = {"protocol": "NewtonRaphson+GMRES(30)+EisenstatWalkerForcing2 to |h^2 F|inf<=1e-8"}
        for name, precs in (("chebyshev_32_300", nls.ChebyshevPrecs(32, 300.0)), ("multigrid_2_31", nls.MultigridPrecs(2, 31)),
                            ("amg_from_the_csr_matrix_alone", nls.ObjectPrecs("amg", "left"))):
            if args.matfree and isinstance(precs, nls.ObjectPrecs):
                continue   # (a preconditioner OBJECT is built from a concrete J)
            alg2 = nls.NewtonRaphson(linsolve=nls.KrylovJL_GMRES(gmres_restart=30, maxiters=300, precs=precs),
                                     forcing=nls.EisenstatWalkerForcing2(), concrete_jac=not args.matfree)
            nls.solve(prob2, alg2, abstol=1e-8, maxiters=50)  # warm-up (allocations, first-touch)
            torch.cuda.synchronize()
            tg = time.perf_counter()
            sol2 = nls.solve(prob2, alg2, abstol=1e-8, maxiters=50)
            torch.cuda.synchronize()
            tg = time.perf_counter() - tg
            ttt[name] = {"gpu_seconds": round(tg, 4), "gpu_newton_steps": sol2.stats.nsteps,
                         "gpu_gmres_iters": sol2.stats.gmres_iters, "gpu_retcode": sol2.retcode}

    # ---- two more whole solves for the record (extras, not `value`): config C2 on the direct linsolve (block cyclic reduction
    # on FP64 MFMA) and config C5 with the two-species multigrid V-cycle behind `precs`
    if ttt is not None:
        try:
            def timed(prob_fn, alg, **kw):
                nls.solve(prob_fn(), alg, **kw)          # warm-up
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                sol_ = nls.solve(prob_fn(), alg, **kw)
                torch.cuda.synchronize()
                return time.perf_counter() - t0, sol_
            tg, s2 = timed(lambda: nls.NonlinearProblem(nls.Bratu2D(256, 6.0)), nls.NewtonRaphson(), abstol=1e-8, maxiters=50)
            ttt["c2_bratu256_newton_direct"] = {"gpu_seconds": round(tg, 4), "gpu_newton_steps": s2.stats.nsteps,
                                                "nfactors": s2.stats.nfactors, "gpu_retcode": s2.retcode}
            PB5 = nls.Brusselator2D(512)
            alg5 = nls.TrustRegion(linsolve=nls.KrylovJL_GMRES(gmres_restart=30, maxiters=300, reltol=1e-9, abstol=0.0,
                                                               precs=nls.MultigridPrecs(2, 16)), concrete_jac=True)
            tg, s5 = timed(lambda: nls.NonlinearProblem(PB5, u0=PB5.initial_guess(device=True)), alg5, abstol=1e-7, maxiters=30)
            ttt["c5_brusselator512_trustregion_multigrid"] = {"gpu_seconds": round(tg, 4), "gpu_newton_steps": s5.stats.nsteps,
                                                              "gpu_gmres_iters": s5.stats.gmres_iters, "gpu_retcode": s5.retcode,
                                                              "note": "includes init (pattern, hierarchy); |F|inf <= 1e-7"}
        except Exception as ex:  # noqa: BLE001
            ttt["extras_error"] = str(ex)

    roof_step = None
    if args.workload != "c5":
        nnz_l = 5 * n_global - 4 * ns
        newton_basis = args.ortho == "sstep" and args.sstep_basis != "monomial"
        hbm_b, alg_b = bytes_per_step(args, n_global, nnz_l, newton_basis, "spmv_powers" in kernels)
        sec = dt / args.steps
        gbs, gbs_alg = hbm_b / sec * 1e-9, alg_b / sec * 1e-9
        fam = max(ksum, key=lambda kname: ksum[kname]["share_of_step_time"] or 0.0) if ksum else None
        roof_step = {"bound": "hbm", "hbm_bytes_per_step": int(hbm_b), "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS * world,
                     "unit": "GB/s", "frac": round(gbs / (HBM_PEAK_GBS * world), 4),
                     "frac_of_achievable_6.29TBs": round(gbs / (HBM_ACHIEVABLE_GBS * world), 4),
                     "floor_ms_at_6.29TBs": round(hbm_b / (HBM_ACHIEVABLE_GBS * world * 1e9) * 1e3, 4),
                     "algorithmic_bytes_per_step": int(alg_b), "achieved_algorithmic": round(gbs_alg, 1),
                     "frac_algorithmic": round(gbs_alg / (HBM_PEAK_GBS * world), 4),
                     "time_dominant_family": fam, "time_dominant_share": ksum[fam]["share_of_step_time"] if fam else None,
                     "note": "whole step. hbm_bytes_per_step: what the LAUNCHED kernels must move between HBM and the chip "
                             "(tools/step_model.py lists them; the resident matrix-powers kernel reads the matrix once per block) ÷ "
                             "ms_per_step → frac. algorithmic_bytes_per_step: the same with every operator application charged at "
                             "the CSR SpMV's 80 N bytes (SURVEY.md §8d) → frac_algorithmic, which on-chip reuse can push past 1"}
    if rank == 0:
        op = "matfree_jvp" if args.matfree else "csr_spmv"
        if args.workload == "c5":
            wl = f"brusselator2d_{ns}x{ns}_trustregion_gmres{args.arnoldi}_fixedwork_{op}"
        else:
            wl = f"bratu2d_{ns}x{ns}_newtonraphson_gmres{args.arnoldi}_fixedwork_{op}"
        if world > 1:
            wl += f"_x{world}"
        line = {
            "metric": "newton_steps_per_sec", "value": round(steps_per_s, 3), "unit": "newton_steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * dt / args.steps, 4), "higher_is_better": True,
            # N > 1 partitions the SAME 1024² system over the ranks (total work fixed): strong scaling, also as the N = 1 point
            "scaling": "strong",
            "step_time_stats": step_stats,
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": wl, "unknowns_global": n_global, "unknowns_per_gpu": n_local, "lambda": 6.0,
                       "arnoldi_steps_per_newton_step": args.arnoldi, "ortho": args.ortho if args.ortho != "sstep" else "sstep_%s_%s" % (args.sstep or "auto", args.sstep_basis),
                       "parallelism": f"row-range x{world}", "comm": comm, "comm_selfcheck": selfchecks, "halo_overlap": overlap,
                       # several ranks on ONE device (a code-path run, not a rate: DESIGN §7)
                       "ranks_share_a_device": bool(ctx.comm_device_shared()) if world > 1 else False},
            "roofline": roof, "roofline_spmv": roof_spmv, "roofline_step": roof_step, "kernels": ksum, "cpu_baseline": cpu, "time_to_tolerance": ttt, "weak_scaling": weak,
            # against the best SUSTAINED figure the CPU leg produced (the median of its samples or a validation run; scan samples
            # can be bursts a container's CPU quota does not sustain and are reported, not used)
            "gpu_vs_cpu": round(steps_per_s / max([cpu["value"]] + list(cpu.get("thread_count_validation_steps_per_s", {}).values())), 1)
            if cpu and "value" in cpu else None,
            # … and against the best figure of ANY kind the CPU leg produced (short scan samples included): the denominator rounds 1–3 used
            "gpu_vs_cpu_best_scan": round(steps_per_s / max([cpu["value"], cpu.get("thread_scan_best", 0.0)]
                                                            + list(cpu.get("thread_count_validation_steps_per_s", {}).values())), 1)
            if cpu and "value" in cpu else None,
            "check": {"fnorm_inf_after_timed_steps": fnorm, "gmres_iters": stats.gmres_iters,
                      "nsteps": stats.nsteps, "allreduces": stats.allreduces, "halo_exchanges": stats.halo_exchanges},
        }
        print(json.dumps(line))
        sys.stdout.flush()
    if cache is not None:
        cache.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
