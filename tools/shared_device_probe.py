"""Discriminating experiments for the round-5 finding "two ranks as two PROCESSES on ONE GPU sometimes give wrong results"
(profiles/r05_m_shared_device.txt). Every mode repeats a deterministic piece of work — `--steps` fixed-work
Newton steps of Bratu `--grid`² from u = 0 — `--trials` times inside ONE set of processes (a `reinit!` between trials) and prints
the histogram of ‖F‖∞ after the steps as hex floats: a deterministic path gives ONE value. Modes:

  ranks      N ranks as N processes on cuda:0 (gloo rendezvous; --transport torch | peer)          — the failing set-up
  threads    N ranks as N THREADS of one process, one context + stream each, in-process callbacks    — no second process, no
             cross-process scheduling: wrong here = a race in the product
  solo       one rank, optionally beside a competitor PROCESS on the same GPU:
               --competitor none | stream (torch elementwise: no LDS, no matrix cores) | gemm (rocBLAS FP64: LDS + MFMA + AGPRs)
                            | solver (a second copy of this workload)
Environment switches (NK_*) are passed through; `--env K=V` sets one for the workers only.
Output: one JSON line per mode with the histogram, the all-reduce counts and the trials that raised / hung."""
import argparse
import ctypes as C
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _make(nls, torch, ctx, grid, args):
    P = nls.Bratu2D(grid, 6.0, ctx=ctx)
    u0 = torch.zeros(P.n_local, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    prob = nls.NonlinearProblem(P, u0=u0, ctx=ctx)
    alg = nls.NewtonRaphson(linsolve=nls.KrylovJL_GMRES(gmres_restart=30, maxiters=30, ortho=args.ortho, sstep=args.sstep,
                                                        sstep_basis=args.sstep_basis, fixed_iters=30), concrete_jac=not args.matfree)
    return prob, u0, nls.init(prob, alg, abstol=1e-300, maxiters=10 ** 9)


def _trials(nls, torch, ctx, args, tag, sync=None):
    import faulthandler
    faulthandler.enable()
    prob, u0, cache = _make(nls, torch, ctx, args.grid, args)
    vals, ars, errs = [], [], []
    audit_ref = []
    if args.audit:
        from nonlinearsolve_jl_amd import _lib as L
        L.lib().nk_debug_audit_enable.restype = C.c_int
        L.lib().nk_debug_audit_enable.argtypes = [C.c_void_p, C.c_int]
        assert L.lib().nk_debug_audit_enable(ctx._h, 1) == 0
    for tr in range(args.trials):
        try:
            if sync is not None:
                sync()
            faulthandler.dump_traceback_later(args.stall_dump, exit=False)   # a trial that stalls prints where every thread is
            cache.reinit(u0=u0)
            a0 = int(cache.stats.allreduces)
            trace = []
            for st_i in range(args.steps):
                cache.step()
                if args.detail:
                    trace.append(float(cache.fnorm_inf).hex())
                if args.audit:
                    log = _audit_fetch(ctx)
                    if tr == 0:
                        audit_ref.append(log)
                    elif [e for e in log if e[0] % 10 != 6] != [e for e in audit_ref[st_i] if e[0] % 10 != 6]:   # (…6: the control block, which carries stale fields)
                        ref = audit_ref[st_i]
                        first = next((i for i in range(min(len(log), len(ref))) if log[i] != ref[i] and log[i][0] % 10 != 6), min(len(log), len(ref)))
                        print(f"[probe {tag}] AUDIT trial {tr} step {st_i}: {len(log)} entries (reference {len(ref)}); first difference at "
                              f"entry {first}: {log[first] if first < len(log) else None} vs {ref[first] if first < len(ref) else None}; "
                              f"tags around: {[t for t, _ in log[max(0, first - 3):first + 4]]}", file=sys.stderr, flush=True)
                        audit_ref[st_i] = log   # (the path may have changed for good: a block-size cap after a breakdown)
            faulthandler.cancel_dump_traceback_later()
            vals.append(float(cache.fnorm_inf).hex())
            ars.append(int(cache.stats.allreduces) - a0)
            if args.detail:
                print(f"[probe {tag}] trial {tr}: ar {ars[-1]} " + " ".join(trace), file=sys.stderr, flush=True)
        except Exception as ex:  # noqa: BLE001
            errs.append((tr, str(ex)[:200]))
            break
    st = None
    try:
        st = nls.gmres_sstep_state(cache) if hasattr(nls, "gmres_sstep_state") else None
    except Exception:  # noqa: BLE001
        pass
    cache.close()
    return {"tag": tag, "values": vals, "allreduces": ars, "errors": errs, "sstep_state": st}


def _audit_fetch(ctx):
    from nonlinearsolve_jl_amd import _lib as L
    f = L.lib().nk_debug_audit_fetch
    f.restype = C.c_int
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int)]
    cap = 1 << 16
    tags = (C.c_int * cap)()
    hs = (C.c_ulonglong * cap)()
    n = C.c_int(0)
    assert f(ctx._h, tags, hs, cap, C.byref(n)) == 0
    return [(int(tags[i]), int(hs[i])) for i in range(n.value)]


def _summary(res):
    hist = {}
    for v in res["values"]:
        hist[v] = hist.get(v, 0) + 1
    mode = max(hist, key=hist.get) if hist else None
    bad = [i for i, v in enumerate(res["values"]) if v != mode]
    return {"tag": res["tag"], "trials": len(res["values"]), "distinct": len(hist), "modal": mode,
            "modal_float": float.fromhex(mode) if mode else None, "off_modal_trials": bad,
            "off_modal_values": sorted({res["values"][i] for i in bad}),
            "off_modal_rel": sorted({abs(float.fromhex(res["values"][i]) / float.fromhex(mode) - 1.0) for i in bad}),
            "allreduce_counts": sorted(set(res["allreduces"])), "errors": res["errors"]}


# ------------------------------------------------------------------------------------------------ processes
def _rank_worker(rank, world, port, q, args, envs):
    for kv in envs:
        k, v = kv.split("=", 1)
        os.environ[k] = v
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch
    import torch.distributed as dist
    import nonlinearsolve_jl_amd as nls
    torch.cuda.set_device(0)
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    ctx = nls.Context(device=0)
    nls.set_default_context(ctx)
    comm = "none"
    if world > 1:
        comm = nls.dist.init_comm(ctx, args.transport)
    res = _trials(nls, torch, ctx, args, f"rank{rank}/{world} {comm}", sync=(dist.barrier if world > 1 else None))
    res["device_shared"] = bool(ctx.comm_device_shared()) if world > 1 else False
    q.put((rank, res))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def _competitor(kind, stop_path, args, envs):
    import torch
    torch.cuda.set_device(0)
    if kind == "stream":
        a = torch.zeros(1 << 27, dtype=torch.float64, device="cuda")   # 1 GiB, elementwise: no LDS, no MFMA
        while not os.path.exists(stop_path):
            for _ in range(50):
                a.add_(1.0)
            torch.cuda.synchronize()
    elif kind == "gemm":
        a = torch.randn(2048, 2048, dtype=torch.float64, device="cuda")
        b = torch.randn(2048, 2048, dtype=torch.float64, device="cuda")
        while not os.path.exists(stop_path):
            for _ in range(20):
                c = a @ b  # noqa: F841
            torch.cuda.synchronize()
    elif kind == "solver":
        for kv in envs:
            k, v = kv.split("=", 1)
            os.environ[k] = v
        import nonlinearsolve_jl_amd as nls
        ctx = nls.Context(device=0)
        prob, u0, cache = _make(nls, torch, ctx, args.grid, args)
        while not os.path.exists(stop_path):
            cache.reinit(u0=u0)
            for _ in range(args.steps):
                cache.step()


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def run_processes(args, world, competitor):
    import multiprocessing as mp
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    port = _free_port()
    stop_path = f"/tmp/probe_stop_{os.getpid()}_{port}"
    comp = None
    if competitor != "none":
        comp = mpc.Process(target=_competitor, args=(competitor, stop_path, args, args.env))
        comp.start()
        time.sleep(args.competitor_lead)
    procs = [mpc.Process(target=_rank_worker, args=(r, world, port, q, args, args.env)) for r in range(world)]
    for p in procs:
        p.start()
    out, deadline = {}, time.time() + args.timeout
    while len(out) < world and time.time() < deadline:
        try:
            r, res = q.get(timeout=1.0)
            out[r] = res
        except Exception:  # noqa: BLE001
            if not any(p.is_alive() for p in procs):
                break
    open(stop_path, "w").close()
    hung = [r for r in range(world) if r not in out]
    for p in procs:
        p.join(timeout=10)
        if p.is_alive():
            p.kill()
    if comp is not None:
        comp.join(timeout=20)
        if comp.is_alive():
            comp.kill()
    try:
        os.remove(stop_path)
    except OSError:
        pass
    return out, hung


# ------------------------------------------------------------------------------------------------ threads of ONE process
class _Hub:
    def __init__(self, world):
        self.world = world
        self.bar = threading.Barrier(world)
        self.slots = [None] * world


def _thread_rank(rank, hub, args, results):
    import torch
    import nonlinearsolve_jl_amd as nls
    from nonlinearsolve_jl_amd import core
    from nonlinearsolve_jl_amd.dist import _byte_view
    torch.cuda.set_device(0)
    stream = torch.cuda.Stream()
    world = hub.world

    def on(s):
        return torch.cuda.stream(torch.cuda.ExternalStream(int(s))) if s else torch.cuda.stream(stream)

    def allreduce(user, buf, count, op, s):
        try:
            with on(s):
                t = core._view(buf, count)
                h = t.cpu()
                torch.cuda.current_stream().synchronize()
                hub.slots[rank] = h
                hub.bar.wait()
                acc = hub.slots[0].clone()
                for p in range(1, world):
                    acc = torch.maximum(acc, hub.slots[p]) if op == 1 else acc + hub.slots[p]
                hub.bar.wait()
                t.copy_(acc)
                torch.cuda.current_stream().synchronize()
            return 0
        except Exception:  # noqa: BLE001
            import traceback
            traceback.print_exc()
            return 1

    def alltoallv(user, send, soff, sbytes, recv, roff, rbytes, s):
        try:
            with on(s):
                mine = {}
                for p in range(world):
                    if p != rank and sbytes[p] > 0:
                        mine[p] = _byte_view((send or 0) + soff[p], sbytes[p]).cpu()
                torch.cuda.current_stream().synchronize()
                hub.slots[rank] = mine
                hub.bar.wait()
                got = []
                for p in range(world):
                    if p != rank and rbytes[p] > 0:
                        got.append((p, hub.slots[p][rank].clone()))
                hub.bar.wait()
                for p, hbuf in got:
                    _byte_view((recv or 0) + roff[p], rbytes[p]).copy_(hbuf)
                torch.cuda.current_stream().synchronize()
            return 0
        except Exception:  # noqa: BLE001
            import traceback
            traceback.print_exc()
            return 1

    with torch.cuda.stream(stream):
        ctx = nls.Context(device=0, stream=stream.cuda_stream)
        ctx.comm_init_callbacks(world, rank, allreduce, alltoallv)
        res = _trials(nls, torch, ctx, args, f"thread{rank}/{world} callbacks", sync=hub.bar.wait)
    results[rank] = res


def run_threads(args, world):
    for kv in args.env:
        k, v = kv.split("=", 1)
        os.environ[k] = v
    import torch
    torch.cuda.set_device(0)
    hub = _Hub(world)
    results = {}
    ths = [threading.Thread(target=_thread_rank, args=(r, hub, args, results), daemon=True) for r in range(world)]
    for t in ths:
        t.start()
    deadline = time.time() + args.timeout
    for t in ths:
        t.join(timeout=max(1.0, deadline - time.time()))
    hung = [r for r in range(world) if r not in results]
    return results, hung


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", default="ranks", choices=["ranks", "threads", "solo"])
    ap.add_argument("--world", type=int, default=2)
    ap.add_argument("--transport", default="torch", choices=["torch", "peer"])
    ap.add_argument("--competitor", default="none", choices=["none", "stream", "gemm", "solver"])
    ap.add_argument("--competitor-lead", type=float, default=20.0, help="seconds the competitor gets to start before the workers")
    ap.add_argument("--grid", type=int, default=512)
    ap.add_argument("--trials", type=int, default=30)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--ortho", default="sstep")
    ap.add_argument("--sstep", type=int, default=0)
    ap.add_argument("--sstep-basis", default="auto")
    ap.add_argument("--matfree", action="store_true")
    ap.add_argument("--timeout", type=float, default=240.0)
    ap.add_argument("--env", action="append", default=[])
    ap.add_argument("--label", default="")
    ap.add_argument("--audit", action="store_true", help="content hashes at named points of every step (nk_audit): the first point where a trial differs from trial 0")
    ap.add_argument("--detail", action="store_true", help="per-trial, per-step residual norms on stderr")
    ap.add_argument("--stall-dump", type=float, default=45.0, help="seconds after which a stalled trial dumps its Python stacks")
    args = ap.parse_args()
    t0 = time.time()
    if args.mode == "threads":
        out, hung = run_threads(args, args.world)
    elif args.mode == "solo":
        out, hung = run_processes(args, 1, args.competitor)
    else:
        out, hung = run_processes(args, args.world, "none")
    line = {"label": args.label, "mode": args.mode, "world": 1 if args.mode == "solo" else args.world, "transport": args.transport,
            "competitor": args.competitor, "grid": args.grid, "trials": args.trials, "steps": args.steps, "env": args.env,
            "ortho": args.ortho, "hung_ranks": hung, "seconds": round(time.time() - t0, 1),
            "ranks": [_summary(out[r]) for r in sorted(out)],
            "device_shared": [out[r].get("device_shared") for r in sorted(out)]}
    print(json.dumps(line), flush=True)
    if hung and args.mode == "threads":
        os._exit(3)   # (a thread stuck inside the library cannot be joined)


if __name__ == "__main__":
    main()
