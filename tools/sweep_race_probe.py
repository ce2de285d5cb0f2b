"""Micro-reproducer for the shared-device corruption: ONE s-step sweep launch, repeated, beside a competitor on the same GPU.
The victim runs nk_ss_sweep_test (upload → one sweep → download) `--reps` times on the same input and compares every output word
(updated columns and the reduced Gram block) with the first repetition's. The competitor is a thread of this process on a stream of
its own, or another process:
  --competitor none | thread-sweep | thread-stream | thread-gemm | proc-sweep | proc-stream | proc-gemm
Prints one JSON line: repetitions that differed, the largest relative difference, which output (columns / Gram)."""
import argparse
import ctypes as C
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _sweep_fn():
    from nonlinearsolve_jl_amd import _lib as L
    f = L.lib().nk_ss_sweep_test
    f.restype = C.c_int
    f.argtypes = [C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_double)]
    return f


def _inputs(n, k, s, seed=1):
    rng = np.random.default_rng(seed)
    V = np.asfortranarray(rng.standard_normal((n, k + s)))
    U = rng.standard_normal((k, s)) * 0.1
    Rup = np.triu(rng.standard_normal((s, s))) * 0.3 + 2 * np.eye(s)
    return V, np.concatenate([U.ravel(), Rup.ravel()])


def _run_sweeps(ctx, mode, n, k, s, reps, stop=None, inner=0):
    f = _sweep_fn()
    V0, coef = _inputs(n, k, s)
    ref = None
    bad, worst, where = [], 0.0, set()
    r = 0
    while (stop is None and r < reps) or (stop is not None and not stop.is_set()):
        V = V0.copy(order="F")
        gram = np.zeros((k + s, s))
        us = C.c_double(0)
        rc = f(ctx._h, mode, n, k, s, V.ctypes.data, coef.ctypes.data, gram.ctypes.data, inner, C.byref(us))
        assert rc == 0, rc
        if ref is None:
            ref = (V.copy(), gram.copy())
        else:
            dv = not np.array_equal(V, ref[0])
            dg = not np.array_equal(gram, ref[1])
            if dv or dg:
                bad.append(r)
                if dv:
                    where.add("columns")
                    worst = max(worst, float(np.max(np.abs(V - ref[0])) / np.max(np.abs(ref[0]))))
                if dg:
                    where.add("gram")
                    worst = max(worst, float(np.max(np.abs(gram - ref[1])) / np.max(np.abs(ref[1]))))
        r += 1
    return {"reps": r, "bad": bad[:40], "nbad": len(bad), "worst_rel": worst, "where": sorted(where)}


def _competitor_body(kind, stop, args, own_ctx):
    import torch
    if kind == "sweep":
        _run_sweeps(own_ctx, args.cmode, args.cn or args.n, args.ck or args.k, args.s, 0, stop=stop, inner=args.cinner)
    elif kind == "stream":
        a = torch.zeros(1 << 26, dtype=torch.float64, device="cuda")
        while not stop.is_set():
            for _ in range(20):
                a.add_(1.0)
            torch.cuda.current_stream().synchronize()
    elif kind == "gemm":
        a = torch.randn(2048, 2048, dtype=torch.float64, device="cuda")
        b = torch.randn(2048, 2048, dtype=torch.float64, device="cuda")
        while not stop.is_set():
            for _ in range(10):
                c = a @ b  # noqa: F841
            torch.cuda.current_stream().synchronize()


def _thread_competitor(kind, stop, args):
    import torch
    import nonlinearsolve_jl_amd as nls
    torch.cuda.set_device(0)
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        ctx = nls.Context(device=0, stream=st.cuda_stream)
        _competitor_body(kind, stop, args, ctx)


class _FileStop:
    def __init__(self, path):
        self.path = path

    def is_set(self):
        return os.path.exists(self.path)


def _proc_competitor(kind, stop_path, args, envs):
    for kv in envs:
        k, v = kv.split("=", 1)
        os.environ[k] = v
    import torch
    import nonlinearsolve_jl_amd as nls
    torch.cuda.set_device(0)
    ctx = nls.Context(device=0)
    _competitor_body(kind, _FileStop(stop_path), args, ctx)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", type=int, default=1, help="0 sweep A, 1 sweep B, 2 sweep C")
    ap.add_argument("--n", type=int, default=131072)
    ap.add_argument("--k", type=int, default=16)
    ap.add_argument("--s", type=int, default=15)
    ap.add_argument("--reps", type=int, default=150)
    ap.add_argument("--inner", type=int, default=0, help="extra back-to-back launches per repetition (the harness's timing loop)")
    ap.add_argument("--competitor", default="none")
    ap.add_argument("--cmode", type=int, default=1)
    ap.add_argument("--cn", type=int, default=0)
    ap.add_argument("--ck", type=int, default=0)
    ap.add_argument("--cinner", type=int, default=20)
    ap.add_argument("--own-stream", action="store_true", help="the victim's context on a non-default stream too")
    ap.add_argument("--env", action="append", default=[])
    ap.add_argument("--label", default="")
    args = ap.parse_args()
    for kv in args.env:
        k, v = kv.split("=", 1)
        os.environ[k] = v
    import torch
    import nonlinearsolve_jl_amd as nls
    torch.cuda.set_device(0)
    t0 = time.time()
    comp, stop, stop_path = None, None, None
    if args.competitor.startswith("thread-"):
        stop = threading.Event()
        comp = threading.Thread(target=_thread_competitor, args=(args.competitor[7:], stop, args), daemon=True)
        comp.start()
        time.sleep(3.0)
    elif args.competitor.startswith("proc-"):
        import multiprocessing as mp
        stop_path = f"/tmp/sweep_probe_stop_{os.getpid()}"
        comp = mp.get_context("spawn").Process(target=_proc_competitor, args=(args.competitor[5:], stop_path, args, args.env))
        comp.start()
        time.sleep(20.0)
    if args.own_stream:
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            ctx = nls.Context(device=0, stream=st.cuda_stream)
            res = _run_sweeps(ctx, args.mode, args.n, args.k, args.s, args.reps, inner=args.inner)
    else:
        ctx = nls.Context(device=0)
        res = _run_sweeps(ctx, args.mode, args.n, args.k, args.s, args.reps, inner=args.inner)
    if stop is not None:
        stop.set()
    if stop_path is not None:
        open(stop_path, "w").close()
    if comp is not None:
        comp.join(timeout=30)
        if stop_path is not None:
            if comp.is_alive():
                comp.kill()
            os.remove(stop_path)
    res.update(label=args.label, mode=args.mode, n=args.n, k=args.k, s=args.s, competitor=args.competitor, env=args.env,
               inner=args.inner, seconds=round(time.time() - t0, 1))
    print(json.dumps(res), flush=True)
    os._exit(0)


if __name__ == "__main__":
    main()
