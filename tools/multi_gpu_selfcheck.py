"""Known-answer self-check of the multi-GPU transports, run by `bench.py --gpus N` (N > 1) before anything is timed, and
runnable on its own under torch.distributed.run. For the transport the bench is about to use it checks, on every rank:

  1. all-reduce (sum and max) of vectors whose result is known in closed form — 1, 64 and 465 values (an s-step block);
  2. the halo exchange inside the row-partitioned CSR SpMV: J(u = 0)·1 of a Bratu grid, whose rows sum to
     4 − (#neighbours) − λh² — an answer computed from the grid indices alone;
  3. three fixed-work Newton steps of a small partitioned Bratu problem against the SAME steps on one rank (rank 0 runs them
     on a private single-rank context): ‖F‖∞ after the steps equal to 1e-10 relative.

The verdict is collective (every rank reports, the minimum counts). It exists because the peer-mapped path (hipIpc arenas,
system-scope flags) was developed with all ranks on ONE device: the first run across xGMI is the driver's scaling run, and a
wrong transport must fail loudly there, not skew a number. Output: a dict for bench.py's `config.comm_selfcheck`."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def selfcheck(nls, ctx, torch, dist, comm_name):
    world, rank = dist.get_world_size(), dist.get_rank()
    dev = torch.device("cuda", ctx.device)
    out = {"transport": comm_name, "world": world}
    t0 = time.perf_counter()
    ok = True

    def collective_ok(flag):   # every rank must agree
        t = torch.tensor([1.0 if flag else 0.0], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return bool(t.item() > 0.5)

    # ---- 1. all-reduces with closed-form answers
    for count in (1, 64, 465):
        idx = torch.arange(count, dtype=torch.float64, device=dev)
        v = (rank + 1.0) * (idx + 1.0)
        s = ctx.allreduce(v.clone(), op="sum")
        want = (world * (world + 1) / 2.0) * (idx + 1.0)
        m = ctx.allreduce(v.clone(), op="max")
        good = bool(torch.equal(s, want)) and bool(torch.equal(m, world * (idx + 1.0)))
        out[f"allreduce_{count}"] = collective_ok(good)
        ok &= out[f"allreduce_{count}"]
    # ---- 2. halo exchange: row sums of the Bratu Jacobian at u = 0
    ns = 32 * world
    P = nls.Bratu2D(ns, 6.0)
    nl, b = P.n_local, P.row_begin
    J = P.jac_csr()
    P.jac_values(torch.zeros(nl, dtype=torch.float64, device=dev), J)
    y = J.matvec(torch.ones(nl, dtype=torch.float64, device=dev))
    k = torch.arange(b, b + nl, device=dev)
    i, j = k % ns, k // ns
    nbrs = (i > 0).double() + (i < ns - 1).double() + (j > 0).double() + (j < ns - 1).double()
    h2 = 1.0 / (ns + 1) ** 2
    want = 4.0 - nbrs - 6.0 * h2
    out["halo_spmv"] = collective_ok(bool(float((y - want).abs().max()) <= 1e-12))
    ok &= out["halo_spmv"]
    # ---- 3. three fixed-work Newton steps, partitioned vs one rank
    alg = nls.NewtonRaphson(linsolve=nls.KrylovJL_GMRES(gmres_restart=20, maxiters=20, fixed_iters=20), concrete_jac=True)
    prob = nls.NonlinearProblem(P, u0=torch.zeros(nl, dtype=torch.float64, device=dev))
    cache = nls.init(prob, alg, abstol=1e-300, maxiters=10 ** 6)
    for _ in range(3):
        cache.step()
    f_part = float(cache.fnorm_inf)
    ar = int(cache.stats.allreduces)
    cache.close()
    ref = torch.zeros(1, dtype=torch.float64)
    if rank == 0:
        solo = nls.Context(device=ctx.device)                 # a private context without a communicator: the same steps on one rank
        P1 = nls.Bratu2D(ns, 6.0, ctx=solo)
        c1 = nls.init(nls.NonlinearProblem(P1, u0=torch.zeros(ns * ns, dtype=torch.float64, device=dev)), alg, abstol=1e-300, maxiters=10 ** 6)
        for _ in range(3):
            c1.step()
        ref[0] = float(c1.fnorm_inf)
        c1.close()
    if dist.get_backend() == "nccl":
        r = ref.to(dev)
        dist.broadcast(r, 0)
        ref = r.cpu()
    else:
        dist.broadcast(ref, 0)
    rel = abs(f_part - float(ref[0])) / abs(float(ref[0]))
    out["newton_3steps_rel_diff_vs_one_rank"] = rel
    out["newton_3steps"] = collective_ok(rel <= 1e-10)
    out["allreduces_in_3_steps"] = ar
    ok &= out["newton_3steps"]
    out["peer_timeouts"] = int(ctx.comm_peer_status()[1])
    out["ok"] = bool(ok and out["peer_timeouts"] == 0)
    out["seconds"] = round(time.perf_counter() - t0, 2)
    return out


def main():
    import torch
    import torch.distributed as dist
    import nonlinearsolve_jl_amd as nls
    backend = os.environ.get("BENCH_BACKEND", "nccl")
    dist.init_process_group(backend)
    local = int(os.environ.get("LOCAL_RANK", "0")) % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local)
    ctx = nls.Context(device=local)
    nls.set_default_context(ctx)
    results = []
    for tr in (["peer", "rccl", "torch"] if backend == "nccl" else ["peer", "torch"]):
        try:
            name = nls.dist.init_comm(ctx, tr)
            results.append(selfcheck(nls, ctx, torch, dist, name))
        except Exception as ex:  # noqa: BLE001
            results.append({"transport": tr, "ok": False, "error": str(ex)})
        try:
            ctx.comm_peer_disable()
        except Exception:  # noqa: BLE001
            pass
    if dist.get_rank() == 0:
        import json
        print(json.dumps(results))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
