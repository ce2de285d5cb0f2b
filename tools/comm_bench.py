#!/usr/bin/env python
"""Small-collective microbenchmark: N processes (default 2) share cuda:0 — hipIpc works between processes on one device —
and run (a) K all-reduces of 62 doubles (the fused inner products of one Arnoldi step), (b) K halo exchanges of one 1024-point
grid line per neighbour (Bratu CSR SpMV of a 1024² problem split by rows, timed as SpMV-with-halo minus the same rows without
neighbours is not separable on one GPU, so the whole distributed SpMV is reported), (c) fixed-work Newton steps, each with the
callback transport (torch.distributed/gloo, host-staged) and with the peer-mapped kernels. One JSON line per transport."""
import json
import os
import socket
import sys
import time

import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def worker(rank, world, port, transport, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import nonlinearsolve_jl_amd as nls
    torch.cuda.set_device(0)
    ctx = nls.Context(device=0)
    nls.set_default_context(ctx)
    comm = nls.dist.init_comm(ctx, transport)
    dev = torch.device("cuda:0")
    K = 200
    out = {"transport": comm, "ranks": world}
    # (a) all-reduce through the library's reductions: ctx.dot = one 2-launch reduction + one all-reduce of 1 double
    x = torch.ones(4096, dtype=torch.float64, device=dev)
    for _ in range(10):
        ctx.dot(x, x)
    dist.barrier(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(K):
        ctx.dot(x, x)
    torch.cuda.synchronize(); out["dot_with_allreduce_us"] = 1e6 * (time.perf_counter() - t0) / K
    # (b) distributed CSR SpMV on Bratu 1024² (one grid line from each neighbour)
    P = nls.Bratu2D(1024, 6.0)
    n = P.n_local
    u = torch.zeros(n, dtype=torch.float64, device=dev)
    J = P.jac_csr(); P.jac_values(u, J)
    v = torch.ones(n, dtype=torch.float64, device=dev); y = torch.empty_like(v)
    for _ in range(10):
        J.matvec(v, out=y)
    dist.barrier(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(K):
        J.matvec(v, out=y)
    torch.cuda.synchronize(); out["spmv_with_halo_us"] = 1e6 * (time.perf_counter() - t0) / K
    # (c) fixed-work Newton steps (30 Arnoldi steps each): per-Arnoldi-step time
    prob = nls.NonlinearProblem(P); prob.u0 = u
    alg = nls.NewtonRaphson(linsolve=nls.KrylovJL_GMRES(gmres_restart=30, maxiters=30, fixed_iters=30), concrete_jac=True)
    cache = nls.init(prob, alg, abstol=1e-300, maxiters=10 ** 9)
    for _ in range(2):
        cache.step()
    dist.barrier(); torch.cuda.synchronize(); t0 = time.perf_counter()
    S = 10
    for _ in range(S):
        cache.step()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    st = cache.stats
    out.update(newton_steps_per_s=S / dt, us_per_arnoldi_step=1e6 * dt / (30 * S), allreduces=st.allreduces,
               halo_exchanges=st.halo_exchanges, fnorm_inf=cache.fnorm_inf, peer_status=ctx.comm_peer_status())
    cache.close()
    if rank == 0:
        q.put(out)
    dist.barrier()
    dist.destroy_process_group()


def main():
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    for transport in ("torch", "peer"):
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        port = free_port()
        procs = [ctx.Process(target=worker, args=(r, world, port, transport, q)) for r in range(world)]
        for p in procs:
            p.start()
        try:
            print(json.dumps(q.get(timeout=240)))
        except Exception as ex:  # noqa: BLE001
            print(json.dumps({"transport": transport, "error": repr(ex)}))
        for p in procs:
            p.join(timeout=60)
            if p.is_alive():
                p.kill()
        sys.stdout.flush()


if __name__ == "__main__":
    main()
