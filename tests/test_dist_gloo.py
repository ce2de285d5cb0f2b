"""world_size-2 `gloo` tests (CPU) of the multi-GPU path's host logic and protocol.

What runs on the GPUs at N > 1 is: (a) the row-range partition from nk_partition_range (C ABI, host logic),
(b) a halo exchange of boundary grid lines per operator apply, (c) an all-reduce of the partial inner products.
Here each rank does the *local arithmetic with the NumPy oracle* (there is deliberately no CPU kernel in the
product) and the collectives with torch.distributed/gloo, and the result must equal the serial oracle:
this pins the partition, the halo plan and the placement of the all-reduces — the parts that cannot be tested
on a 1-GPU box."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _allreduce(x):
    t = torch.tensor(np.atleast_1d(np.asarray(x, dtype=np.float64)))
    dist.all_reduce(t)
    out = t.numpy()
    return out if np.ndim(x) else float(out[0])


def _exchange_lines(rank, world, lo_send, hi_send, n):
    """send my first line down / last line up, receive the neighbours' lines (None at the physical boundary)."""
    lo = hi = None
    reqs = []
    if rank > 0:
        reqs.append(dist.isend(torch.tensor(lo_send), rank - 1))
        lo_t = torch.empty(n, dtype=torch.float64)
        reqs.append(dist.irecv(lo_t, rank - 1))
    if rank < world - 1:
        reqs.append(dist.isend(torch.tensor(hi_send), rank + 1))
        hi_t = torch.empty(n, dtype=torch.float64)
        reqs.append(dist.irecv(hi_t, rank + 1))
    for r in reqs:
        r.wait()
    if rank > 0:
        lo = lo_t.numpy()
    if rank < world - 1:
        hi = hi_t.numpy()
    return lo, hi


def _worker(rank, world, port, ns, q):
    import sys
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import nonlinearsolve_jl_amd as nls
        from oracle import reference_restatement as R
        p = R.Bratu2D(ns)
        # (a) the partition the device code uses
        b, e = nls.partition_range(ns, 1, world, rank)          # grid lines
        rb, re_ = nls.partition_range(ns * ns, ns, world, rank)  # rows, granule = one grid line
        assert (rb, re_) == (b * ns, e * ns)
        rng = np.random.default_rng(0)
        u = 0.1 * rng.standard_normal(p.n)
        J = p.jac(u)
        Jloc = J[rb:re_]  # local rows, global columns

        # (b) distributed operator apply: local rows × [local | halo lines]
        def matvec_local(x_loc):
            lo, hi = _exchange_lines(rank, world, x_loc[:ns].copy(), x_loc[-ns:].copy(), ns)
            xg = np.zeros(p.n)
            xg[rb:re_] = x_loc
            if lo is not None:
                xg[rb - ns:rb] = lo
            if hi is not None:
                xg[re_:re_ + ns] = hi
            # every column the local rows touch must lie in [local | the two halo lines]
            cols = np.unique(Jloc.indices)
            assert cols.min() >= max(rb - ns, 0) and cols.max() < min(re_ + ns, p.n)
            return Jloc @ xg

        x = rng.standard_normal(p.n)
        y_loc = matvec_local(x[rb:re_])
        assert np.allclose(y_loc, (J @ x)[rb:re_], rtol=1e-14, atol=1e-13)

        # (c) distributed GMRES (CGS2: 3 all-reduces per Arnoldi step) == serial oracle
        bvec = rng.standard_normal(p.n)
        n_ar = [0]

        def ar(z):
            n_ar[0] += 1
            return _allreduce(z)
        x_loc, info = R.gmres(matvec_local, bvec[rb:re_], rtol=1e-9, restart=30, itmax=4000, ortho="cgs2", allreduce=ar)
        x_ser, info_s = R.gmres(lambda z: J @ z, bvec, rtol=1e-9, restart=30, itmax=4000, ortho="cgs2")
        assert info.converged and info.iters == info_s.iters
        assert np.linalg.norm(x_loc - x_ser[rb:re_]) <= 1e-7 * np.linalg.norm(x_ser)
        assert n_ar[0] == 3 * info.iters + 1 + info.restarts  # β₀ + 3 per step + one per restart

        # (d) the device's default scheme — CGS2 with delayed re-orthogonalisation, ONE all-reduce per Arnoldi step
        n_ar[0] = 0
        x_1r, info_1r = R.gmres_dcgs2_1r(matvec_local, bvec[rb:re_], rtol=1e-9, restart=30, itmax=4000, allreduce=ar)
        assert info_1r.converged and abs(info_1r.iters - info_s.iters) <= 1  # (its stopping test lags one column)
        assert np.linalg.norm(x_1r - x_ser[rb:re_]) <= 1e-7 * np.linalg.norm(x_ser)
        # β₀ per cycle (first + one per restart) + one per Arnoldi step + one scalar closing each cycle that ran to its end
        cycles = info_1r.restarts + 1
        assert info_1r.iters + cycles <= n_ar[0] <= info_1r.iters + 2 * cycles + 1

        # (e) the s-step form: per block of s columns TWO all-reduces of (k + s)·s values (the block's projections and its
        # Gram matrix travel together), β₀ per cycle — 2·⌈30/s⌉ + 1 per full cycle instead of 30 + 2
        n_ar[0] = 0
        x_ss, info_ss = R.gmres(matvec_local, bvec[rb:re_], restart=30, fixed_iters=60, ortho=("sstep", 6), allreduce=ar)
        x_ser60, _ = R.gmres(lambda z: J @ z, bvec, restart=30, fixed_iters=60, ortho="cgs2")
        assert info_ss.iters == 60 and info_ss.restarts == 1
        assert np.linalg.norm(x_ss - x_ser60[rb:re_]) <= 1e-9 * np.linalg.norm(x_ser60)
        assert n_ar[0] == 2 * (2 * 5) + 2

        # distributed residual + ∞-norm (max all-reduce) as in the Newton driver
        lo, hi = _exchange_lines(rank, world, u[rb:rb + ns].copy(), u[re_ - ns:re_].copy(), ns)
        f_loc = p.f(u)[rb:re_]
        t = torch.tensor([np.max(np.abs(f_loc))])
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        assert np.isclose(float(t), np.max(np.abs(p.f(u))))
        q.put((rank, "ok"))
    except Exception as ex:  # pragma: no cover
        import traceback
        q.put((rank, "FAIL: " + traceback.format_exc()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("ns", [16, 33])
def test_row_partitioned_gmres_gloo_world2(ns):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, ns, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert all(r[1] == "ok" for r in res), res
