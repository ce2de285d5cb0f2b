"""One process per GPU: communicator bootstrap for libmi355x_nk.

`init_comm(ctx)` picks the transport:
  * "rccl"  (default on GPUs when torch.distributed's backend is nccl): the library owns its own RCCL
    communicator, created from a ncclUniqueId that rank 0 generates and torch.distributed broadcasts. All
    Krylov all-reduces and halo send/recvs are then enqueued by the C++ code on the compute stream — no Python
    in the inner loop.
  * "torch" : collectives are routed through torch.distributed from C callbacks (nk_comm_callbacks). Works with
    the gloo backend (host staging), which is how the multi-rank code path is exercised on a single GPU
    (2 processes sharing cuda:0) and on CPU-only CI; also usable with nccl as an escape hatch.
  * "peer"  : one of the above for set-up and large exchanges, plus the xGMI-native fast path for the small collectives of
    the Krylov loop — every rank exports an uncached arena (hipIpc), torch.distributed all-gathers the 64-byte handles,
    every rank maps all of them; all-reduces and halo exchanges then run as one small kernel each (csrc/nk_ctx.hip).
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np
import torch
import torch.distributed as dist

from . import core


def _backend():
    return dist.get_backend() if dist.is_initialized() else None


def init_comm(ctx: core.Context, transport: str | None = None) -> str:
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return "none"
    world, rank = dist.get_world_size(), dist.get_rank()
    transport = transport or os.environ.get("NK_COMM", "rccl" if _backend() == "nccl" else "torch")
    if transport == "peer":
        base = init_comm(ctx, os.environ.get("NK_COMM_BASE", "rccl" if _backend() == "nccl" else "torch"))
        try:
            ok = enable_peer(ctx)
        except Exception as ex:  # noqa: BLE001 — e.g. no IPC between these devices: every rank fails alike
            print(f"[nk dist] peer path unavailable on rank {rank}: {ex}")
            ok = False
        return ("peer+" + base) if ok else base + "(peer path unavailable)"
    if transport == "rccl":
        dev = torch.device("cuda", ctx.device)
        if rank == 0:
            uid = torch.tensor(list(core.comm_unique_id()), dtype=torch.uint8)
        else:
            uid = torch.zeros(128, dtype=torch.uint8)
        if _backend() == "nccl":
            uid = uid.to(dev)
        dist.broadcast(uid, 0)
        ctx.comm_init_rccl(world, rank, bytes(uid.cpu().tolist()))
        return "rccl"
    if transport == "torch":
        _init_torch_callbacks(ctx, world, rank)
        return "torch"
    raise ValueError(f"unknown transport {transport!r}")


def enable_peer(ctx: core.Context, arena_bytes: int = 0) -> bool:
    """Layer the peer-mapped fast path over an initialised communicator (collective). Every decision is taken by all
    ranks together: a rank that cannot export or map an arena makes everybody stay on the base transport."""
    world = dist.get_world_size()
    try:
        mine = ctx.comm_peer_handle(arena_bytes)
    except Exception as ex:  # noqa: BLE001
        mine = repr(ex)
    parts = [None] * world
    dist.all_gather_object(parts, mine)
    if not all(isinstance(h, bytes) for h in parts):
        return False
    try:
        ctx.comm_enable_peer(b"".join(parts))
        mapped = True
    except Exception:  # noqa: BLE001
        mapped = False
    flags = [None] * world
    dist.all_gather_object(flags, mapped)
    if not all(flags):
        ctx.comm_peer_disable()
        return False
    return ctx.comm_peer_selftest()   # collective verdict: all ranks keep the fast path, or none does


def _init_torch_callbacks(ctx: core.Context, world: int, rank: int):
    host_staged = _backend() != "nccl"

    def _stream(s):
        return torch.cuda.stream(torch.cuda.ExternalStream(int(s))) if s else core._nullctx()

    def allreduce(user, buf, count, op, stream):
        try:
            with _stream(stream):
                t = core._view(buf, count)
                rop = dist.ReduceOp.MAX if op == 1 else dist.ReduceOp.SUM
                if host_staged:
                    h = t.cpu()  # synchronises the stream
                    dist.all_reduce(h, op=rop)
                    t.copy_(h)
                    torch.cuda.current_stream().synchronize()
                else:
                    dist.all_reduce(t, op=rop)
            return 0
        except Exception:  # pragma: no cover
            import traceback
            traceback.print_exc()
            return 1

    def alltoallv(user, send, soff, sbytes, recv, roff, rbytes, stream):
        try:
            with _stream(stream):
                reqs, stage = [], []
                for p in range(world):
                    if p == rank:
                        continue
                    if sbytes[p] > 0:
                        v = _byte_view((send or 0) + soff[p], sbytes[p])
                        src = v.cpu() if host_staged else v
                        reqs.append(dist.isend(src, p))
                        stage.append(src)
                    if rbytes[p] > 0:
                        v = _byte_view((recv or 0) + roff[p], rbytes[p])
                        if host_staged:
                            h = torch.empty(rbytes[p], dtype=torch.uint8)
                            reqs.append(dist.irecv(h, p))
                            stage.append((h, v))
                        else:
                            reqs.append(dist.irecv(v, p))
                for r in reqs:
                    r.wait()
                for item in stage:
                    if isinstance(item, tuple):
                        item[1].copy_(item[0])
                torch.cuda.current_stream().synchronize()
            return 0
        except Exception:  # pragma: no cover
            import traceback
            traceback.print_exc()
            return 1

    ctx.comm_init_callbacks(world, rank, allreduce, alltoallv)


class _ByteView:
    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": "|u1", "data": (int(ptr), False), "version": 2}


def _byte_view(ptr, n):
    return torch.as_tensor(_ByteView(ptr, n), device="cuda")


def gather_vector(local, n_global: int, row_begin: int):
    """Assemble a row-partitioned vector on every rank (test/bench helper)."""
    loc = local.detach().cpu().numpy() if isinstance(local, torch.Tensor) else np.asarray(local)
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return loc.copy()
    parts = [None] * dist.get_world_size()
    dist.all_gather_object(parts, (int(row_begin), loc))
    out = np.empty(n_global)
    for b, a in parts:
        out[b:b + a.size] = a
    return out
