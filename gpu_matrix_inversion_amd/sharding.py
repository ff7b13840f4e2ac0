"""Batch sharding across GPUs: one process per GPU, ``torch.distributed`` (RCCL on ROCm).

The inversion of one matrix touches only that matrix (there is no exchange step
anywhere in Gauss-Jordan), so a batch is partitioned into contiguous index ranges
and every rank inverts its own range.  No data-path collective exists; the only
collective is the optional gather of per-matrix status words / results for a
caller that wants them on every rank.  N = 4096 and N = 16384 single matrices are
"replicas only": one GPU per matrix, never sharded.

The reference has no multi-device code at all (always platforms[0]/devices[0],
mat_inv_32.cpp:239-244); this module is additive.
"""
from __future__ import annotations

from typing import Callable, Tuple


def shard_range(batch: int, world_size: int, rank: int) -> Tuple[int, int]:
    """Rank ``rank`` owns matrices [lo, hi): ceil-sized contiguous ranges, the last ones may be
    short or empty (GPU g owns [g*ceil(B/G), min(B,(g+1)*ceil(B/G)))."""
    if batch < 0 or world_size <= 0 or not (0 <= rank < world_size):
        raise ValueError("bad shard arguments")
    per = -(-batch // world_size)
    lo = min(batch, rank * per)
    hi = min(batch, lo + per)
    return lo, hi


def invert_sharded(batch_tensor, invert_fn: Callable, group=None, gather: bool = True):
    """Invert this rank's shard of a replicated (B,N,N) batch with ``invert_fn(shard) ->
    (inverses, status)`` and, if ``gather``, all-gather inverses and status so every rank
    returns the full (B,N,N) result.  Works on any backend (``nccl`` = RCCL on the GPUs,
    ``gloo`` in the CPU tests where ``invert_fn`` is injected by the test)."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    b = batch_tensor.shape[0]
    lo, hi = shard_range(b, world, rank)
    per = -(-b // world)
    n = batch_tensor.shape[1]
    inv_local = torch.zeros(per, n, n, dtype=batch_tensor.dtype, device=batch_tensor.device)
    st_local = torch.zeros(per, dtype=torch.int32, device=batch_tensor.device)
    if hi > lo:
        inv, st = invert_fn(batch_tensor[lo:hi])
        inv_local[: hi - lo] = inv
        st_local[: hi - lo] = st
    if not gather or world == 1:
        return inv_local[: hi - lo], st_local[: hi - lo], (lo, hi)
    inv_all = [torch.empty_like(inv_local) for _ in range(world)]
    st_all = [torch.empty_like(st_local) for _ in range(world)]
    dist.all_gather(inv_all, inv_local, group=group)
    dist.all_gather(st_all, st_local, group=group)
    return torch.cat(inv_all)[:b], torch.cat(st_all)[:b], (lo, hi)


def invert_distributed(batch_on_root, invert_fn: Callable, root: int = 0, group=None, shard_buffers=None):
    """The xGMI distribution path: the batch lives on ``root`` only.

    1. **scatter** -- root sends every other rank its contiguous shard with ONE grouped batch of
       point-to-point operations (``dist.batch_isend_irecv`` = ``ncclGroupStart/End`` around
       ``ncclSend/ncclRecv`` on RCCL: one direct xGMI link per peer, all seven in flight at once; a ring
       collective would be bound by a single link);
    2. every rank inverts its own shard with ``invert_fn(shard) -> (inverses, status)`` -- no data-path
       collective;
    3. **gather** -- the reverse grouped sends of inverses and status words into root's result, and an
       ``all_reduce(MAX)`` of the worst status so that every rank knows whether the whole batch is valid.

    ``batch_on_root``: (B,N,N) tensor on root, ``None`` elsewhere (shape and dtype travel in a two-word
    broadcast).  Returns ``(inverses, status, worst_status, timings)``: on root the full (B,N,N) result and
    int32[B] status, elsewhere this rank's shard of both; ``timings`` = seconds spent in scatter / compute /
    gather on this rank (device-synchronised where the tensors live on a GPU).
    Works on ``nccl`` (= RCCL) and, for the CPU tests, on ``gloo``.
    ``shard_buffers``: optional dict reused across calls (receive / result buffers are allocated once).
    """
    import time

    import torch
    import torch.distributed as dist

    if not dist.is_initialized():
        inv, st = invert_fn(batch_on_root)
        return inv, st, int(st.max()) if st.numel() else 0, {"scatter": 0.0, "compute": 0.0, "gather": 0.0}
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    is_root = rank == root
    if is_root:
        if batch_on_root is None or batch_on_root.dim() != 3 or batch_on_root.shape[1] != batch_on_root.shape[2]:
            raise ValueError("root must pass a (B,N,N) batch")
        dev = batch_on_root.device
        meta = torch.tensor([batch_on_root.shape[0], batch_on_root.shape[1],
                             1 if batch_on_root.dtype == torch.float64 else 0], dtype=torch.int64, device=dev)
    else:
        dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
        meta = torch.zeros(3, dtype=torch.int64, device=dev)
    dist.broadcast(meta, src=root, group=group)
    b, n, is64 = (int(v) for v in meta.tolist())
    dtype = torch.float64 if is64 else torch.float32
    lo, hi = shard_range(b, world, rank)
    cnt = hi - lo

    def sync():
        if dev.type == "cuda":
            torch.cuda.synchronize(dev)

    bufs = shard_buffers if shard_buffers is not None else {}
    key = (b, n, dtype, str(dev))
    if bufs.get("key") != key:
        bufs.clear()
        bufs["key"] = key
        if not is_root:
            bufs["shard"] = torch.empty(cnt, n, n, dtype=dtype, device=dev)
        else:
            bufs["out"] = torch.empty(b, n, n, dtype=dtype, device=dev)
            bufs["status"] = torch.empty(b, dtype=torch.int32, device=dev)

    # 1. scatter: one group of sends on root, one receive on every other rank
    sync()
    t0 = time.perf_counter()
    ops = []
    if is_root:
        for r in range(world):
            rlo, rhi = shard_range(b, world, r)
            if r != root and rhi > rlo:
                ops.append(dist.P2POp(dist.isend, batch_on_root[rlo:rhi], r, group))
        shard = batch_on_root[lo:hi]
    else:
        shard = bufs["shard"]
        if cnt > 0:
            ops.append(dist.P2POp(dist.irecv, shard, root, group))
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    sync()
    t1 = time.perf_counter()

    # 2. every rank inverts its own shard
    if cnt > 0:
        inv, st = invert_fn(shard)
        st = st.to(torch.int32)
    else:
        inv = torch.empty(0, n, n, dtype=dtype, device=dev)
        st = torch.empty(0, dtype=torch.int32, device=dev)
    sync()
    t2 = time.perf_counter()

    # 3. gather into root (grouped), worst status to everyone
    ops = []
    if is_root:
        out, status = bufs["out"], bufs["status"]
        out[lo:hi] = inv
        status[lo:hi] = st
        for r in range(world):
            rlo, rhi = shard_range(b, world, r)
            if r != root and rhi > rlo:
                ops.append(dist.P2POp(dist.irecv, out[rlo:rhi], r, group))
                ops.append(dist.P2POp(dist.irecv, status[rlo:rhi], r, group))
    elif cnt > 0:
        ops.append(dist.P2POp(dist.isend, inv.contiguous(), root, group))
        ops.append(dist.P2POp(dist.isend, st.contiguous(), root, group))
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    worst = torch.tensor([int(st.max()) if cnt > 0 else 0], dtype=torch.int32, device=dev)
    dist.all_reduce(worst, op=dist.ReduceOp.MAX, group=group)
    sync()
    t3 = time.perf_counter()
    timings = {"scatter": t1 - t0, "compute": t2 - t1, "gather": t3 - t2}
    if is_root:
        return out, status, int(worst.item()), timings
    return inv, st, int(worst.item()), timings
