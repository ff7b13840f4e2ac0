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
