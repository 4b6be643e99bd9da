"""Multi-GPU inference: shard a scene's reference views across ranks, gather the finished maps once.

Every reference view is an independent forward (the reference's drivers loop over them one at a time,
test_dtu_dypcd.py:424-439), so the path shards with NO data-path collective: one process per GPU, each
owning a contiguous slice of the (scan, ref_view) list.  The only exchange is the final gather of the
depth / confidence maps to rank 0 (RCCL over xGMI when the backend is "nccl"; "gloo" on CPU for tests).
"""
from __future__ import annotations

from typing import Callable, List, Sequence

import torch
import torch.distributed as dist


def shard_bounds(n_items: int, rank: int, world: int):
    """Contiguous, balanced slice [lo, hi) of n_items for this rank (sizes differ by at most one)."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_views(items: Sequence, rank: int, world: int) -> List:
    lo, hi = shard_bounds(len(items), rank, world)
    return list(items[lo:hi])


def gather_maps(local: torch.Tensor, n_total: int, dst: int = 0):
    """Gather per-view maps [n_local, ...] from every rank to ``dst`` in view order -> [n_total, ...] on dst,
    None elsewhere.  Shards may differ in length by one: every rank pads to the longest shard so that a
    single fixed-size gather suffices (one collective per tensor, sized for the whole shard)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return local
    world, rank = dist.get_world_size(), dist.get_rank()
    n_max = -(-n_total // world)
    pad = n_max - local.shape[0]
    send = local if pad == 0 else torch.cat([local, local.new_zeros((pad,) + tuple(local.shape[1:]))])
    send = send.contiguous()
    bufs = [torch.empty_like(send) for _ in range(world)] if rank == dst else None
    dist.gather(send, bufs, dst=dst)
    if rank != dst:
        return None
    parts = []
    for r in range(world):
        lo, hi = shard_bounds(n_total, r, world)
        parts.append(bufs[r][: hi - lo])
    return torch.cat(parts)


def run_sharded(items: Sequence, forward: Callable, dst: int = 0):
    """Run ``forward(item) -> (depth [H,W], confidence [h,w])`` on this rank's shard of ``items`` and gather
    both maps to ``dst``.  Returns {"depth": [n,H,W], "confidence": [n,h,w]} on dst, None elsewhere."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    if len(items) < world:
        # checked on EVERY rank before any forward: raising only on the empty ranks would leave the others waiting in the gather
        raise ValueError(f"more ranks ({world}) than reference views ({len(items)})")
    mine = shard_views(items, rank, world)
    depths, confs = [], []
    for it in mine:
        d, c = forward(it)
        depths.append(d)
        confs.append(c)
    d_all = gather_maps(torch.stack(depths), len(items), dst)
    c_all = gather_maps(torch.stack(confs), len(items), dst)
    if rank != dst:
        return None
    return {"depth": d_all, "confidence": c_all}
