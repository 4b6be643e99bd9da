"""A whole evaluation set -- scans x reference views -- on one or more GPUs (BASELINE.json cfg5: DTU's 22 test scans x 49 reference
views = 1,078 forwards, lists/dtu/test.txt, datasets/general_eval.py:41-51).

Every (scan, reference view) item is an independent forward (test_dtu_dypcd.py:424-439), so the set shards by item with no data-path
collective: one process per GPU, contiguous balanced shards (a rank then sees whole scans or long runs of
one scan).  Two things beyond ``shard.run_sharded``:

* **per-scan feature cache.**  pair.txt names up to 10 source views per reference view (README.md:51-58) and every image of a scan
  is the reference view once, so with N = 5 an image is needed ~5 times; its feature pyramid (and, when it is the reference, its
  context pyramid) is computed ONCE per rank and scan and dropped when the rank moves to the next scan (49 images x 26 MB of
  pyramids at 1600x1184 = 1.3 GB: nothing next to 288 GB of HBM).
* **batched, overlapped gather.**  Finished depth / confidence maps go to the destination rank K views at a time with an
  asynchronous ``gather`` (RCCL over xGMI with backend "nccl": its own stream, so the transfer of batch b overlaps the forwards of
  batch b + 1; 9.5 MB per view at 1600x1184 against ~150 GB/s per link).  Every rank issues the same number of collectives
  (ceil(longest shard / K)); short shards pad their last batch.

The reference writes each depth map to disk from the process that computed it (test_dtu_dypcd.py:454-478) and has no gather; the
gather exists because BASELINE.json's north star asks for one.
"""
from __future__ import annotations

from typing import Callable, Dict, List, Sequence, Tuple

import torch
import torch.distributed as dist

from .shard import shard_bounds

Item = Tuple[int, int, Tuple[int, ...]]          # (scan, reference image, source images)


def synthetic_pairs(n_images: int, n_src: int) -> List[Tuple[int, Tuple[int, ...]]]:
    """pair.txt of a synthetic scan: reference image i with its ``n_src`` best sources = its nearest neighbours on a ring of
    ``n_images`` cameras (the real files rank 10 neighbours by a view-selection score, README.md:51-58; the drivers take the first
    nviews - 1 of them, datasets/general_eval.py:125)."""
    if n_src >= n_images:
        raise ValueError("a scan needs more images than source views per reference view")
    pairs = []
    for i in range(n_images):
        srcs, k = [], 1
        while len(srcs) < n_src:
            for cand in ((i + k) % n_images, (i - k) % n_images):
                if cand not in srcs and cand != i and len(srcs) < n_src:
                    srcs.append(cand)
            k += 1
        pairs.append((i, tuple(srcs)))
    return pairs


def build_items(n_scans: int, n_images: int, n_src: int) -> List[Item]:
    """The flat (scan, ref, sources) list in the reference's order: scans in list order, reference views in pair.txt order
    (datasets/general_eval.py:26-51)."""
    pairs = synthetic_pairs(n_images, n_src)
    return [(s, ref, srcs) for s in range(n_scans) for ref, srcs in pairs]


class ScanFeatureCache:
    """``get(scan, image)`` -> the image's feature pyramid, computed by ``compute(scan, image)`` once per scan on this rank.
    Entries of a scan are dropped when a later scan is first asked for (shards are contiguous: a rank never returns to a scan)."""

    def __init__(self, compute: Callable[[int, int], object]):
        self.compute = compute
        self.scan = None
        self.store: Dict[int, object] = {}
        self.hits = self.misses = 0
        self.max_entries = 0

    def get(self, scan: int, image: int):
        if scan != self.scan:
            self.store.clear()
            self.scan = scan
        if image in self.store:
            self.hits += 1
        else:
            self.misses += 1
            self.store[image] = self.compute(scan, image)
            self.max_entries = max(self.max_entries, len(self.store))
        return self.store[image]


class BatchedGather:
    """Gather per-view maps to ``dst`` in batches of ``k`` views while later views are still being computed.

    ``add(depth, conf)`` after each view; ``finish()`` -> {"depth": [n_total,...], "confidence": [n_total,...]} on dst (view
    order), None elsewhere.  Collectives are issued with ``async_op=True``; at most ``max_in_flight`` batches are pending (their
    staging tensors stay alive until waited for).  ``to_host``: stage through CPU tensors (gloo)."""

    def __init__(self, n_total: int, k: int, dst: int = 0, to_host: bool = False, max_in_flight: int = 2):
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        if n_total < self.world:
            raise ValueError(f"more ranks ({self.world}) than reference views ({n_total})")      # on EVERY rank, before any collective
        self.n_total, self.k, self.dst, self.to_host, self.max_in_flight = n_total, max(1, k), dst, to_host, max_in_flight
        self.n_max = -(-n_total // self.world)
        self.n_batches = -(-self.n_max // self.k)
        self.stage = None          # (depth [k,...], conf [k,...]) of the batch being filled; views are COPIED in (a graph replay's
        self.fill = 0              # outputs are static buffers that the next replay overwrites)
        self.pending = []          # (handles, staging tensors, receive buffers)
        self.received = []         # per batch: (depth bufs, conf bufs) on dst
        self.issued = 0
        self.n_added = 0
        self._shapes = None

    def _new_stage(self):
        (sd, dd, dev), (sc, dc, _) = self._shapes
        return (torch.zeros((self.k,) + sd, dtype=dd, device=dev), torch.zeros((self.k,) + sc, dtype=dc, device=dev))

    def add(self, depth: torch.Tensor, conf: torch.Tensor):
        if self._shapes is None:
            self._shapes = ((tuple(depth.shape), depth.dtype, depth.device), (tuple(conf.shape), conf.dtype, conf.device))
        if self.stage is None:
            self.stage = self._new_stage()
        self.stage[0][self.fill].copy_(depth)
        self.stage[1][self.fill].copy_(conf)
        self.fill += 1
        self.n_added += 1
        if self.fill == self.k:
            self._issue()

    def _issue(self):
        d, c = self.stage if self.stage is not None else self._new_stage()        # an empty (all-zero) batch keeps the collectives matched
        self.stage, self.fill = None, 0
        if self.to_host:
            d, c = d.cpu(), c.cpu()
        self.issued += 1
        if self.world == 1:
            self.received.append(([d], [c]))
            return
        while len(self.pending) >= self.max_in_flight:
            self._wait_one()
        bd = [torch.empty_like(d) for _ in range(self.world)] if self.rank == self.dst else None
        bc = [torch.empty_like(c) for _ in range(self.world)] if self.rank == self.dst else None
        hd = dist.gather(d, bd, dst=self.dst, async_op=True)
        hc = dist.gather(c, bc, dst=self.dst, async_op=True)
        self.pending.append(((hd, hc), (d, c), (bd, bc)))

    def _wait_one(self):
        (hd, hc), _, (bd, bc) = self.pending.pop(0)
        hd.wait()
        hc.wait()
        if self.rank == self.dst:
            self.received.append((bd, bc))

    def finish(self):
        if self.n_added == 0:
            raise ValueError("BatchedGather.finish: this rank produced no view")
        if self.fill:
            self._issue()
        while self.issued < self.n_batches:        # shorter shard: empty batches keep the collectives matched
            self._issue()
        while self.pending:
            self._wait_one()
        if self.rank != self.dst:
            return None
        depth, conf = [], []
        for r in range(self.world):
            lo, hi = shard_bounds(self.n_total, r, self.world)
            n_r = hi - lo
            for b in range(self.n_batches):
                take = min(max(n_r - b * self.k, 0), self.k)
                if take:
                    bd, bc = self.received[b]
                    depth.append(bd[r][:take])
                    conf.append(bc[r][:take])
        return {"depth": torch.cat(depth), "confidence": torch.cat(conf)}


def run_scans(items: Sequence[Item], forward: Callable[[Item], Tuple[torch.Tensor, torch.Tensor]], gather_batch: int = 8, dst: int = 0,
              to_host: bool = False, on_view: Callable[[int], None] = None):
    """This rank's contiguous shard of ``items`` through ``forward(item) -> (depth, confidence)`` with the batched gather.
    Returns ({"depth", "confidence"} on dst else None, number of views this rank computed)."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    gather = BatchedGather(len(items), gather_batch, dst=dst, to_host=to_host)       # validates len(items) >= world on every rank
    lo, hi = shard_bounds(len(items), rank, world)
    for i in range(lo, hi):
        d, c = forward(items[i])
        gather.add(d, c)
        if on_view is not None:
            on_view(i)
    return gather.finish(), hi - lo
