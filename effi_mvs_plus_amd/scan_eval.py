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

    def __init__(self, compute: Callable[[int, int], object], on_drop: Callable[[], None] = None):
        self.compute = compute
        self.on_drop = on_drop          # called before a scan's entries are released (stream ordering of their memory, ScanRunner)
        self.scan = None
        self.store: Dict[int, object] = {}
        self.hits = self.misses = 0
        self.max_entries = 0

    def drop(self):
        if self.store and self.on_drop is not None:
            self.on_drop()
        self.store.clear()
        self.scan = None

    def get(self, scan: int, image: int):
        if scan != self.scan:
            self.drop()
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

    def __init__(self, n_total: int, k: int, dst: int = 0, to_host: bool = False, max_in_flight: int = 2, lanes=None):
        # ``lanes``: the streams ``add`` is called under when several views are in flight (run_scans with a ScanRunner).  Staging
        # batches are then allocated on the stream that was current at construction ("main"), every lane waits for a new batch's
        # zero fill before copying into it, and main waits for the lanes before a batch leaves (collective / host copy / finish).
        self.lanes = list(lanes) if lanes else None
        self.main = torch.cuda.current_stream() if self.lanes else None
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
        if self.lanes is None:
            return (torch.zeros((self.k,) + sd, dtype=dd, device=dev), torch.zeros((self.k,) + sc, dtype=dc, device=dev))
        with torch.cuda.stream(self.main):
            st = (torch.zeros((self.k,) + sd, dtype=dd, device=dev), torch.zeros((self.k,) + sc, dtype=dc, device=dev))
        for lane in self.lanes:
            lane.wait_stream(self.main)
        return st

    def _join_lanes(self):
        if self.lanes is not None:
            for lane in self.lanes:
                self.main.wait_stream(lane)

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
        self.issued += 1
        if self.world == 1 and not self.to_host:
            self.received.append(([d], [c]))       # nothing leaves the device: finish() joins the lanes once
            return
        if self.lanes is None:
            self._send(d, c)
        else:
            self._join_lanes()                     # the batch's views were written on the lanes
            with torch.cuda.stream(self.main):
                self._send(d, c)

    def _send(self, d, c):
        if self.to_host:
            d, c = d.cpu(), c.cpu()
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
        self._join_lanes()
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
    Returns ({"depth", "confidence"} on dst else None, number of views this rank computed).

    ``forward`` may be a ``ScanRunner``: it has ``lanes`` (streams) and is then called as ``forward(item, slot)`` under lane
    ``slot`` = position in the shard modulo the number of lanes, so that many views are in flight; the maps it returns are the
    slot's static graph outputs, copied into the gather's staging batch on the same lane before the slot is replayed again."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    lanes = getattr(forward, "lanes", None)
    gather = BatchedGather(len(items), gather_batch, dst=dst, to_host=to_host, lanes=lanes)   # validates len(items) >= world on every rank
    lo, hi = shard_bounds(len(items), rank, world)
    for j, i in enumerate(range(lo, hi)):
        if lanes:
            slot = j % len(lanes)
            with torch.cuda.stream(lanes[slot]):
                d, c = forward(items[i], slot)
                gather.add(d, c)
        else:
            d, c = forward(items[i])
            gather.add(d, c)
        if on_view is not None:
            on_view(i)
    return gather.finish(), hi - lo


class ScanRunner:
    """images -> feature pyramids (cached per scan) -> context pyramid + cost-volume hot path, ``slots`` reference views in flight.

    What one item (scan, ref, sources) costs on the device: the feature pyramid of every image of the item this rank has not seen in
    this scan (on average ONE per item), the reference image's context pyramid and the hot path.  How it is launched:

    * a **producer stream** computes missing feature pyramids eagerly (``net.feature``: channel-last stage maps) into the per-scan
      cache; it owns every cache allocation, and waits for all lanes before a scan's entries are dropped, so that the memory is
      never handed out again while a replay still reads it;
    * each of the ``slots`` **lanes** (a stream + a captured hipGraph with static inputs) replays ``context pyramid -> forward_hot``.
      The graph reads the item's feature maps through an ``ops.ViewTable`` -- a 3 x 14 table of device pointers that one small
      launch rewrites per item -- so nothing of the 150 MB of maps is copied; the reference image is written straight into the
      slot's static image buffer by ``image_fn(scan, image, out)``; cameras / depth range are copied into the slot (a few KB);
    * the lane waits for the producer's event of the newest pyramid it reads, nothing else.

    ``__call__(item, slot)`` must run under ``lanes[slot]`` (``run_scans`` does that) and returns the slot's static
    (final depth [H, W], confidence [H/2, W/2]); they are valid until the slot's next replay.  Results are bitwise those of
    ``net.forward_hot`` on the same maps (tests/test_gpu_scan.py).  Reference behaviour: every reference view is an independent
    forward over its pair.txt sources (test_dtu_dypcd.py:424-439, datasets/general_eval.py:26-51)."""

    def __init__(self, net, image_fn, example_item: Item, proj_matrices, depth_values, slots: int = 3, branches=None):
        from . import ops
        from .graph import ReplayGraph
        self.net, self.image_fn = net, image_fn
        self.slots = max(1, int(slots))
        dev = depth_values.device
        self.device = dev
        scan, ref, srcs = example_item
        self.n_views = 1 + len(srcs)
        self.producer = torch.cuda.Stream(device=dev)
        self.lanes = [torch.cuda.Stream(device=dev) for _ in range(self.slots)]
        self._seq = 0
        self._lane_seq = [0] * self.slots
        self.cache = ScanFeatureCache(self._pyramid, on_drop=self._before_drop)
        self.n_images_prepared = 0
        with torch.no_grad():
            img0 = self._image(scan, ref)
            self._img_shape = tuple(img0.shape)
            self._scratch = torch.empty_like(img0)          # the producer's image buffer (stream-ordered reuse)
            ents = [self.cache.get(scan, v) for v in (ref,) + tuple(srcs)]
            shapes = [(m.shape[2], m.shape[0], m.shape[1]) for m in ents[0].maps]
            proto = ops.ViewTable(shapes, self.n_views, dev)
            torch.cuda.current_stream().wait_stream(self.producer)
            proto.set([[e.maps[s] for e in ents] for s in range(len(shapes))])

            def fn(img, ptrs, pm, dv):
                return net.forward_hot(proto.rebind(ptrs), net.cnet_depth(img), pm, dv)

            br0 = ops.get_branches()
            ops.set_branches((self.slots > 1) if branches is None else bool(branches))     # views in flight: each graph keeps the pass's side stream
            try:
                self.graph = ReplayGraph(fn, (img0, proto.ptrs, proj_matrices, depth_values), slots=self.slots)
            finally:
                ops.set_branches(br0)
            self.tables = [proto.rebind(self.graph.inputs[i][1]) for i in range(self.slots)]
            torch.cuda.synchronize(dev)
            del ents
        self.cache.drop()
        self.cache.hits = self.cache.misses = self.cache.max_entries = 0
        self.n_images_prepared = 0

    # -- producer side ------------------------------------------------------------------------------------------------------------
    def _image(self, scan, image, out=None):
        self.n_images_prepared += 1
        return self.image_fn(scan, image, out)

    def _pyramid(self, scan, image):
        from . import ops
        with torch.cuda.stream(self.producer):
            img = self._image(scan, image, getattr(self, "_scratch", None))
            f = self.net.feature(img)
            maps = ops.to_nhwc([f[k][0] for k in sorted(f)])       # zero-copy: the heads write channel-last
            ev = torch.cuda.Event()
            ev.record(self.producer)
        self._seq += 1
        return _CacheEntry(maps, ev, self._seq)

    def _before_drop(self):
        for lane in self.lanes:                     # every replay enqueued so far may still read the entries being dropped
            self.producer.wait_stream(lane)

    # -- lane side ----------------------------------------------------------------------------------------------------------------
    def __call__(self, item: Item, slot: int, proj_matrices=None, depth_values=None):
        scan, ref, srcs = item
        if 1 + len(srcs) != self.n_views:
            raise ValueError(f"ScanRunner: captured for {self.n_views} views, item has {1 + len(srcs)}")
        lane = self.lanes[slot]
        ents = [self.cache.get(scan, v) for v in (ref,) + tuple(srcs)]
        newest = max(ents, key=lambda e: e.seq)
        if newest.seq > self._lane_seq[slot]:       # the producer stream is in order: its newest event covers the older ones
            lane.wait_event(newest.event)
            self._lane_seq[slot] = newest.seq
        img, _, pm, dv = self.graph.inputs[slot]
        self._image(scan, ref, img)
        if proj_matrices is not None:
            for k in pm:
                pm[k].copy_(proj_matrices[k], non_blocking=True)
        if depth_values is not None:
            dv.copy_(depth_values, non_blocking=True)
        self.tables[slot].set([[e.maps[s] for e in ents] for s in range(len(self.tables[slot].shapes))])
        out = self.graph.replay(slot)
        return out["depth"][-1][0], out["photometric_confidence"][0]


class _CacheEntry:
    __slots__ = ("maps", "event", "seq")

    def __init__(self, maps, event, seq):
        self.maps, self.event, self.seq = maps, event, seq
