"""Multi-process (world_size 2, gloo, CPU) test of the view sharding + final gather used for N > 1."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from effi_mvs_plus_amd import shard


def test_shard_bounds_cover_everything_once():
    for n in (1, 7, 49, 1078):
        for world in (1, 2, 3, 8):
            if world > n:
                continue
            seen = []
            for r in range(world):
                lo, hi = shard.shard_bounds(n, r, world)
                assert hi - lo in (n // world, n // world + 1)
                seen += list(range(lo, hi))
            assert seen == list(range(n))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_items, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        items = list(range(n_items))

        def forward(i):            # stand-in for the per-view cascade: maps that encode the view index
            return torch.full((6, 8), float(i)), torch.full((3, 4), float(i) + 0.5)

        res = shard.run_sharded(items, forward, dst=0)
        if rank == 0:
            torch.save(res, out_path)
        else:
            assert res is None
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_items", [4, 7])
def test_run_sharded_two_ranks_gloo(tmp_path, n_items):
    out = str(tmp_path / "res.pt")
    mp.spawn(_worker, args=(2, _free_port(), n_items, out), nprocs=2, join=True)
    res = torch.load(out)
    assert res["depth"].shape == (n_items, 6, 8) and res["confidence"].shape == (n_items, 3, 4)
    for i in range(n_items):                      # view order is preserved, uneven shards are un-padded
        assert torch.all(res["depth"][i] == float(i)) and torch.all(res["confidence"][i] == float(i) + 0.5)


# ---- cfg5: whole evaluation set -- uneven shards, per-scan feature cache, batched asynchronous gather, REAL (tiny) forwards ---------
def _scan_image(scan, image, H=64, W=96):
    g = torch.Generator().manual_seed(1000 * scan + image)
    return torch.rand(1, 3, H, W, generator=g)


def _tiny_eval_setup():
    """Weights + rig shared by every rank (seeded): the CPU oracle is the forward (tests may use it; the product path has no CPU form)."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__))))
    from common import build_model
    from effi_mvs_plus_amd import synth
    from oracle import effi_oracle as O
    _, sd = build_model("8,8,8", seed=21)
    _, pm, dv = synth.synth_sample(64, 96, 3, seed=0)
    return O, sd, pm, dv


def _eval_forward_factory(O, sd, pm, dv, stats):
    from effi_mvs_plus_amd import scan_eval
    feats = scan_eval.ScanFeatureCache(lambda s, i: O.feature_net(sd, "feature", _scan_image(s, i)))
    stats["cache"] = feats

    def forward(item):
        scan, ref, srcs = item
        with torch.no_grad():
            f = [feats.get(scan, ref)] + [feats.get(scan, v) for v in srcs]
            ctx = O.feature_net(sd, "cnet_depth", _scan_image(scan, ref))
            out = O.hot_path(sd, f, ctx, pm, dv, ndepths=(8, 8, 8))
        return out["depth"][-1][0], out["photometric_confidence"][0]
    return forward


def _eval_worker(rank, world, port, n_scans, n_images, k, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from effi_mvs_plus_amd import scan_eval
        O, sd, pm, dv = _tiny_eval_setup()
        items = scan_eval.build_items(n_scans, n_images, 2)
        stats = {}
        res, n_mine = scan_eval.run_scans(items, _eval_forward_factory(O, sd, pm, dv, stats), gather_batch=k, dst=0, to_host=True)
        c = stats["cache"]
        torch.save({"res": res, "n": n_mine, "hits": c.hits, "misses": c.misses, "max_entries": c.max_entries}, f"{out_path}.{rank}")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_scans,n_images,k", [(1, 5, 2), (3, 3, 4)])
def test_whole_set_two_ranks_uneven_shards_batched_gather_real_forwards(tmp_path, n_scans, n_images, k):
    """5 / 9 items over 2 ranks (shards 3 + 2 / 5 + 4), batches of 2 / 4 views (a padded last batch; with k = 4 the shorter shard
    issues a different number of FULL batches than the longer one), forwards = the CPU oracle on 96x64 images: the gathered maps
    equal a single-process evaluation of every item bit for bit, in item order; an image's pyramid is computed once per rank and scan."""
    from effi_mvs_plus_amd import scan_eval
    out = str(tmp_path / "eval")
    mp.spawn(_eval_worker, args=(2, _free_port(), n_scans, n_images, k, out), nprocs=2, join=True)
    r0, r1 = torch.load(out + ".0"), torch.load(out + ".1")
    items = scan_eval.build_items(n_scans, n_images, 2)
    assert r0["n"] + r1["n"] == len(items) and r1["res"] is None
    O, sd, pm, dv = _tiny_eval_setup()
    stats = {}
    fwd = _eval_forward_factory(O, sd, pm, dv, stats)
    assert r0["res"]["depth"].shape[0] == len(items) and r0["res"]["confidence"].shape[0] == len(items)
    threads = torch.get_num_threads()
    torch.set_num_threads(2)                      # as the workers: ATen's CPU reductions depend on the thread count in the last bit
    try:
        for i, it in enumerate(items):
            d, c = fwd(it)
            assert torch.equal(r0["res"]["depth"][i], d), f"item {i} {it}"
            assert torch.equal(r0["res"]["confidence"][i], c)
        for r in (r0, r1):
            assert r["misses"] <= n_images * n_scans and r["hits"] > 0 and r["max_entries"] <= n_images
        # single process, same API (no process group): one rank computes everything
        res, n = scan_eval.run_scans(items[:3], fwd, gather_batch=2)
        assert n == 3 and torch.equal(res["depth"], r0["res"]["depth"][:3])
    finally:
        torch.set_num_threads(threads)


def test_synthetic_pairs_and_cache_bookkeeping():
    from effi_mvs_plus_amd import scan_eval
    pairs = scan_eval.synthetic_pairs(49, 4)
    assert len(pairs) == 49 and all(len(set(s)) == 4 and r not in s for r, s in pairs)
    assert pairs[0] == (0, (1, 48, 2, 47))
    items = scan_eval.build_items(22, 49, 4)
    assert len(items) == 1078 and items[49][0] == 1                 # DTU test set: 22 scans x 49 reference views (lists/dtu/test.txt)
    calls = []
    cache = scan_eval.ScanFeatureCache(lambda s, i: calls.append((s, i)) or (s, i))
    for s, ref, srcs in items[:98]:
        for v in (ref,) + srcs:
            assert cache.get(s, v) == (s, v)
    assert len(calls) == 98 and cache.hits == 98 * 5 - 98 and cache.max_entries == 49      # every image once per scan
    with pytest.raises(ValueError):
        scan_eval.BatchedGather(0, 4)
