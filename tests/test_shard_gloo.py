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
