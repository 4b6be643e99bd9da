#!/usr/bin/env python3
"""tests/test_gpu_model.py::test_views_in_flight_on_two_stream_graphs_are_bitwise_equal_to_eager with knobs (bisection of an in-flight
difference): --branches 0/1 (two internal streams in the captured pass), --lanes 1/3 (views in flight), --rounds."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

from common import build_model  # noqa: E402
from effi_mvs_plus_amd import ops, synth  # noqa: E402
from effi_mvs_plus_amd.graph import HotPathGraph  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--branches", type=int, default=1)
ap.add_argument("--lanes", type=int, default=3)
ap.add_argument("--rounds", type=int, default=25)
ap.add_argument("--size", default="192x256")
ap.add_argument("--gemm", type=int, default=0, help="N > 0: a fourth stream runs N bf16 GEMMs (2048^3, hipBLASLt) per round next to the views in flight")
ap.add_argument("--ndepths", default="8,8,8")
ap.add_argument("--inter", type=int, default=0, help="1: capture forward_hot(want_intermediates=True) and compare the stage volumes too")
args = ap.parse_args()
DEV = "cuda:0"
H, W = [int(v) for v in args.size.split("x")]
net, sd = build_model(args.ndepths, seed=6, device=DEV)
samples = []
with torch.no_grad():
    for seed in (31, 32, 33):
        imgs, pm, dv = synth.synth_sample(H, W, 3, seed=seed)
        imgs = imgs.to(DEV)
        feats = [net.feature(imgs[:, v]) for v in range(3)]
        ctx = net.cnet_depth(imgs[:, 0])
        samples.append((feats, ctx, {k: v.to(DEV) for k, v in pm.items()}, dv.to(DEV)))
    from effi_mvs_plus_amd.graph import ReplayGraph

    def fn(f, c, p, d):
        o = net.forward_hot(f, c, p, d, want_intermediates=bool(args.inter))
        res = list(o["depth"])
        if args.inter:
            res += [o["intermediates"][k] for k in sorted(o["intermediates"])]
        return {"depth": res}

    names = [f"depth[{i}]" for i in range(13)] + (sorted(net.forward_hot(*samples[0], want_intermediates=True)["intermediates"]) if args.inter else [])
    want = [[d.clone() for d in fn(*smp)["depth"]] for smp in samples]
    ops.set_branches(bool(args.branches))
    try:
        g = ReplayGraph(fn, samples[0], slots=3)
    finally:
        ops.set_branches(False)
    for i, smp in enumerate(samples):
        g.load(i, *smp)
    torch.cuda.synchronize()
    lanes = [torch.cuda.Stream() for _ in range(args.lanes)]
    cur = torch.cuda.current_stream()
    for st in lanes:
        st.wait_stream(cur)
    bad, first = {}, None
    gg = torch.Generator().manual_seed(5)
    GA = torch.randn(2048, 2048, generator=gg).to(DEV).bfloat16()
    GB = torch.randn(2048, 2048, generator=gg).to(DEV).bfloat16()
    gemm_stream = torch.cuda.Stream()
    for rnd_ in range(args.rounds):
        kept = []
        with torch.cuda.stream(gemm_stream):
            for _ in range(args.gemm):
                GA @ GB
        for i in range(12):
            with torch.cuda.stream(lanes[i % args.lanes]):
                out = g.replay(i % 3)
                kept.append((i % 3, [d.clone() for d in out["depth"]]))
        for st in lanes:
            cur.wait_stream(st)
        torch.cuda.synchronize()
        for slot, depths in kept:
            for k, (a, b) in enumerate(zip(depths, want[slot])):
                if not torch.equal(a, b):
                    bad[(slot, k)] = bad.get((slot, k), 0) + 1
                    if first is None or names[k].startswith("reg_volume1") or names[k].startswith("cur_volume1"):
                        d = (a != b).nonzero()
                        first = (first or "") + f" || round {rnd_} slot {slot} {names[k]} shape {tuple(a.shape)}: {d.shape[0]} px differ, dims {[(int(d[:, i].min()), int(d[:, i].max())) for i in range(d.shape[1])]}, max abs {(a - b).abs().max().item():.3e}"
        for st in lanes:
            st.wait_stream(cur)
import sys as _s
print(f"size {args.size} ndepths {args.ndepths} branches {args.branches} lanes {args.lanes} gemms/round {args.gemm} rounds {args.rounds}: differing (slot, depth) counts: {len(bad)}; starts: {sorted(set(min(k for (s, k) in bad if s == sl) for sl in set(s for s, _ in bad)))}; {first}")
_s.exit(1 if bad else 0)
