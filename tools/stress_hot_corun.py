#!/usr/bin/env python3
"""The whole hot path next to hostile co-runners: forward_hot (eager, or a captured graph) on one stream while two more streams run
matrix-core GEMMs (hipBLASLt) back to back; every pass's 13 depth maps + confidence compared bit for bit with a quiet pass.
A kernel whose result depends on what shares its CU shows up here (round 4: the compiler's packed-FMA form of conv3d_c8to1_kernel did).
Usage: stress_hot_corun.py [--size 192x256] [--passes 60] [--ndepths 8,8,8] [--graph 1]"""
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
ap.add_argument("--size", default="192x256")
ap.add_argument("--passes", type=int, default=60)
ap.add_argument("--ndepths", default="8,8,8")
ap.add_argument("--views", type=int, default=3)
ap.add_argument("--graph", type=int, default=0)
ap.add_argument("--gemm", type=int, default=2048)
args = ap.parse_args()
DEV = "cuda:0"
H, W = [int(v) for v in args.size.split("x")]
net, _ = build_model(args.ndepths, seed=6, device=DEV)
g = torch.Generator().manual_seed(1)
A = torch.randn(args.gemm, args.gemm, generator=g).to(DEV).bfloat16()
B = torch.randn(args.gemm, args.gemm, generator=g).to(DEV).bfloat16()
with torch.no_grad():
    imgs, pm, dv = synth.synth_sample(H, W, args.views, seed=31)
    imgs = imgs.to(DEV)
    feats = [net.feature(imgs[:, v]) for v in range(args.views)]
    ctx = net.cnet_depth(imgs[:, 0])
    smp = (feats, ctx, {k: v.to(DEV) for k, v in pm.items()}, dv.to(DEV))
    ref = net.forward_hot(*smp)
    want = [d.clone() for d in ref["depth"]] + [ref["photometric_confidence"].clone()]
    runner = None
    if args.graph:
        runner = HotPathGraph(net, *smp, slots=1)
        runner.load(0, *smp)
    torch.cuda.synchronize()
    s0, s1, s2 = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
    bad, names = {}, [f"depth[{i}]" for i in range(len(want) - 1)] + ["confidence"]
    for it in range(args.passes):
        for _ in range(40):
            with torch.cuda.stream(s1):
                A @ B
            with torch.cuda.stream(s2):
                B @ A
        with torch.cuda.stream(s0):
            out = runner.replay(0) if runner else net.forward_hot(*smp)
            got = [d.clone() for d in out["depth"]] + [out["photometric_confidence"].clone()]
        torch.cuda.synchronize()
        for k, (a, b) in enumerate(zip(got, want)):
            if not torch.equal(a, b):
                bad.setdefault(names[k], []).append(it)
first = sorted(bad, key=lambda n: names.index(n))[:1]
print(f"hot path {args.size} ndepths {args.ndepths} next to GEMMs ({'graph' if args.graph else 'eager'}): {args.passes} passes, "
      f"maps that ever differed: {len(bad)}; first in path order: {first}; passes with a difference: {sorted(set(i for v in bad.values() for i in v))[:20]}")
sys.exit(1 if bad else 0)
