#!/usr/bin/env python3
"""Stress of the dedicated 8 -> 1 channel 3-D kernels under concurrency: the same launches on three streams at once, many times,
each result compared with the result of a quiet run; prints where the first differing elements sit."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from effi_mvs_plus_amd import ops  # noqa: E402

dev = "cuda:0"
g = torch.Generator().manual_seed(0)
rnd = lambda *s: torch.randn(*s, generator=g).to(dev)
cases = []
for (D, h, w) in [(8, 24, 32), (8, 48, 64), (8, 96, 128)]:
    x8 = [rnd(8, D, h, w) for _ in range(3)]
    w81, wd, bd = rnd(8, 27, 1) * 0.1, rnd(8, 27, 1) * 0.1, rnd(1)
    cases.append(("conv", x8, w81, None, (D, h, w)))
    cases.append(("deconv", x8, wd, bd, (D, h, w)))


def run(kind, x, wt, b):
    if kind == "conv":
        return ops.conv3d_k3([x], wt, b, 1, relu=False)
    return ops.deconv3d_k3_pair(x, wt, b, x, wt, b, 1, sz=1)[1]


want = {}
for ci, (kind, xs, wt, b, _) in enumerate(cases):
    for j, x in enumerate(xs):
        want[(ci, j)] = run(kind, x, wt, b).clone()
torch.cuda.synchronize()
lanes = [torch.cuda.Stream() for _ in range(3)]
filler = rnd(4096, 4096)
bad = 0
for it in range(200):
    outs = []
    for ci, (kind, xs, wt, b, _) in enumerate(cases):
        for j, x in enumerate(xs):
            with torch.cuda.stream(lanes[j]):
                outs.append((ci, j, run(kind, x, wt, b)))
                if it % 2:
                    filler.mul_(1.0)                     # another kernel in between
    torch.cuda.synchronize()
    for ci, j, o in outs:
        if not torch.equal(o, want[(ci, j)]):
            bad += 1
            d = (o != want[(ci, j)]).nonzero()
            if bad <= 6:
                print(f"iter {it} case {cases[ci][0]} {cases[ci][4]} lane {j}: {d.shape[0]} elements differ; first {d[:3].tolist()} last {d[-3:].tolist()}; "
                      f"max abs {(o - want[(ci, j)]).abs().max().item():.3e}; z set {sorted(set(d[:, 1].tolist()))[:10]}")
print("differing results:", bad, "of", 200 * len(cases) * 3)
