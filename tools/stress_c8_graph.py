#!/usr/bin/env python3
"""The dedicated 8 -> 1 channel kernels inside captured graphs replayed on three streams at once (the in-flight mode of bench.py),
each replay compared with a quiet eager run.  Chain per graph: producer (elementwise) -> kernel under test -> consumer."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from effi_mvs_plus_amd import ops  # noqa: E402

dev = "cuda:0"
g = torch.Generator().manual_seed(0)
rnd = lambda *s: torch.randn(*s, generator=g).to(dev)
D, h, w = 8, 24, 32
w81 = rnd(8, 27, 1) * 0.1
KIND = sys.argv[1] if len(sys.argv) > 1 else "conv"
CHAIN = int(sys.argv[2]) if len(sys.argv) > 2 else 4


def body(x):
    y = x
    for _ in range(CHAIN):                                   # a few dependent kernels around the one under test
        y = y * 1.0009765625 + 0.5
        if KIND == "conv":
            o = ops.conv3d_k3([y], w81, None, 1, relu=False)
        else:
            o = ops.deconv3d_k3_pair(y, w81, None, y, w81, None, 1, sz=1)[0]
        y = y + o.mean() * 0.0 + (o[:, :, :h, :w] if KIND != "conv" else o) * 0.125
    return y.clone()


xs = [rnd(8, D, h, w) for _ in range(3)]
want = [body(x).clone() for x in xs]
torch.cuda.synchronize()
graphs, outs = [], []
for x in xs:
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        body(x)
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=st):
            o = body(x)
    graphs.append(gr)
    outs.append(o)
torch.cuda.synchronize()
lanes = [torch.cuda.Stream() for _ in range(3)]
bad = 0
for it in range(300):
    kept = []
    for i in range(12):
        with torch.cuda.stream(lanes[i % 3]):
            graphs[i % 3].replay()
            kept.append((i % 3, outs[i % 3].clone()))
    torch.cuda.synchronize()
    for j, o in kept:
        if not torch.equal(o, want[j]):
            bad += 1
            if bad <= 4:
                d = (o != want[j]).nonzero()
                print(f"iter {it} lane {j}: {d.shape[0]} elements differ, first {d[0].tolist()} last {d[-1].tolist()} max abs {(o - want[j]).abs().max().item():.3e}")
print(f"{KIND} chain {CHAIN}: differing replays {bad} of {300 * 12}")
