#!/usr/bin/env python3
"""Does capturing the hot path into a HIP graph (torch.cuda.CUDAGraph) work with the ctypes-launched kernels,
and what does replay cost vs eager launches?"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from common import build_model  # noqa: E402
from effi_mvs_plus_amd import synth  # noqa: E402

dev = "cuda:0"
H, W, N = 1184, 1600, 5
net, sd = build_model("48,8,8", seed=1, device=dev)
imgs, pm, dv = synth.synth_sample(H, W, N, seed=0)
with torch.no_grad():
    imgs = imgs.to(dev)
    feats = [net.feature(imgs[:, v]) for v in range(N)]
    ctx = net.cnet_depth(imgs[:, 0])
    pm = {k: v.to(dev) for k, v in pm.items()}
    dv = dv.to(dev)
    for _ in range(3):
        ref = net.forward_hot(feats, ctx, pm, dv)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        net.forward_hot(feats, ctx, pm, dv)          # warm-up on the side stream
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g):
        out = net.forward_hot(feats, ctx, pm, dv)
    torch.cuda.synchronize()
    g.replay()
    torch.cuda.synchronize()
    ok = all(torch.equal(a, b) for a, b in zip(out["depth"], ref["depth"]))
    print("graph replay bitwise equal to eager:", ok)
    for name, fn in (("eager", lambda: net.forward_hot(feats, ctx, pm, dv)), ("graph", g.replay)):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            fn()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"{name}: host {1e3 * (t1 - t0) / 20:.3f} ms/step, total {1e3 * (t2 - t0) / 20:.3f} ms/step")
