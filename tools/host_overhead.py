#!/usr/bin/env python3
"""How long does the HOST need to enqueue one hot-path step (Python + ctypes launches), vs the GPU time?"""
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
        net.forward_hot(feats, ctx, pm, dv)
    torch.cuda.synchronize()
    n = 20
    t0 = time.perf_counter()
    for _ in range(n):
        net.forward_hot(feats, ctx, pm, dv)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
print(f"host enqueue {1e3 * (t1 - t0) / n:.3f} ms/step, total {1e3 * (t2 - t0) / n:.3f} ms/step")
import cProfile, pstats
pr = cProfile.Profile()
with torch.no_grad():
    pr.enable()
    for _ in range(5):
        net.forward_hot(feats, ctx, pm, dv)
    pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(25)
