#!/usr/bin/env python3
"""Time the stock PyTorch-ROCm FPN (out of scope) next to the HIP hot path at cfg3."""
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
imgs = imgs.to(dev)
pm = {k: v.to(dev) for k, v in pm.items()}
dv = dv.to(dev)


def fpn():
    feats = [net.feature(imgs[:, v]) for v in range(N)]
    return feats, net.cnet_depth(imgs[:, 0])


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


with torch.no_grad():
    for cl in (False, True):
        if cl:
            net.feature = net.feature.to(memory_format=torch.channels_last)
            net.cnet_depth = net.cnet_depth.to(memory_format=torch.channels_last)
            imgs = imgs.contiguous()
        torch.backends.cudnn.benchmark = True
        print(f"channels_last={cl}: FPN (5 feature nets + context net) {timeit(fpn):.2f} ms")
    feats, ctx = fpn()
    print(f"hot path {timeit(lambda: net.forward_hot(feats, ctx, pm, dv)):.2f} ms;  full forward {timeit(lambda: net(imgs, pm, dv)):.2f} ms")
    print("stage1 feature strides", feats[0]["stage1"].stride(), "contig", feats[0]["stage1"].is_contiguous())
