#!/usr/bin/env python3
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from common import build_model
from effi_mvs_plus_amd import ops, synth
dev = "cuda:0"
net, sd = build_model("48,8,8", seed=1, device=dev)
imgs, pm, dv = synth.synth_sample(1184, 1600, 5, seed=0)
img = imgs[:, 0].to(dev)
with torch.no_grad():
    for name, mod in (("feature", net.feature), ("cnet_depth", net.cnet_depth)):
        for _ in range(3):
            mod(img)
        recs = []
        orig = ops._call
        def spy(key, work, fn, *args):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); rc = fn(*args); e1.record()
            recs.append((key, work(), e0, e1, args))
            return rc
        ops._call = spy
        mod(img)
        ops._call = orig
        torch.cuda.synchronize()
        tot = 0
        print(name)
        for key, w, e0, e1, args in recs:
            ms = e0.elapsed_time(e1); tot += ms
            print(f"  {key:22s} {ms*1e3:8.1f} us  {w['flops']/ms/1e9:7.1f} TFLOP/s  {w['bytes']/ms/1e6:7.1f} GB/s")
        print(f"  total {tot*1e3:.1f} us")
