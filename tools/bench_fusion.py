#!/usr/bin/env python3
"""Time the depth-fusion filter kernel (scope row n3) at 1600x1184 with 10 source views."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from effi_mvs_plus_amd import ops, synth
dev = "cuda:0"
H, W, V = 1184, 1600, 10
d, cams = synth.synth_depth_maps(H, W, V + 1, seed=4)
d, cams = d.to(dev), cams.to(dev)
conf = torch.rand(H, W, device=dev)
args = (d[0].contiguous(), d[1:].contiguous(), cams[0].contiguous(), cams[1:].contiguous(), conf, 0.3, 2, 4.0, 1.3)
for _ in range(3):
    ops.fusion_dynamic_filter(*args)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    r = ops.fusion_dynamic_filter(*args)
torch.cuda.synchronize()
print(f"fusion filter {W}x{H} V={V}: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms per reference view; mask mean {float(r['mask'].float().mean()):.4f}")
