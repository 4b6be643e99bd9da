#!/usr/bin/env python3
"""Micro-benchmark of the stride-1 3-D convolutions of the regulariser at cfg3 shapes (GPU)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from effi_mvs_plus_amd import ops  # noqa: E402
from effi_mvs_plus_amd.models.module import Conv3d  # noqa: E402

dev = "cuda:0"
shapes = [("st1 conv1 8->8", (8,), 8, (48, 148, 200)), ("st1 conv3 16->16", (16,), 16, (24, 74, 100)),
          ("csp st2 [8+8]->8", (8, 8), 8, (8, 148, 200)), ("csp st3 [8+8]->8", (8, 8), 8, (8, 296, 400))]
g = torch.Generator().manual_seed(0)
for name, cins, cout, dims in shapes:
    m = Conv3d(sum(cins), cout, padding=1).eval().to(dev)
    xs = [torch.randn(c, *dims, generator=g).to(dev) for c in cins]
    line = f"{name:18s}"
    for mode in ("split", "fp32"):
        ops.set_precision(mode)
        for _ in range(3):
            m.run(xs)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            m.run(xs)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        vox = dims[0] * dims[1] * dims[2]
        gb = 4.0 * vox * (sum(cins) + cout) / 1e9
        line += f"  {mode}: {us:7.1f} us ({gb / us * 1e6:6.0f} GB/s algorithmic)"
    print(line)
