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
    for mode in (("split",) if os.environ.get("EFFI_BENCH_SPLIT_ONLY") else ("split", "fp32")):
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

# transposed convolutions of the stage-1 U-Net (stride 2, with the skip tensor added after the ReLU)
from effi_mvs_plus_amd.models.module import Deconv3d  # noqa: E402
ops.set_precision("split")
for name, cin, cout, dims in [("st1 deconv 32->16", 32, 16, (12, 37, 50)), ("st1 deconv 16->8", 16, 8, (24, 74, 100))]:
    m = Deconv3d(cin, cout, stride=2, padding=1, output_padding=1).eval().to(dev)
    x = torch.randn(cin, *dims, generator=g).to(dev)
    skip = torch.randn(cout, 2 * dims[0], 2 * dims[1], 2 * dims[2], generator=g).to(dev)
    for _ in range(3):
        m.run(x, skip)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        m.run(x, skip)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    vox = dims[0] * dims[1] * dims[2]
    gb = 4.0 * vox * (cin + 16 * cout) / 1e9
    print(f"{name:18s}  split: {us:7.1f} us ({gb / us * 1e6:6.0f} GB/s algorithmic)")
