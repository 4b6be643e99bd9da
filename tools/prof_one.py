#!/usr/bin/env python3
"""Run ONE convolution shape a few times (for rocprofv3 --pmc passes).  usage: prof_one.py {roll8|roll16|c2d16|c2dzr}"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from effi_mvs_plus_amd import ops, packing  # noqa: E402
from effi_mvs_plus_amd.models.module import Conv3d  # noqa: E402

dev = "cuda:0"
which = sys.argv[1] if len(sys.argv) > 1 else "roll8"
g = torch.Generator().manual_seed(0)
if which.startswith("roll"):
    cins, cout, dims = {"roll8": ((8,), 8, (48, 148, 200)), "roll16": ((8, 8), 8, (8, 296, 400))}[which]
    m = Conv3d(sum(cins), cout, padding=1).eval().to(dev)
    xs = [torch.randn(c, *dims, generator=g).to(dev) for c in cins]
    fn = lambda: m.run(xs)
else:
    h, w = 592, 800
    cins, cout, epi = {"c2d16": ((16,), 16, 0), "c2dzr": ((16, 16), 32, 1)}[which]
    xs = [torch.randn(c, h, w, generator=g).to(dev) for c in cins]
    wt = (torch.randn(cout, sum(cins), 3, 3, generator=g) * 0.05).to(dev)
    wx, bx = packing.pack_conv2d_bf16x3(wt, torch.zeros(cout, device=dev))
    aux0 = torch.randn(cout // 2 if epi == 1 else cout, h, w, generator=g).to(dev)
    fn = lambda: ops.conv2d_k3_bf16x3(xs, wx, bx, cout, epilogue=epi, act=1, aux0=aux0 if epi else None)
for _ in range(10):
    fn()
torch.cuda.synchronize()
