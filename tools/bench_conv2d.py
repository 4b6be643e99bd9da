#!/usr/bin/env python3
"""Micro-benchmark of the GRU convolution shapes at cfg3 (GPU).  Usage: EFFI_CONV2D_VARIANT=k python tools/bench_conv2d.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from effi_mvs_plus_amd import ops, packing  # noqa: E402

dev = "cuda:0"
shapes = [  # (name, h, w, cins, cout, ks, epi)
    ("st3 16->16", 592, 800, (16,), 16, 3, 0), ("st3 32->12", 592, 800, (16, 16), 12, 3, 0), ("st3 zr 32->32", 592, 800, (16, 16), 32, 3, 1),
    ("st3 q 32->16", 592, 800, (16, 16), 16, 3, 2), ("st3 k1 16->16", 592, 800, (12, 4), 16, 1, 0),
    ("st2 32->32", 296, 400, (32,), 32, 3, 0), ("st2 zr 64->64", 296, 400, (32, 32), 64, 3, 1), ("st2 q 64->32", 296, 400, (32, 32), 32, 3, 2),
    ("st1 48->48", 148, 200, (48,), 48, 3, 0), ("st1 zr 96->96", 148, 200, (48, 48), 96, 3, 1), ("st1 q 96->48", 148, 200, (48, 48), 48, 3, 2),
]
g = torch.Generator().manual_seed(0)
print("variant", os.environ.get("EFFI_CONV2D_VARIANT", "0"))
for name, h, w, cins, cout, ks, epi in shapes:
    xs = [torch.randn(c, h, w, generator=g).to(dev) for c in cins]
    cin = sum(cins)
    wt = (torch.randn(cout, cin, ks, ks, generator=g) * 0.05).to(dev)
    b = torch.zeros(cout).to(dev)
    wp, bp = packing.pack_conv2d_mfma(wt, b)
    hd = cout // 2 if epi == 1 else cout
    aux0 = torch.randn(hd, h, w, generator=g).to(dev)
    aux1 = torch.rand(hd, h, w, generator=g).to(dev)
    kw = dict(epilogue=epi, act=1, aux0=aux0 if epi else None, aux1=aux1 if epi == 2 else None)
    for _ in range(3):
        out = ops.conv2d(xs, wp, bp, cout, ks, **kw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 30
    e0.record()
    for _ in range(n):
        ops.conv2d(xs, wp, bp, cout, ks, **kw)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / n * 1e3
    fl = 2.0 * h * w * cin * cout * ks * ks
    line = f"{name:16s} {us:8.1f} us  {fl / us / 1e6:7.1f} TFLOP/s"
    if ks == 3 and w % 4 == 0:
        wx, bx = packing.pack_conv2d_bf16x3(wt, b)
        for _ in range(3):
            outx = ops.conv2d_k3_bf16x3(xs, wx, bx, cout, **kw)
        torch.cuda.synchronize()
        # EFFI_ROTATE=R: cycle through R distinct input / output buffer sets so the working set exceeds the 256 MB MALL
        # (in the pipeline a layer's input was written several hundred MB of traffic earlier)
        R = int(os.environ.get("EFFI_ROTATE", "1"))
        xsets = [xs] + [[x.clone() for x in xs] for _ in range(R - 1)]
        outs = [torch.empty_like(outx[0] if isinstance(outx, tuple) else outx) for _ in range(R)]
        outs1 = [torch.empty_like(outx[1]) for _ in range(R)] if isinstance(outx, tuple) else [None] * R
        for r_ in range(R):
            ops.conv2d_k3_bf16x3(xsets[r_], wx, bx, cout, out0=outs[r_], out1=outs1[r_], **kw)
        torch.cuda.synchronize()
        e0.record()
        for i_ in range(n):
            ops.conv2d_k3_bf16x3(xsets[i_ % R], wx, bx, cout, out0=outs[i_ % R], out1=outs1[i_ % R], **kw)
        e1.record()
        torch.cuda.synchronize()
        usx = e0.elapsed_time(e1) / n * 1e3
        o0 = out[0] if isinstance(out, tuple) else out
        x0 = outx[0] if isinstance(outx, tuple) else outx
        err = (o0 - x0).abs().max().item() / max(o0.abs().max().item(), 1e-30)
        line += f"   | bf16x3 {usx:8.1f} us  {fl / usx / 1e6:7.1f} TFLOP/s  x{us / usx:4.2f}  max rel-to-peak diff {err:.2e}"
    print(line)
