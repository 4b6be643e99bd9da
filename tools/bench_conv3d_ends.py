#!/usr/bin/env python3
"""The one-channel ends of the 3-D blocks at cfg3's shapes (U-Net conv0 1->8 / prob 8->1; cross-scale conv0 | conv_cost pairs 1->8 and the
transposed conv2 pair 8->1): us per launch and algorithmic GB/s.  For A/B runs of two builds (EFFI_MVS_LIB)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from effi_mvs_plus_amd import ops  # noqa: E402

dev = "cuda:0"
g = torch.Generator().manual_seed(0)
rnd = lambda *s: torch.randn(*s, generator=g).to(dev)


def timed(fn, n=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def row(name, fn, mbytes):
    us = timed(fn)
    print(f"{name:46s} {us:7.1f} us   {mbytes / us:6.2f} TB/s algorithmic")


D, h, w = 48, 148, 200
x1, w18, b8 = rnd(1, D, h, w), rnd(1, 27, 8) * 0.2, rnd(8)
row("U-Net conv0 1->8 (48x148x200)", lambda: ops.conv3d_k3([x1], w18, b8, 8, relu=True), 4e-6 * D * h * w * 9)
x8, w81 = rnd(8, D, h, w), rnd(8, 27, 1) * 0.1
row("U-Net prob 8->1", lambda: ops.conv3d_k3([x8], w81, None, 1, relu=False), 4e-6 * D * h * w * 9)
for tag, D, h, w in (("stage 2 (8x296x400)", 8, 296, 400), ("stage 3 (8x592x800)", 8, 592, 800)):
    xa, xb = rnd(1, D, h, w), rnd(1, D, h, w)
    wa, wb, ba, bb = rnd(1, 27, 8) * 0.2, rnd(1, 27, 8) * 0.2, rnd(8), rnd(8)
    row(f"{tag} conv0 pair 1->8 s(1,2,2)", lambda: ops.conv3d_k3_pair(xa, wa, ba, xb, wb, bb, 8, sxy=2), 2 * 4e-6 * D * h * w * (1 + 2))
    pa, pb = rnd(1, D, h // 2, w // 2), rnd(1, D, h // 2, w // 2)
    row(f"{tag} conv_cost pair 1->8", lambda: ops.conv3d_k3_pair(pa, wa, ba, pb, wb, bb, 8, sxy=1), 2 * 4e-6 * D * h * w / 4 * 9)
    ca, cb = rnd(8, D, h // 2, w // 2), rnd(8, D, h // 2, w // 2)
    wda, wdb, bda, bdb = rnd(8, 27, 1) * 0.1, rnd(8, 27, 1) * 0.1, rnd(1), rnd(1)
    row(f"{tag} conv2 transposed pair 8->1", lambda: ops.deconv3d_k3_pair(ca, wda, bda, cb, wdb, bdb, 1, sz=1), 2 * 4e-6 * D * h * w * (2 + 1))
