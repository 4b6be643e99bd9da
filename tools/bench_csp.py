#!/usr/bin/env python3
"""Cross-scale blocks of a stage (CSP_R | CSP_C as a pair) at cfg3's shapes: the three launches conv0 | conv_cost | conv1 against the
generated-input kernel (option csp_gen), then the whole block pair.  us per call."""
import contextlib
import io
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from effi_mvs_plus_amd import ops, synth  # noqa: E402
from effi_mvs_plus_amd.models.module import cost_up_small  # noqa: E402

dev = "cuda:0"
ops.set_precision("split")


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


blocks = []
for seed in (5, 6):
    with contextlib.redirect_stdout(io.StringIO()):
        m = cost_up_small(in_channels=1, base_channels=8).eval()
    m.load_state_dict(synth.randomize_state_dict(m.state_dict(), seed=seed))
    blocks.append(m.to(dev))
a, b = blocks
g = torch.Generator().manual_seed(0)
for tag, D, h, w in (("stage 2 (8x148x200 coarse)", 8, 148, 200), ("stage 3 (8x296x400 coarse)", 8, 296, 400)):
    x = torch.randn(1, D, 2 * h, 2 * w, generator=g).to(dev)
    pa, pb = torch.randn(1, D, h, w, generator=g).to(dev), torch.randn(1, D, h, w, generator=g).to(dev)
    (w0a, b0a), (w0b, b0b) = a.conv0._packed(), b.conv0._packed()
    (wca, bca), (wcb, bcb) = a.conv_cost._packed(), b.conv_cost._packed()
    (w1a, b1a), (w1b, b1b) = a._roll_packed(), b._roll_packed()

    def three():
        fa, fb = ops.conv3d_k3_pair(x, w0a, b0a, x, w0b, b0b, 8, sxy=2, relu=True)
        ga, gb = ops.conv3d_k3_pair(pa, wca, bca, pb, wcb, bcb, 8, sxy=1, relu=True)
        return ops.conv3d_k3s1_roll_pair([fa, ga], w1a, b1a, [fb, gb], w1b, b1b, 8, relu=True)

    t3 = timed(three)
    t1 = timed(lambda: ops.csp_gen_roll_pair(x, pa, w0a, b0a, wca, bca, w1a, b1a, pb, w0b, b0b, wcb, bcb, w1b, b1b))
    ops.set_option("csp_gen", 0)
    p0 = timed(lambda: cost_up_small.run_pair(a, b, x, pa, pb))
    ops.set_option("csp_gen", 1)
    p1 = timed(lambda: cost_up_small.run_pair(a, b, x, pa, pb))
    print(f"{tag}: conv0|conv_cost|conv1 three launches {t3:6.1f} us, generated-input kernel {t1:6.1f} us; block pair {p0:6.1f} -> {p1:6.1f} us")
