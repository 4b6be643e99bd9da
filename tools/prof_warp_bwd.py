#!/usr/bin/env python3
"""Forward + backward of the stage-1 warp + correlation (autograd.warp_correlate) at the cfg3 stage-1 shape, a few times:
run under `rocprofv3 --kernel-trace --stats` to see which kernels the 'warp_correlate_fwd_bwd' figure of bench.py is made of."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from effi_mvs_plus_amd import autograd as A, ops, synth  # noqa: E402

DEV = "cuda:0"


def main():
    h, w, D, N, (H, W) = 148, 200, 48, 5, (1184, 1600)
    feats = [f[0].to(DEV).contiguous() for f in synth.smooth_features(N, 32, h, w, seed=1)]
    pm = synth.synth_cameras(H, W, N)["stage1"][0].to(DEV).contiguous()
    hyp = (1.0 / torch.linspace(1 / 935.0, 1 / 425.0, D)).to(DEV)
    gsim = torch.randn(N - 1, D, h, w, device=DEV)
    reps = int(os.environ.get("REPS", "5"))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for i in range(reps + 2):
        if i == 2:
            e0.record()
        leaves = [x.detach().requires_grad_(True) for x in feats]
        sim = A.warp_correlate(leaves[0], leaves[1:], pm, hyp)
        sim.backward(gsim)
    e1.record()
    torch.cuda.synchronize()
    print(f"forward + backward: {e0.elapsed_time(e1) / reps:.3f} ms  (EFFI_MVS_LIB={os.environ.get('EFFI_MVS_LIB', '')})")


if __name__ == "__main__":
    main()
