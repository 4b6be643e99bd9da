#!/usr/bin/env python3
"""Time the stage-1 warp + correlation kernel (effi_warpcorr_views_f32) at the benchmark shapes, for each EFFI_WARP_LDS_KB
setting given on the command line (default: unset = LDS window, 0 = same kernel on global loads, -1 = direct-gather kernel)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from effi_mvs_plus_amd import ops, synth  # noqa: E402

DEV = "cuda:0"
SHAPES = {"cfg3": (148, 200, 48, 5, (1184, 1600)), "cfg4": (132, 240, 96, 7, (1056, 1920)), "cfg2": (72, 100, 48, 5, (576, 800))}


def main():
    modes = sys.argv[1:] or ["", "0", "-1"]
    only = os.environ.get("SHAPES")
    for name, (h, w, D, N, (H, W)) in SHAPES.items():
        if only and name not in only.split(","):
            continue
        feats = synth.smooth_features(N, 32, h, w, seed=1)
        pm = synth.synth_cameras(H, W, N)["stage1"]
        nhwc = ops.to_nhwc([f[0].to(DEV).contiguous() for f in feats])
        rt = ops.compose_rel_proj(pm[0].to(DEV).contiguous())
        hyp = (1.0 / torch.linspace(1 / 935.0, 1 / 425.0, D)).to(DEV)
        outs = {}
        for mode in modes:
            ops.set_option("warp_lds_kb", None if mode == "" else int(mode))
            for _ in range(3):
                sim, ent = ops.warpcorr_views(nhwc[0], nhwc[1:], rt, hyp, D)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n = 50
            e0.record()
            for _ in range(n):
                sim, ent = ops.warpcorr_views(nhwc[0], nhwc[1:], rt, hyp, D)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / n * 1e3
            outs[mode] = sim
            alg = 4.0 * h * w * ((N - 1) * 32 + 32 + (N - 1) * D + (N - 1))
            print(f"{name} {h}x{w} D={D} S={N - 1} EFFI_WARP_LDS_KB={mode or 'unset':5s}: {us:8.1f} us  "
                  f"({alg / us / 1e3:.0f} GB/s algorithmic, {(N - 1) * D * h * w * 512 / us / 1e3:.0f} GB/s of taps)")
        for prec in ("split", "bf16"):                      # the matrix-core form (correlate first: effi_warpcorr_views_x3_f32)
            ops.set_option("warp_lds_kb", None)
            ops.set_precision(prec)
            for _ in range(3):
                sim, ent = ops.warpcorr_views(nhwc[0], nhwc[1:], rt, hyp, D, x3=True)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50):
                sim, ent = ops.warpcorr_views(nhwc[0], nhwc[1:], rt, hyp, D, x3=True)
            e1.record()
            torch.cuda.synchronize()
            ops.set_precision("split")
            msg = f"  max |x3 - window| = {float((sim - outs[''][0 if isinstance(outs[''], tuple) else slice(None)]).abs().max()):.3e}" if "" in outs else ""
            print(f"{name} {h}x{w} D={D} S={N - 1} matrix-core form ({prec:5s}): {e0.elapsed_time(e1) / 50 * 1e3:8.1f} us{msg}")
        if "" in outs and "0" in outs:
            print("   window == global path bitwise:", bool(torch.equal(outs[""], outs["0"])))
        if "" in outs and "-1" in outs:
            print("   max |window - direct| =", float((outs[""] - outs["-1"]).abs().max()))


if __name__ == "__main__":
    main()
