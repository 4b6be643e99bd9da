#!/usr/bin/env python3
"""Stage-2/3 warp kernel (effi_warpcorr_dyn_f32): how large is the SOURCE window of a tile of reference pixels?

Runs the hot path once on the benchmark rig (bench.py's inputs), takes the depth map each of the stages 2 / 3 starts from, and
computes per tile of TWxTH reference pixels and per source view the bounding box of all (pixel, hypothesis) sampling positions
(+ the second bilinear tap) -- the window an LDS-staged form of the kernel would have to hold.  Prints the distribution of the
box sizes and the fraction of (tile, view) pairs that fit a given budget, then times the kernel standalone for each option
setting given in DYN_WIN (comma-separated values of the option dyn_win: '' = LDS-window kernel, -1 = gather kernel, 0 = window
kernel on global loads)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from common import build_model  # noqa: E402
from effi_mvs_plus_amd import ops, synth  # noqa: E402

DEV = "cuda:0"
WL = {"cfg3": (1184, 1600, 5, "48,8,8"), "cfg2": (576, 800, 5, "48,8,8"), "cfg4": (1056, 1920, 7, "48,8,8")}


def boxes(cur, rt, interval, D, tw, th):
    """cur [h,w] depth, rt [S,12] -> (ww, wh) int tensors [S, tiles]"""
    h, w = cur.shape
    ys, xs = torch.meshgrid(torch.arange(h, device=cur.device, dtype=torch.float32),
                            torch.arange(w, device=cur.device, dtype=torch.float32), indexing="ij")
    inv = 1.0 / cur
    half = (D // 2) * interval
    smin = torch.clamp(inv - half, min=1e-4)
    smax = torch.clamp(torch.clamp(inv + half, min=1e-4), max=1e4)
    step = (smax - smin) / (D - 1)
    out_w, out_h = [], []
    hp, wp = (h + th - 1) // th * th, (w + tw - 1) // tw * tw
    for v in range(rt.shape[0]):
        r = rt[v]
        rx = r[0] * xs + r[1] * ys + r[2]
        ry = r[3] * xs + r[4] * ys + r[5]
        rz = r[6] * xs + r[7] * ys + r[8]
        mnx = torch.full((hp, wp), 1e9, device=cur.device)
        mxx = torch.full((hp, wp), -1e9, device=cur.device)
        mny, mxy = mnx.clone(), mxx.clone()
        for d in range(D):
            dep = 1.0 / torch.clamp(smin + d * step, min=1e-5)
            Z = rz * dep + r[11]
            ix = torch.clamp((rx * dep + r[9]) / Z, -2.0, w + 1.0)
            iy = torch.clamp((ry * dep + r[10]) / Z, -2.0, h + 1.0)
            fx, fy = torch.floor(ix), torch.floor(iy)
            mnx[:h, :w] = torch.minimum(mnx[:h, :w], fx)
            mxx[:h, :w] = torch.maximum(mxx[:h, :w], fx + 1)
            mny[:h, :w] = torch.minimum(mny[:h, :w], fy)
            mxy[:h, :w] = torch.maximum(mxy[:h, :w], fy + 1)

        def tile(t, red):
            t = t.reshape(hp // th, th, wp // tw, tw)
            return red(red(t, 3).values, 1).values.reshape(-1)
        x0 = torch.clamp(tile(mnx, torch.min), 0, w - 1)
        x1 = torch.clamp(tile(mxx, torch.max), 0, w - 1)
        y0 = torch.clamp(tile(mny, torch.min), 0, h - 1)
        y1 = torch.clamp(tile(mxy, torch.max), 0, h - 1)
        out_w.append((x1 - x0 + 1).clamp(min=1))
        out_h.append((y1 - y0 + 1).clamp(min=1))
    return torch.stack(out_w), torch.stack(out_h)


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
    H, W, N, nd = WL[wl]
    net, _ = build_model(nd, seed=1, device=DEV)
    with torch.no_grad():
        imgs, pm, dv = synth.synth_sample(H, W, N, seed=0)
        imgs = imgs.to(DEV)
        feats = [net.feature(imgs[:, v]) for v in range(N)]
        ctx = net.cnet_depth(imgs[:, 0])
        pm = {k: v.to(DEV) for k, v in pm.items()}
        dv = dv.to(DEV)
        out = net.forward_hot(feats, ctx, pm, dv, want_intermediates=True)
    depths = out["depth"]
    weights = out["intermediates"]["view_weights"][0]
    hyp, misc = ops.stage1_hypotheses(dv[0].contiguous(), 48)
    keys = ["stage1", "stage2", "stage3"]
    rts = ops.compose_rel_proj_stages([pm[k][0].contiguous() for k in keys])
    for s, idx in ((1, 4), (2, 8)):
        cur = depths[idx][0].contiguous()
        D = net.depth_stage_nums[s]
        C = feats[0][keys[s]].shape[1]
        h, w = cur.shape
        lpp = C // 4
        tw, th = 16, 256 // lpp // 16
        itv = float(misc[s])
        print(f"--- {wl} stage {s + 1}: {h}x{w} C={C} D={D} interval={itv:.3e}  depth min/mean/max = "
              f"{float(cur.min()):.1f}/{float(cur.mean()):.1f}/{float(cur.max()):.1f}  tile {tw}x{th}")
        g = (cur[:, 1:] - cur[:, :-1]).abs()
        print(f"    |d(depth)/dx| mean {float(g.mean()):.3f} p99 {float(torch.quantile(g.flatten()[::7], 0.99)):.3f} max {float(g.max()):.3f}")
        for (tw_, th_) in (() if os.environ.get("PROBE_SKIP_BOXES") else ((tw, th), (16, 8), (16, 16), (32, 8))):
            ww, wh = boxes(cur, rts[s], itv, D, tw_, th_)
            px = (ww * wh).flatten().float()
            q = torch.quantile(px, torch.tensor([0.5, 0.9, 0.99, 0.999], device=px.device))
            line = f"    tile {tw_}x{th_}: window px median {q[0]:.0f} p90 {q[1]:.0f} p99 {q[2]:.0f} p99.9 {q[3]:.0f} max {px.max():.0f};" \
                   f" ww med {ww.float().median():.0f} max {ww.max():.0f}; wh med {wh.float().median():.0f} max {wh.max():.0f}; fit:"
            for b in (192, 256, 320, 384, 512, 768):
                line += f" {b}:{float((px <= b).float().mean()):.4f}"
            print(line)
        nhwc = ops.to_nhwc([f[keys[s]][0].contiguous() for f in feats])
        base = None
        for form in os.environ.get("DYN_WIN", ",-1,0").split(","):
            ops.set_option("dyn_win", None if form == "" else int(form))
            for _ in range(3):
                sim, smp = ops.warpcorr_dyn(nhwc[0], nhwc[1:], rts[s], cur, misc[s:s + 1], weights, D)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n = 30
            e0.record()
            for _ in range(n):
                sim, smp = ops.warpcorr_dyn(nhwc[0], nhwc[1:], rts[s], cur, misc[s:s + 1], weights, D)
            e1.record()
            torch.cuda.synchronize()
            msg = ""
            if base is None:
                base = sim.clone()
            else:
                msg = f"  max|diff vs first| {float((sim - base).abs().max()):.3e} bitwise {bool(torch.equal(sim, base))}"
            print(f"    dyn_win={form or 'default':8s} {e0.elapsed_time(e1) / n * 1e3:7.1f} us{msg}")
        ops.set_option("dyn_win", None)


if __name__ == "__main__":
    main()
