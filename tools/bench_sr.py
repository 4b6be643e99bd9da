#!/usr/bin/env python3
"""Per-layer timing of one GRU iteration of each stage at a workload's map sizes: the split-precision kernels on fp32 maps
(round-2 form) against the same kernels on split-resident maps.  HIP events around N back-to-back launches of ONE layer
(so the working set of a layer is warm in the MALL -- the in-graph figure is higher for the large maps; compare forms, not
absolutes).  Usage: python tools/bench_sr.py [--workload cfg3] [--n 50] [--stages 0,1,2]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="cfg3")
    ap.add_argument("--n", type=int, default=50)
    ap.add_argument("--stages", default="0,1,2")
    ap.add_argument("--rows", default="", help="substring filter of the layer names to print")
    ap.add_argument("--rotate", type=int, default=1, help="rotate over this many buffer sets (>1: working set leaves the caches)")
    args = ap.parse_args()
    from common import build_model
    from effi_mvs_plus_amd import ops, packing
    from effi_mvs_plus_amd.models.update import _pack
    sizes = {"cfg2": (576, 800), "cfg3": (1184, 1600), "cfg4": (1056, 1920),
             # three / four cfg3 views side by side: what a BATCHED launch of the views in flight would give each layer
             "cfg3x3": (1184, 4800), "cfg3x4": (1184, 6400)}[args.workload]
    dev = "cuda:0"
    net, _ = build_model("48,8,8", seed=1, device=dev)
    ops.set_precision("split")
    g = torch.Generator().manual_seed(0)

    def timed(fn):
        for _ in range(3):
            fn(0)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(args.n):
            fn(i)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / args.n * 1e3

    for s in [int(x) for x in args.stages.split(",")]:
        h, w = sizes[0] // (8 >> s), sizes[1] // (8 >> s)
        blk = net.update_block[s]
        hd, cd = net.hdim_stage[s], net.cdim_stage[s]
        e = blk.encoder
        R = args.rotate
        rnd = lambda c: [torch.randn(c, h, w, generator=g).to(dev) for _ in range(R)]
        cor1, dfm1, hcur, x, rh, z, ctx = rnd(hd), rnd(hd), rnd(hd), rnd(hd), rnd(hd), rnd(hd), rnd(cd)
        sets = [ops.sr_alloc(6, hd, h, w, dev) for _ in range(R)]
        for k in range(R):
            for m, t_ in zip(sets[k], (cor1[k], dfm1[k], hcur[k], x[k], rh[k], z[k])):
                ops.sr_from_planar(t_, out=m)
        outs = [[torch.empty(hd, h, w, device=dev) for _ in range(4)] for _ in range(R)]
        wd2, bd2 = _pack(e._caches["d2"], e.convd2)
        wc2, bc2 = _pack(e._caches["c2"], e.convc2)
        wd, bd = _pack(e._caches["d"], e.convd)
        cmix = e.convd.out_channels
        wca, bca = packing.pack_conv1x1_after(e.convc.weight, e.convc.bias, cmix, cd)
        wzr, bzr = blk.depth_gru._packed_zr()
        wq, bq = _pack(blk.depth_gru._cq, blk.depth_gru.convq)
        dh = blk.depth_head
        wh1, bh1 = _pack(dh._c1, dh.conv1)
        wh2, bh2 = packing.pack_head_taps(dh.conv2.weight, hd)
        part = torch.empty(16, h, w, device=dev)
        wm, bm = _pack(blk._m0, blk.mask[0])
        c1 = blk.mask[0].out_channels
        w2m, b2m = packing.pack_mask_taps_per_lane(blk.mask[2].weight, blk.mask[2].bias, c1, scale=0.25)
        inv = torch.rand(1, h, w, generator=g).to(dev)
        dr = torch.linspace(1 / 935.0, 1 / 425.0, 384).to(dev)
        rows = []
        k_ = lambda i: i % R
        # the generated-input pair (lookups + 1x1 | 7x7, then convc2 | convd2) against encoder_inputs + pair on fp32 maps
        wc1, bc1 = e.convc1_raw()
        w7, b7 = e.conv7_packed()
        D = 8
        cur = [torch.randn(D, h, w, generator=g).to(dev) for _ in range(R)]
        reg = [torch.randn(D, h, w, generator=g).to(dev) for _ in range(R)]
        itv = torch.full((1,), (1 / 425.0 - 1 / 935.0) / 384 * (4 >> s), device=dev)
        dmin, dmax = (1.0 / dr[-1:]).contiguous(), (1.0 / dr[:1]).contiguous()
        rows.append(("encgen pair (lookup|7x7 + 3x3)",
                     timed(lambda i: (ops.encoder_inputs(inv, dr, itv, cur[k_(i)], reg[k_(i)], dmin, dmax, 3, h, w, wc1, bc1, w7, b7, hd, outs[k_(i)][2], outs[k_(i)][3]),
                                      ops.conv2d_k3_bf16x3_pair([outs[k_(i)][2]], wc2.wx, bc2, [outs[k_(i)][3]], wd2.wx, bd2, hd, act=1, out_a=outs[k_(i)][0], out_b=outs[k_(i)][1]))),
                     timed(lambda i: ops.encoder_pair_gen_sr(inv, dr, itv, cur[k_(i)], reg[k_(i)], dmin, dmax, 3, h, w, wc1, bc1, w7, b7, hd, wc2.wx, bc2, sets[k_(i)][4],
                                                             wd2.wx, bd2, sets[k_(i)][5], hd))))
        rows.append(("convc2|convd2 (pair)",
                     timed(lambda i: ops.conv2d_k3_bf16x3_pair([cor1[k_(i)]], wc2.wx, bc2, [dfm1[k_(i)]], wd2.wx, bd2, hd, act=1, out_a=outs[k_(i)][0], out_b=outs[k_(i)][1])),
                     timed(lambda i: ops.conv2d_k3_pair_sr([sets[k_(i)][0]], wc2.wx, bc2, sets[k_(i)][4], [sets[k_(i)][1]], wd2.wx, bd2, sets[k_(i)][5], hd, act=1))))
        rows.append(("convd+convc (k3k1)",
                     timed(lambda i: ops.conv2d_k3_k1_x3([cor1[k_(i)], dfm1[k_(i)]], wd.wx, bd, cmix, ctx[k_(i)], wca, bca, hd, relu=True, out=outs[k_(i)][0])),
                     timed(lambda i: ops.conv2d_k3_k1_sr([sets[k_(i)][0], sets[k_(i)][1]], wd.wx, bd, cmix, ctx[k_(i)], wca, bca, hd, relu=True, out_sr=sets[k_(i)][4]))))
        rows.append(("convz|convr",
                     timed(lambda i: ops.conv2d_k3_bf16x3([hcur[k_(i)], x[k_(i)]], wzr.wx, bzr, 2 * hd, epilogue=ops.EPI_GRU_ZR, aux0=hcur[k_(i)], out0=outs[k_(i)][0], out1=outs[k_(i)][1])),
                     timed(lambda i: ops.conv2d_k3_sr([sets[k_(i)][2], sets[k_(i)][3]], wzr.wx, bzr, 2 * hd, epilogue=ops.EPI_GRU_ZR, aux0=hcur[k_(i)], out0=outs[k_(i)][0], out_sr=sets[k_(i)][4]))))
        rows.append(("convq + update",
                     timed(lambda i: ops.conv2d_k3_bf16x3([rh[k_(i)], x[k_(i)]], wq.wx, bq, hd, epilogue=ops.EPI_GRU_Q, aux0=hcur[k_(i)], aux1=z[k_(i)], out0=outs[k_(i)][0])),
                     timed(lambda i: ops.conv2d_k3_sr([sets[k_(i)][4], sets[k_(i)][3]], wq.wx, bq, hd, epilogue=ops.EPI_GRU_Q, aux0=hcur[k_(i)], aux1=z[k_(i)], out0=outs[k_(i)][0], out_sr=sets[k_(i)][5]))))
        # the shipped form: fp32 state / gate in the Q4 layout (first column: planar fp32 aux maps, second: Q4)
        hq = [ops.q4_from_planar(t_) for t_ in hcur]
        zq = [ops.q4_from_planar(t_) for t_ in z]
        oq = [torch.empty_like(t_) for t_ in hq]
        rows.append(("convz|convr  (SR: planar aux / Q4)",
                     timed(lambda i: ops.conv2d_k3_sr([sets[k_(i)][2], sets[k_(i)][3]], wzr.wx, bzr, 2 * hd, epilogue=ops.EPI_GRU_ZR, aux0=hcur[k_(i)], out0=outs[k_(i)][0], out_sr=sets[k_(i)][4])),
                     timed(lambda i: ops.conv2d_k3_sr([sets[k_(i)][2], sets[k_(i)][3]], wzr.wx, bzr, 2 * hd, epilogue=ops.EPI_GRU_ZR, aux0=hq[k_(i)], out0=oq[k_(i)], out_sr=sets[k_(i)][4], q4=True))))
        rows.append(("convq + update (SR: planar aux / Q4)",
                     timed(lambda i: ops.conv2d_k3_sr([sets[k_(i)][4], sets[k_(i)][3]], wq.wx, bq, hd, epilogue=ops.EPI_GRU_Q, aux0=hcur[k_(i)], aux1=z[k_(i)], out0=outs[k_(i)][0], out_sr=sets[k_(i)][5])),
                     timed(lambda i: ops.conv2d_k3_sr([sets[k_(i)][4], sets[k_(i)][3]], wq.wx, bq, hd, epilogue=ops.EPI_GRU_Q, aux0=hq[k_(i)], aux1=zq[k_(i)], out0=oq[k_(i)], out_sr=sets[k_(i)][5], q4=True))))
        rows.append(("head conv1 + taps",
                     timed(lambda i: ops.conv2d_k3_k1_x3([hcur[k_(i)]], wh1.wx, bh1, hd, None, wh2, bh2, 9, relu=False, relu1=True, out=part[:9])),
                     timed(lambda i: ops.conv2d_k3_k1_sr([sets[k_(i)][2]], wh1.wx, bh1, hd, None, wh2, bh2, 9, relu=False, relu1=True, out=part[:9]))))
        rows.append(("mask head + upsample",
                     timed(lambda i: ops.conv2d_k3_k1_up2x([hcur[k_(i)]], wm.wx, bm, c1, w2m, b2m, inv, dr)),
                     timed(lambda i: ops.conv2d_k3_k1_up2x_sr([sets[k_(i)][2]], wm.wx, bm, c1, w2m, b2m, inv, dr))))
        print(f"stage {s + 1}: {h}x{w}, hd {hd}   (us per launch: fp32 maps / split-resident maps)")
        for name, a, b in rows:
            if args.rows and args.rows not in name:
                continue
            print(f"  {name:38s} {a:8.1f} {b:8.1f}   {b / a:5.2f}x")


if __name__ == "__main__":
    main()
