#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the REFERENCE itself on CPU.

Run in the build container only (needs /root/reference; see tests/refimport.py for the import recipe):
    python -B tests/golden/make_golden.py
Every .npz holds the inputs that are not derivable from a seed, the seeds, and the reference's
outputs (fp32).  Weights are never stored: both sides rebuild them with
``effi_mvs_plus_amd.synth.randomize_state_dict(state_dict, seed)`` (per-key seeded, so independent
of module registration order); the reference's checkpoints are not copied (no licence).
"""
import functools
import json
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from refimport import reference_model  # noqa: E402
from effi_mvs_plus_amd import synth  # noqa: E402

WSEED = 7


def save(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **out)
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB  " + ", ".join(f"{k}{list(np.shape(v))}" for k, v in out.items()))


def edge_cameras(N, H, W):
    """A rig with one strongly rotated view (large out-of-bounds share) and one view whose camera sits
    in front of part of the depth range (z <= 0 for some hypotheses)."""
    pm = synth.synth_cameras(H * 8, W * 8, N)["stage1"].clone()          # [1,N,2,4,4] at 1/8 scale
    import math
    a = math.radians(35.0)
    R = torch.tensor([[math.cos(a), 0, math.sin(a)], [0, 1, 0], [-math.sin(a), 0, math.cos(a)]], dtype=torch.float32)
    pm[0, 1, 0, :3, :3] = R
    pm[0, 1, 0, :3, 3] = torch.tensor([-300.0, 5.0, 120.0])
    if N > 2:
        pm[0, 2, 0, :3, 3] = torch.tensor([10.0, -20.0, -600.0])       # camera pushed 600 mm forward: z<=0 near
    return pm


@torch.no_grad()
def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    ref, net = reference_model("8,8,8")
    sd = synth.randomize_state_dict(net.state_dict(), seed=WSEED)
    net.load_state_dict(sd, strict=True)
    M, U = ref.main, ref.update

    # ---- G1: homo_warping_new ------------------------------------------------------------
    h, w, C, D = 20, 24, 8, 6
    feats = synth.smooth_features(3, C, h, w, seed=11)
    cams = synth.synth_cameras(h * 8, w * 8, 3)["stage1"]
    proj = [None] * 3
    for v in range(3):
        p = cams[0, v, 0].clone()
        p[:3, :4] = cams[0, v, 1, :3, :3] @ cams[0, v, 0, :3, :4]
        proj[v] = p.unsqueeze(0)
    d2 = torch.linspace(425.0, 935.0, D).unsqueeze(0)
    g = torch.Generator().manual_seed(5)
    d4 = 425.0 + 510.0 * torch.rand(1, D, h, w, generator=g)
    out2 = ref.module.homo_warping_new(feats[1], proj[1], proj[0], d2)
    out4 = ref.module.homo_warping_new(feats[2], proj[2], proj[0], d4)
    ecams = edge_cameras(3, h, w)
    eproj = []
    for v in range(3):
        p = ecams[0, v, 0].clone()
        p[:3, :4] = ecams[0, v, 1, :3, :3] @ ecams[0, v, 0, :3, :4]
        eproj.append(p.unsqueeze(0))
    oute1 = ref.module.homo_warping_new(feats[1], eproj[1], eproj[0], d2)
    oute2 = ref.module.homo_warping_new(feats[2], eproj[2], eproj[0], d2)
    oob = float((oute1 == 0).all(dim=1).float().mean())
    save("g01_homo_warp.npz", src1=feats[1], src2=feats[2], proj=torch.cat(proj), eproj=torch.cat(eproj), d2=d2, d4=d4,
         out2=out2, out4=out4, oute1=oute1, oute2=oute2, edge_oob_fraction=oob)
    print("   edge case: fraction of all-zero (out-of-bounds) samples =", oob)

    # ---- G2/G3: DepthNet.forward, view-weight net -------------------------------------------
    h, w, C, D, N = 16, 20, 32, 8, 4
    feats = synth.smooth_features(N, C, h, w, seed=21)
    pm = synth.synth_cameras(h * 8, w * 8, N)["stage1"]
    dv = torch.linspace(1 / 935.0, 1 / 425.0, 384).unsqueeze(0)
    samples = 1.0 / ref.module.get_depth_range_samples(dv, D, None, "cpu", torch.float32, [1, h, w])
    out = net.depthnet(feats, pm, depth_values=samples, num_depth=D, cost_regularization=net.cost_regularization,
                       pixel_wise_net=net.PixelwiseNet, G=1)
    save("g02_depthnet.npz", feat_seed=21, N=N, C=C, h=h, w=w, D=D, proj=pm, depth_samples=samples[0, :, 0, 0],
         **{"out_" + k: v for k, v in out.items()})
    # the same on the edge rig (out-of-bounds + behind-camera taps)
    epm = edge_cameras(N, h, w)
    oute = net.depthnet(feats, epm, depth_values=samples, num_depth=D, cost_regularization=net.cost_regularization,
                        pixel_wise_net=net.PixelwiseNet, G=1)
    save("g02e_depthnet_edge.npz", feat_seed=21, proj=epm, depth_samples=samples[0, :, 0, 0],
         **{"out_" + k: v for k, v in oute.items()})
    g = torch.Generator().manual_seed(31)
    ent = 2.1 * torch.rand(3, 1, 23, 37, generator=g)
    save("g03_pixelwise.npz", entropy=ent, weight=net.PixelwiseNet(ent))

    # ---- G4/G5: 3-D regulariser and cross-scale propagation block ---------------------------
    g = torch.Generator().manual_seed(41)
    vol = 0.5 * torch.randn(1, 1, 8, 16, 20, generator=g)
    prob, pro = net.cost_regularization(vol)
    save("g04_costreg.npz", vol=vol, prob=prob, pro=pro)
    x = 0.5 * torch.randn(1, 1, 8, 24, 40, generator=g)
    prior = 0.5 * torch.randn(1, 1, 8, 12, 20, generator=g)
    c2, c1 = net.CSP_R[0](x, prior)
    save("g05_cost_up_small.npz", x=x, prior=prior, conv2=c2, conv1=c1)

    # ---- G6: GetCost_initvolume ---------------------------------------------------------------
    h, w, C, D, N = 32, 40, 16, 8, 4
    feats = synth.smooth_features(N, C, h, w, seed=61)
    pm = synth.synth_cameras(h * 4, w * 4, N)["stage2"]
    g = torch.Generator().manual_seed(62)
    cur = 425.0 + 510.0 * torch.rand(1, 1, h, w, generator=g)
    cur[0, 0, 0, :4] = torch.tensor([1e5, 2e4, 0.2, 0.05])        # hit the 1e-4 / 1e4 / 1e-5 clamps
    vw = torch.rand(1, N - 1, h, w, generator=g)
    itv = ((dv[:, -1] - dv[:, 0]) / 384 * 2).view(1, 1, 1, 1)
    sim, smp = net.GetCost_initvolume(cur, features=feats, proj_matrices=pm, depth_interval=itv, depth_max=None,
                                      depth_min=None, view_weights=vw, CostNum=D, Inverse=True, G=1)
    save("g06_initvolume.npz", feat_seed=61, N=N, C=C, h=h, w=w, proj=pm, cur_depth=cur, view_weights=vw, interval=itv,
         similarity=sim, samples=smp)

    # ---- G7: pro_bilinear_sampler (global and per-pixel range, out-of-range queries) -----------
    g = torch.Generator().manual_seed(71)
    h, w, Dp, d = 12, 14, 8, 5
    volp = torch.randn(1, Dp, h, w, generator=g)
    pro_ = volp.permute(0, 2, 3, 1).reshape(h * w, 1, 1, Dp)
    q = 300.0 + 800.0 * torch.rand(1, d, h, w, generator=g)                  # partly outside [425, 935]
    gmin, gmax = torch.tensor(425.0).view(1, 1, 1, 1), torch.tensor(935.0).view(1, 1, 1, 1)
    out_g = M.pro_bilinear_sampler(pro_, q, gmin, gmax)
    pmax = 700.0 + 300.0 * torch.rand(1, 1, h, w, generator=g)
    pmin = 400.0 + 250.0 * torch.rand(1, 1, h, w, generator=g)
    out_p = M.pro_bilinear_sampler(pro_, q, pmin, pmax)
    save("g07_lookup.npz", vol=volp, query=q, gmin=gmin, gmax=gmax, pmin=pmin, pmax=pmax, out_global=out_g, out_pixel=out_p)

    # ---- G8..G10: GetCost, update-block parts, full block ---------------------------------------
    for si, (hd, cd, h, w) in enumerate([(48, 12, 16, 20), (32, 8, 20, 28), (16, 4, 24, 36)]):
        g = torch.Generator().manual_seed(80 + si)
        blk = net.update_block[si]
        Dv = 8
        reg = torch.randn(1, Dv, h, w, generator=g)
        curv = torch.randn(1, Dv, h, w, generator=g)
        pro_l = [reg.permute(0, 2, 3, 1).reshape(h * w, 1, 1, Dv), curv.permute(0, 2, 3, 1).reshape(h * w, 1, 1, Dv)]
        disp_min, disp_max = dv[:, 0, None, None, None], dv[:, -1, None, None, None]
        dmax_, dmin_ = 1.0 / disp_min, 1.0 / disp_max
        if si == 0:
            rmax, rmin = dmax_, dmin_
        else:
            rmax = 700.0 + 235.0 * torch.rand(1, 1, h, w, generator=g)
            rmin = 425.0 + 200.0 * torch.rand(1, 1, h, w, generator=g)
        itv = (disp_max - disp_min) / 384 * [4, 2, 1][si]
        inv0 = torch.rand(1, 1, h, w, generator=g)
        net_h = torch.tanh(torch.randn(1, hd, h, w, generator=g))
        ctx = torch.relu(torch.randn(1, cd, h, w, generator=g))
        scale = functools.partial(M.disp_to_depth, min_depth=dmin_, max_depth=dmax_)
        costf = functools.partial(net.GetCost, pro=pro_l, features=[torch.zeros(1, 8, h, w)], proj_matrices=torch.zeros(1, 1, 2, 4, 4),
                                  depth_interval=itv, depth_max=disp_max, depth_min=disp_min, view_weights=None, CostNum=3,
                                  Inverse=True, G=1, depth_max_cur_volume=rmax, depth_min_cur_volume=rmin)
        depth0 = scale(inv0)[1]
        cost = costf(depth0, iter=0)
        enc = blk.encoder(inv0, cost, ctx)
        hnew = blk.depth_gru(net_h, enc)
        delta = blk.depth_head(hnew)
        mask = 0.25 * blk.mask(hnew)
        up = M.upsample_depth(inv0, mask, ratio=2)
        n_out, masks, invs = blk(net_h, costf, inv0, ctx, seq_len=3, scale_inv_depth=scale)
        save(f"g08_update_stage{si + 1}.npz", reg=reg, cur=curv, rmax=rmax, rmin=rmin, interval=itv, inv0=inv0, net=net_h, ctx=ctx,
             depth_values=dv, cost=cost, enc=enc, hnew=hnew, delta=delta, mask=mask, up=up, blk_net=n_out, blk_mask=masks[-1],
             blk_inv=torch.stack(invs))

    # ---- G11: full model, small (S=3, 8,8,8) and mid (S=4, 48,8,8), with intermediates ------------
    for tag, (H, W, N, nd) in {"small": (128, 160, 4, "8,8,8"), "mid": (256, 320, 5, "48,8,8")}.items():
        _, fnet = reference_model(nd)
        fsd = synth.randomize_state_dict(fnet.state_dict(), seed=WSEED)
        fnet.load_state_dict(fsd, strict=True)
        imgs, pmd, dvv = synth.synth_sample(H, W, N, seed=3)
        inter = {}
        hooks = [fnet.depthnet.register_forward_hook(lambda m, i, o: inter.update(view_weights=o["view_weights"], reg_volume1=o["reg_volume"], cur_volume1=o["volume"].squeeze(1)))]
        for s in (0, 1):
            hooks.append(fnet.CSP_R[s].register_forward_hook(lambda m, i, o, s=s: inter.update({f"reg_volume{s + 2}": o[0].squeeze(1)})))
            hooks.append(fnet.CSP_C[s].register_forward_hook(lambda m, i, o, s=s: inter.update({f"cur_volume{s + 2}": o[0].squeeze(1)})))
        out = fnet(imgs, pmd, dvv)
        for hk in hooks:
            hk.remove()
        arrs = {f"depth{i:02d}": d for i, d in enumerate(out["depth"])}
        arrs["photometric_confidence"] = out["photometric_confidence"]
        arrs.update({"inter_" + k: v for k, v in inter.items()})
        save(f"g11_full_{tag}.npz", H=H, W=W, N=N, ndepths=np.array([int(e) for e in nd.split(",")]), img_seed=3, weight_seed=WSEED, **arrs)

    # ---- interface fixture: the reference's state-dict keys/shapes --------------------------------
    keys = {k: [list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in net.state_dict().items()}
    with open(os.path.join(HERE, "state_dict_keys.json"), "w") as f:
        json.dump(keys, f, indent=0, sort_keys=True)
    print("state_dict_keys.json:", len(keys), "entries")


if __name__ == "__main__":
    main()
