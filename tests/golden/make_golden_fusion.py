#!/usr/bin/env python3
"""Golden vectors for scope row n3 (dynamic geometric-consistency filter + depth averaging of the T&T driver), produced by
RUNNING the reference's own misc/fusion.py functions (get_reproj_dynamic, vis_filter_dynamic, idx_img2cam, idx_cam2world)
on synthetic depth maps; the few tensor lines of test_tank.py:488-505 that combine their outputs (masked sum, averaged depth,
dynamic view-count rule, photometric mask) are data-generation arithmetic typed here with their line numbers.

The reference hard-codes ``.cuda()`` in get_pixel_grids (misc/fusion.py:9-10); this CPU-only container runs it with
``Tensor.cuda`` as the identity.  Run from the repo root:  python tests/golden/make_golden_fusion.py
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")
torch.Tensor.cuda = lambda self, *a, **k: self
import misc.fusion as fusion  # noqa: E402  (the reference)

from effi_mvs_plus_amd import synth  # noqa: E402

CASES = {"a": dict(H=96, W=128, N=6, seed=3, prob=0.3, dh=2, dist=4.0, dfilt=1.3, relative=False),
         "b": dict(H=70, W=90, N=11, seed=5, prob=0.55, dh=3, dist=6.0, dfilt=400.0, relative=True)}


@torch.no_grad()
def main():
    out = {}
    for tag, c in CASES.items():
        d, cams = synth.synth_depth_maps(c["H"], c["W"], c["N"], seed=c["seed"])
        g = torch.Generator().manual_seed(c["seed"] + 100)
        conf = torch.rand(1, 2 * c["H"], 2 * c["W"], generator=g)
        ref_depth, src = d[0][None, None], d[1:][None, :, None]
        ref_cam, src_cams = cams[0][None], cams[1:][None]
        v = src.shape[1]
        reproj_xyd, ref_idx_cam, src2ref_idx_cam = fusion.get_reproj_dynamic(ref_depth, src, ref_cam, src_cams)
        vis_masks, vis_mask = fusion.vis_filter_dynamic(ref_depth, reproj_xyd, ref_idx_cam, src2ref_idx_cam, dist_base=c["dist"],
                                                        rel_diff_base=c["dfilt"], thres_view=c["dh"], relative=c["relative"])
        h, w = ref_depth.shape[-2:]
        prob_mask = (F.interpolate(conf.unsqueeze(1), size=[h, w], mode="nearest") > c["prob"]).squeeze(1)   # test_tank.py:471-473
        reproj_depth = reproj_xyd[:, :, -1].clone()                                                       # :490
        reproj_depth[~vis_mask.squeeze(2)] = 0                                                            # :495
        geo_mask_sums = vis_masks.sum(dim=1)                                                              # :496
        geo_mask_sum = vis_mask.sum(dim=1)                                                                # :497
        depth_avg = (torch.sum(reproj_depth, dim=1, keepdim=True) + ref_depth) / (geo_mask_sum + 1)       # :498-499
        dy_range = v + 1
        geo_mask = geo_mask_sum >= dy_range                                                               # :502
        for i in range(c["dh"], dy_range):                                                                # :503-504
            geo_mask = torch.logical_or(geo_mask, geo_mask_sums[:, i - c["dh"]] >= i)
        mask = fusion.bin_op_reduce([prob_mask, geo_mask], torch.min)                                     # :506
        idx_img = fusion.get_pixel_grids(*depth_avg.size()[-2:]).unsqueeze(0)                             # :507
        idx_cam = fusion.idx_img2cam(idx_img, depth_avg, ref_cam)                                         # :508
        points = fusion.idx_cam2world(idx_cam, ref_cam)[..., :3, 0].permute(0, 3, 1, 2)                   # :509
        out.update({f"{tag}_{k}": np.asarray(val) for k, val in c.items()})
        out[f"{tag}_reproj_xyd"] = reproj_xyd.numpy()
        out[f"{tag}_vis_masks"] = vis_masks.numpy().astype(np.uint8)
        out[f"{tag}_depth"] = depth_avg.numpy()
        out[f"{tag}_geo_mask"] = geo_mask.reshape(1, 1, h, w).numpy().astype(np.uint8)
        out[f"{tag}_prob_mask"] = prob_mask.reshape(1, 1, h, w).numpy().astype(np.uint8)
        out[f"{tag}_mask"] = mask.reshape(1, 1, h, w).numpy().astype(np.uint8)
        out[f"{tag}_points"] = points.numpy()
        out[f"{tag}_idx_cam"] = idx_cam.numpy()                       # idx_img2cam of the averaged depth, [1,h,w,4,1]
        # the two remaining point transforms, on the same points (misc/fusion.py:37-47)
        w2c = fusion.idx_world2cam(fusion.idx_cam2world(idx_cam, ref_cam), src_cams[:, 0])
        out[f"{tag}_world2cam_src0"] = w2c.numpy()
        out[f"{tag}_cam2img_src0"] = fusion.idx_cam2img(w2c, src_cams[:, 0]).numpy()
        print(tag, "geo", float(geo_mask.float().mean()), "mask", float(mask.float().mean()), "vis", vis_masks.float().mean(dim=(0, 1, 3, 4)))
    path = os.path.join(HERE, "g12_fusion.npz")
    np.savez_compressed(path, **out)
    print(f"g12_fusion.npz: {os.path.getsize(path) / 1024:.1f} KiB")


if __name__ == "__main__":
    main()
