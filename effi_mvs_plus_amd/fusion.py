"""Depth-map filtering + fusion on the HIP path: the functions of the reference's
``misc/fusion.py`` that the Tanks-and-Temples driver calls (``test_tank.py:455-571``) under their own names and signatures --
``get_pixel_grids``, ``bin_op_reduce``, ``idx_img2cam``, ``idx_cam2world``, ``idx_world2cam``, ``idx_cam2img``,
``get_reproj_dynamic``, ``vis_filter_dynamic`` -- so that ``import effi_mvs_plus_amd.fusion as fusion`` lets the driver's lines
486-509 run unchanged; each is one kernel of csrc/fusion.hip behind the C ABI.  ``dynamic_filter`` is the same block of the driver
as ONE fused kernel per reference view (no ``[n,v,3,h,w]`` intermediates), what bench.py times.  CUDA (ROCm) fp32 tensors only; no
CPU fallback.
"""
from __future__ import annotations

from typing import List

import torch

from . import ops


def get_pixel_grids(height, width):
    """misc/fusion.py:8-13 -> [h,w,3,1] homogeneous pixel centres (x + 0.5, y + 0.5, 1) on the current CUDA device (the reference
    calls ``.cuda()``).  Index arithmetic only: plain tensor construction."""
    dev = torch.device("cuda", torch.cuda.current_device())
    grid = torch.ones(height, width, 3, 1, dtype=torch.float32, device=dev)
    grid[:, :, 0, 0] = torch.arange(width, dtype=torch.float32, device=dev).add_(0.5).view(1, width)
    grid[:, :, 1, 0] = torch.arange(height, dtype=torch.float32, device=dev).add_(0.5).view(height, 1)
    return grid


def bin_op_reduce(lst: List, func):
    """misc/fusion.py:16-20."""
    result = lst[0]
    for i in range(1, len(lst)):
        result = func(result, lst[i])
    return result


@ops.on_tensor_device
def idx_img2cam(idx_img_homo, depth, cam):  # nhw31, n1hw -> nhw41
    """misc/fusion.py:23-28: K^-1 . pixel, normalised by its z (+1e-9), times depth; homogeneous 1 appended."""
    return ops.fusion_points(ops.FUSION_IMG2CAM, idx_img_homo.contiguous(), cam.contiguous(), depth.contiguous())


@ops.on_tensor_device
def idx_cam2world(idx_cam_homo, cam):  # nhw41 -> nhw41
    """misc/fusion.py:31-34: E^-1 . point, normalised by its w (+1e-9)."""
    return ops.fusion_points(ops.FUSION_CAM2WORLD, idx_cam_homo.contiguous(), cam.contiguous())


@ops.on_tensor_device
def idx_world2cam(idx_world_homo, cam):  # nhw41 -> nhw41
    """misc/fusion.py:37-40."""
    return ops.fusion_points(ops.FUSION_WORLD2CAM, idx_world_homo.contiguous(), cam.contiguous())


@ops.on_tensor_device
def idx_cam2img(idx_cam_homo, cam):  # nhw41 -> nhw31
    """misc/fusion.py:43-47."""
    return ops.fusion_points(ops.FUSION_CAM2IMG, idx_cam_homo.contiguous(), cam.contiguous())


@ops.on_tensor_device
def get_reproj_dynamic(ref_depth, srcs_depth, ref_cam, srcs_cam):
    """misc/fusion.py:117-156.  ref_depth [n,1,h,w]; srcs_depth [n,v,1,h,w]; ref_cam [n,2,4,4]; srcs_cam [n,v,2,4,4]
    -> (reproj_xyd [n,v,3,h,w], ref_idx_cam [n*v,h,w,4,1], src2ref_idx_cam [n*v,h,w,4,1]) like the reference.

    ``reproj_xyd`` (x, y in reference pixels, depth in the reference camera) comes from the fused kernel.  The two point tensors
    are by-products the reference's caller only forwards to ``vis_filter_dynamic``, which ignores them; ``ref_idx_cam`` is
    ``idx_img2cam`` of the pixel grid, ``src2ref_idx_cam`` is rebuilt from the reprojected pixel and depth (K_ref^-1 . [x, y, 1] . d),
    which equals the reference's chain to fp32 rounding (~1e-6 relative), not bitwise."""
    n, v, _, h, w = srcs_depth.shape
    outs = []
    for b in range(n):
        r = ops.fusion_dynamic_filter(ref_depth[b, 0].contiguous(), srcs_depth[b, :, 0].contiguous(), ref_cam[b].contiguous(),
                                      srcs_cam[b].contiguous(), dh_view_num=1, want_points=False, want_reproj=True)
        outs.append(r["reproj_xyd"])
    reproj = torch.stack(outs)
    ref_cam_r = ref_cam[:, None].expand(n, v, 2, 4, 4).reshape(n * v, 2, 4, 4)          # every (sample, source view) pair sees its reference camera
    ref_depth_f = ref_depth[:, None].expand(n, v, 1, h, w).reshape(n * v, 1, h, w)
    ref_idx_cam = idx_img2cam(get_pixel_grids(h, w)[None], ref_depth_f, ref_cam_r)
    rp = reproj.view(n * v, 3, h, w)
    back_pix = torch.stack([rp[:, 0], rp[:, 1], torch.ones_like(rp[:, 0])], dim=-1).unsqueeze(-1)
    src2ref_idx_cam = idx_img2cam(back_pix, rp[:, 2:3].contiguous(), ref_cam_r)
    return reproj, ref_idx_cam, src2ref_idx_cam


@ops.on_tensor_device
def vis_filter_dynamic(ref_depth, reproj_xyd, ref_idx_world, src2ref_idx_cam, dist_base=4, rel_diff_base=1300, thres_view=2,
                       relative=False):
    """misc/fusion.py:157-181 -> (masks [n,v,v+1-thres_view,h,w] bool, mask = the loosest threshold's plane [n,v,1,h,w]).
    ``ref_idx_world`` and ``src2ref_idx_cam`` are accepted and, as in the reference (:161-165: only reshaped), do not enter the result."""
    masks = ops.fusion_vis_filter(ref_depth.contiguous(), reproj_xyd.contiguous(), dist_base, rel_diff_base, thres_view, relative).bool()
    return masks, masks[:, :, -1:, :, :]


@ops.on_tensor_device
def dynamic_filter(ref_depth, src_depths, ref_cam, src_cams, ref_conf, prob_threshold, dh_view_num, dist_filter, depth_filter,
                   relative=False):
    """The tensor part of ``dynamic_filter_depth`` (test_tank.py:466-512) for a batch of reference views as ONE kernel each:
    -> dict(depth [n,1,h,w] averaged depth, geo_mask / prob_mask / mask [n,1,h,w] bool, points [n,3,h,w])."""
    n = ref_depth.shape[0]
    res = [ops.fusion_dynamic_filter(ref_depth[b, 0].contiguous(), src_depths[b, :, 0].contiguous(), ref_cam[b].contiguous(),
                                     src_cams[b].contiguous(), None if ref_conf is None else ref_conf[b].contiguous(),
                                     prob_threshold, dh_view_num, dist_filter, depth_filter, relative) for b in range(n)]
    out = {"depth": torch.stack([r["depth"] for r in res]).unsqueeze(1), "points": torch.stack([r["points"] for r in res])}
    for k in ("geo_mask", "prob_mask", "mask"):
        out[k] = torch.stack([r[k] for r in res]).unsqueeze(1).bool()
    return out
