"""Depth-map filtering + fusion on the HIP path: the names of the reference's
``misc/fusion.py`` that the Tanks-and-Temples driver uses (``test_tank.py:455-571``), batched tensors in, one fused kernel
per reference view underneath (``ops.fusion_dynamic_filter``).  CUDA (ROCm) fp32 tensors only; no CPU fallback.
"""
from __future__ import annotations

import torch

from . import ops


def get_reproj_dynamic(ref_depth, srcs_depth, ref_cam, srcs_cam):
    """misc/fusion.py:117-156.  ref_depth [n,1,h,w]; srcs_depth [n,v,1,h,w]; ref_cam [n,2,4,4]; srcs_cam [n,v,2,4,4]
    -> reproj_xyd [n,v,3,h,w] (x, y in reference pixels, depth in the reference camera).  The reference also returns two
    intermediate point tensors that its caller only forwards to ``vis_filter_dynamic``, which ignores them; they are not
    materialised here (None, None)."""
    n, v, _, h, w = srcs_depth.shape
    outs = []
    for b in range(n):
        r = ops.fusion_dynamic_filter(ref_depth[b, 0].contiguous(), srcs_depth[b, :, 0].contiguous(), ref_cam[b].contiguous(),
                                      srcs_cam[b].contiguous(), dh_view_num=1, want_points=False, want_reproj=True)
        outs.append(r["reproj_xyd"])
    return torch.stack(outs), None, None


def dynamic_filter(ref_depth, src_depths, ref_cam, src_cams, ref_conf, prob_threshold, dh_view_num, dist_filter, depth_filter,
                   relative=False):
    """The tensor part of ``dynamic_filter_depth`` (test_tank.py:466-512) for a batch of reference views:
    -> dict(depth [n,1,h,w] averaged depth, geo_mask / prob_mask / mask [n,1,h,w] bool, points [n,3,h,w])."""
    n = ref_depth.shape[0]
    res = [ops.fusion_dynamic_filter(ref_depth[b, 0].contiguous(), src_depths[b, :, 0].contiguous(), ref_cam[b].contiguous(),
                                     src_cams[b].contiguous(), None if ref_conf is None else ref_conf[b].contiguous(),
                                     prob_threshold, dh_view_num, dist_filter, depth_filter, relative) for b in range(n)]
    out = {"depth": torch.stack([r["depth"] for r in res]).unsqueeze(1), "points": torch.stack([r["points"] for r in res])}
    for k in ("geo_mask", "prob_mask", "mask"):
        out[k] = torch.stack([r[k] for r in res]).unsqueeze(1).bool()
    return out
