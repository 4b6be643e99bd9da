"""Depth-map filtering + fusion on the HIP path: the names of the reference's
``misc/fusion.py`` that the Tanks-and-Temples driver uses (``test_tank.py:455-571``), batched tensors in, one fused kernel
per reference view underneath (``ops.fusion_dynamic_filter``).  CUDA (ROCm) fp32 tensors only; no CPU fallback.
"""
from __future__ import annotations

import torch

from . import ops


def _backproject(kinv, x, y, z):
    """K^-1 . [x, y, 1] * z per pixel as element-wise 3x3 algebra (no batched GEMM): kinv [3,3]; x, y, z [..., h, w] -> [..., h, w, 4, 1]
    homogeneous camera points (the layout of the reference's idx_img2cam, misc/fusion.py:23-28)."""
    k = kinv
    px = k[0, 0] * x + k[0, 1] * y + k[0, 2]
    py = k[1, 0] * x + k[1, 1] * y + k[1, 2]
    pz = k[2, 0] * x + k[2, 1] * y + k[2, 2]
    s_ = z / (pz + 1e-9)
    return torch.stack([px * s_, py * s_, pz * s_, torch.ones_like(z)], dim=-1).unsqueeze(-1)


@ops.on_tensor_device
def get_reproj_dynamic(ref_depth, srcs_depth, ref_cam, srcs_cam):
    """misc/fusion.py:117-156.  ref_depth [n,1,h,w]; srcs_depth [n,v,1,h,w]; ref_cam [n,2,4,4]; srcs_cam [n,v,2,4,4]
    -> (reproj_xyd [n,v,3,h,w], ref_idx_cam [n*v,h,w,4,1], src2ref_idx_cam [n*v,h,w,4,1]) like the reference.

    ``reproj_xyd`` (x, y in reference pixels, depth in the reference camera) comes from the fused kernel.  The two point tensors
    are by-products the reference's caller only forwards to ``vis_filter_dynamic``, which ignores them; they are rebuilt here
    from the kernel's outputs with element-wise 3x3 algebra (K_ref^-1 applied to the pixel grid / to the reprojected pixel), so
    they equal the reference's to fp32 rounding (~1e-6 relative), not bitwise."""
    n, v, _, h, w = srcs_depth.shape
    outs = []
    for b in range(n):
        r = ops.fusion_dynamic_filter(ref_depth[b, 0].contiguous(), srcs_depth[b, :, 0].contiguous(), ref_cam[b].contiguous(),
                                      srcs_cam[b].contiguous(), dh_view_num=1, want_points=False, want_reproj=True)
        outs.append(r["reproj_xyd"])
    reproj = torch.stack(outs)
    dev = ref_depth.device
    xs = (torch.arange(w, dtype=torch.float32, device=dev) + 0.5).view(1, w).expand(h, w)
    ys = (torch.arange(h, dtype=torch.float32, device=dev) + 0.5).view(h, 1).expand(h, w)
    ref_pts, back_pts = [], []
    for b in range(n):
        kinv = torch.linalg.inv(ref_cam[b, 1, :3, :3].double()).float()
        ref_pts.append(_backproject(kinv, xs, ys, ref_depth[b, 0]).unsqueeze(0).expand(v, h, w, 4, 1))
        back_pts.append(_backproject(kinv, reproj[b, :, 0], reproj[b, :, 1], reproj[b, :, 2]))
    return reproj, torch.cat(ref_pts).contiguous(), torch.cat(back_pts)


@ops.on_tensor_device
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
