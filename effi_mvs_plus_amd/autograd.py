"""Scope row n2 (started): differentiable forms of the path's operators, forward AND backward on the HIP kernels.

``warp_correlate(ref_fea, src_feas, pairs, depth_values)`` is the stage-1 warp + correlation of ``DepthNet.forward``
(reference: models/Effi_MVS_plus.py:34-40 with ``homo_warping_new``, models/module.py:303-344): per source view the similarity
volume ``mean_c(ref * warp(src))``.  Gradients flow to the reference and source features; the sampling grid is constant, as in the
reference (``torch.no_grad()`` around the grid, module.py:313).  The rest of the training path (3-D / 2-D convolution backward)
is not built: the modules of ``effi_mvs_plus_amd.models`` still raise in training mode.
"""
import torch

from . import ops


class _WarpCorrelate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ref_fea, pairs, depth_values, *src_feas):
        # ref_fea / src_feas planar [C,h,w]; pairs [N,2,4,4]; depth_values [D] or [D,h,w]
        D = depth_values.shape[0]
        nhwc = ops.to_nhwc([ref_fea.contiguous()] + [s.contiguous() for s in src_feas])
        rt = ops.compose_rel_proj(pairs.contiguous())
        sim, _ = ops.warpcorr_views(nhwc[0], nhwc[1:], rt, depth_values, D)
        ctx.save_for_backward(*nhwc, rt, depth_values)
        ctx.D = D
        return sim

    @staticmethod
    def backward(ctx, grad_sim):
        *nhwc, rt, depth_values = ctx.saved_tensors
        g_ref, g_src = ops.warpcorr_views_bwd(nhwc[0], nhwc[1:], rt, depth_values, ctx.D, grad_sim.contiguous())
        planar = lambda t: t.permute(2, 0, 1).contiguous()  # noqa: E731
        return (planar(g_ref), None, None) + tuple(planar(g) for g in g_src)


class _HomoWarp(torch.autograd.Function):
    @staticmethod
    def forward(ctx, src_fea, src_proj, ref_proj, depth_values):
        # src_fea planar [C,h,w]; src_proj / ref_proj [4,4] (already K.[R|t]); depth_values [D] or [D,h,w]
        C_, h, w = src_fea.shape
        D = depth_values.shape[0]
        nhwc = ops.to_nhwc([src_fea.contiguous()])[0]
        rt = ops.rel_proj(src_proj.contiguous(), ref_proj.contiguous())
        ctx.save_for_backward(rt, depth_values)
        ctx.dims = (D, h, w)
        return ops.homo_warp(nhwc, rt, depth_values, D)

    @staticmethod
    def backward(ctx, grad_out):
        rt, depth_values = ctx.saved_tensors
        D, h, w = ctx.dims
        g = ops.homo_warp_bwd(rt, depth_values, D, grad_out.contiguous(), h, w)
        return g.permute(2, 0, 1).contiguous(), None, None, None


def homo_warp(src_fea, src_proj, ref_proj, depth_values):
    """Differentiable ``homo_warping_new`` for one sample (models/module.py:303-344): src_fea [C,h,w] -> warped [C,D,h,w]; the
    gradient flows to ``src_fea`` (the grid is constant, module.py:313)."""
    return _HomoWarp.apply(src_fea, src_proj, ref_proj, depth_values)


def warp_correlate(ref_fea, src_feas, pairs, depth_values):
    """ref_fea [C,h,w], src_feas list of [C,h,w] (C in 8/16/32), pairs [N,2,4,4] (view 0 = reference), depth_values [D] or
    [D,h,w] -> similarity [S,D,h,w]; differentiable w.r.t. the feature maps."""
    return _WarpCorrelate.apply(ref_fea, pairs, depth_values, *src_feas)
