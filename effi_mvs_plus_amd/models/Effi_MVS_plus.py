"""Drop-in mirror of the reference's ``models/Effi_MVS_plus.py`` (cascade orchestration and cost builders).

Public classes / functions keep the reference's names, signatures, return structures and state-dict
keys (``models/Effi_MVS_plus.py`` in bdwsq1996/Effi-MVS-plus); the work is done by the gfx950 kernels.
``Effi_MVS_plus.forward`` runs the feature / context pyramids (scope row n1, also on the HIP conv kernels) and then
``forward_hot`` -- the cost-volume path this repository accelerates and ``bench.py`` times.  Inference only.
"""
from __future__ import annotations

from functools import partial

import torch
import torch.nn as nn

from .. import ops, packing
from .module import (ConvBnReLU, CostRegNet_2_sample_FPN3D_Fast, P_1to8_FeatureNet_Fast, _require_eval,
                     cost_up_small, mvs_loss)  # noqa: F401  (mvs_loss is re-exported like the reference does)
from .update import BasicUpdateBlock

Align_Corners_Range = False


def _stack(ts):
    return ts[0].unsqueeze(0) if len(ts) == 1 else torch.stack(ts)


# =============================================================================================
# small tensor helpers of the reference (plain tensor algebra; the fused path does them in-kernel)
# =============================================================================================
def disp_to_depth(disp, min_depth, max_depth):
    """normalised inverse depth -> (scaled inverse depth, depth)  (reference: Effi_MVS_plus.py:138-148)."""
    lo, hi = 1 / max_depth, 1 / min_depth
    scaled = (lo + (hi - lo) * disp).clamp(min=1e-4)
    return scaled, 1 / scaled


def depth_to_disp(depth, min_depth, max_depth):
    """depth -> normalised inverse depth (reference: Effi_MVS_plus.py:151-164)."""
    lo, hi = 1 / max_depth, 1 / min_depth
    return (1 / depth - lo) / ((hi - lo) + 1e-10)


@ops.on_tensor_device
def bilinear_sampler(img, coords, mode='bilinear', mask=False):
    """Wrapper for grid_sample on pixel coordinates, as the reference defines it (Effi_MVS_plus.py:102-117): img [N,C,1,W],
    coords [N,Ho,Wo,2] = (x in pixels, y) -> [N,C,Ho,Wo] (with ``mask``: also the in-range mask [N,Ho,Wo,1] as float).  Like
    the reference it only serves the 1-D ("stereo") case: H must be 1 (AssertionError otherwise); the other half of its assert,
    ``torch.unique(ygrid).numel() == 1``, is a device sort + host sync with no effect on the values when H == 1 (every y
    un-normalises to row 0) and is not reproduced.  ``mode`` other than 'bilinear' is ignored by the reference too (:112 does
    not forward it)."""
    assert img.shape[-2] == 1   # This is a stereo problem
    return ops.bilinear_sampler1d(img.contiguous(), coords.contiguous(), want_mask=bool(mask))


@ops.on_tensor_device
def pro_bilinear_sampler(pro, depth_sample, depth_min, depth_max):
    """1-D linear lookup of per-pixel D-vectors at ``depth_sample`` (reference: Effi_MVS_plus.py:118-134).

    pro [B*h*w,1,1,D] (pixel-major, as the reference reshapes it; any strides); depth_sample [B,d,h,w];
    depth_min / depth_max broadcastable to [B,1,h,w] -> [B,d,h,w].  The torch.unique assert of
    bilinear_sampler (:109, a device sort + host sync) is not reproduced.
    """
    B, d, h, w = depth_sample.shape
    outs = []
    for b in range(B):
        vol = pro[b * h * w:(b + 1) * h * w]
        lo = depth_min[b] if depth_min.shape[0] == B else depth_min[0]
        hi = depth_max[b] if depth_max.shape[0] == B else depth_max[0]
        outs.append(ops.vol_lookup1d(vol, depth_sample[b].contiguous(), lo, hi, h, w))
    return _stack(outs)


@ops.on_tensor_device
def upsample_depth(depth, mask, ratio=8):
    """Convex-combination upsampling [N,1,H,W] x [N,9*r*r,H,W] -> [N,r*H,r*W] (reference: :167-178); r = 2."""
    if ratio != 2:
        raise NotImplementedError("upsample_depth: the HIP path instantiates ratio 2 (feat_ratio of every stage)")
    return _stack([ops.convex_upsample2x(depth[n].contiguous(), mask[n].contiguous())[0] for n in range(depth.shape[0])])


# =============================================================================================
# view-weight net container (keys PixelwiseNet.{0,1,2}.{conv,bn}.*, PixelwiseNet.3.{weight,bias})
# =============================================================================================
class PixelwiseNet2d(nn.Sequential):
    """nn.Sequential(ConvBnReLU(1,16), ConvBnReLU(16,16), ConvBnReLU(16,8), Conv2d(8,1,1), Sigmoid) of the
    reference (Effi_MVS_plus.py:361-362) whose forward is ONE fused kernel."""

    def __init__(self):
        super().__init__(ConvBnReLU(1, 16), ConvBnReLU(16, 16), ConvBnReLU(16, 8), nn.Conv2d(8, 1, 1), nn.Sigmoid())
        self._cache = packing.PackCache()

    def run(self, entropy):
        """entropy [n,h,w] -> weights [n,h,w]."""
        _require_eval(self)
        t = []
        for i in range(3):
            t += [self[i].conv.weight, self[i].bn.weight, self[i].bn.bias, self[i].bn.running_mean, self[i].bn.running_var]
        t += [self[3].weight, self[3].bias]
        return ops.pixelwise_net(entropy, self._cache.get(t, lambda: packing.pack_pixelwise_net(self)))

    @ops.on_tensor_device

    def forward(self, x):
        n, c, h, w = x.shape
        return self.run(x.reshape(n * c, h, w).contiguous()).view(n, c, h, w)


# =============================================================================================
# a2: stage-1 cost volume (reference: Effi_MVS_plus.py:9-89)
# =============================================================================================
class DepthNet(nn.Module):
    def __init__(self, cnnpixel=False):
        super().__init__()

    @staticmethod
    def run(feats, pairs, depth, num_depth, cost_regularization, pixel_wise_net):
        """Unbatched: feats list of [C,h,w]; pairs [N,2,4,4]; depth [D] or [D,h,w]."""
        nhwc = ops.to_nhwc(feats)
        rt = ops.compose_rel_proj(pairs)
        sim_views, entropy = ops.warpcorr_views(nhwc[0], nhwc[1:], rt, depth, num_depth)
        if pixel_wise_net is None:              # unweighted average of the views (reference :55-58,70); view_weights stays the empty list
            weights = None
        else:
            weights = pixel_wise_net.run(entropy) if hasattr(pixel_wise_net, "run") else \
                pixel_wise_net(entropy.unsqueeze(1)).squeeze(1).contiguous()
        volume = ops.view_aggregate(sim_views, weights)
        reg, _ = cost_regularization.run(volume.unsqueeze(0))
        d, conf = ops.softmax_regress_conf(reg[0], depth)
        return {"depth": d, "photometric_confidence": conf, **({"view_weights": weights} if weights is not None else {}),
                "reg_volume": reg[0], "volume": volume.unsqueeze(0)}

    @ops.on_tensor_device

    def forward(self, features, proj_matrices, depth_values, num_depth, cost_regularization, pixel_wise_net, G=8):
        assert len(features) == proj_matrices.shape[1], "Different number of images and projection matrices"
        assert depth_values.shape[1] == num_depth, "depth_values.shape[1]:{}  num_depth:{}".format(
            depth_values.shape[1], num_depth)
        if G != 1:
            raise NotImplementedError("DepthNet: group-wise correlation is instantiated for G=1 (the shipped model)")
        if self.training:
            if pixel_wise_net is None:
                raise NotImplementedError("DepthNet (training): the unweighted average (pixel_wise_net=None) has no backward on the "
                                          "HIP path; the shipped model always passes its PixelwiseNet")
            from .. import train_path
            return train_path.depthnet(pixel_wise_net, cost_regularization, features, proj_matrices, depth_values)
        outs = []
        for b in range(features[0].shape[0]):
            outs.append(self.run([f[b] for f in features], proj_matrices[b].contiguous(), depth_values[b], num_depth,
                                 cost_regularization, pixel_wise_net))
        res = {k: _stack([o[k] for o in outs]) for k in outs[0]}
        if pixel_wise_net is None:
            res["view_weights"] = []            # what the reference returns in this branch (:20,89)
        return res


# =============================================================================================
# a8: stage-2/3 dynamic cost volume (reference: Effi_MVS_plus.py:180-251)
# =============================================================================================
class GetCost_initvolume(nn.Module):
    def __init__(self):
        super().__init__()

    @ops.on_tensor_device

    def forward(self, depth_values, features, proj_matrices, depth_interval, depth_max, depth_min, view_weights,
                CostNum=4, Inverse=True, G=8, iter=1, inter_iter=[1, 1, 1, 1]):
        if not Inverse or G != 1 or view_weights is None:
            raise NotImplementedError("GetCost_initvolume: HIP path covers Inverse=True, G=1, weighted views")
        interval = depth_interval * inter_iter[iter] if inter_iter[iter] != 1 else depth_interval
        sims, samples = [], []
        if torch.is_grad_enabled() and (self.training or any(f.requires_grad for f in features) or view_weights.requires_grad):
            from .. import autograd as A          # differentiable form: gradients to the feature maps and the view weights
            for b in range(depth_values.shape[0]):
                s, d = A.warp_correlate_dyn(features[0][b], [f[b] for f in features[1:]], view_weights[b], proj_matrices[b],
                                            depth_values[b, 0], interval[b].reshape(1), CostNum)
                sims.append(s), samples.append(d)
            return _stack(sims), _stack(samples)
        for b in range(depth_values.shape[0]):
            nhwc = ops.to_nhwc([f[b] for f in features])
            rt = ops.compose_rel_proj(proj_matrices[b].contiguous())
            s, d = ops.warpcorr_dyn(nhwc[0], nhwc[1:], rt, depth_values[b, 0].contiguous(),
                                    interval[b].reshape(1).contiguous(), view_weights[b].contiguous(), CostNum)
            sims.append(s), samples.append(d)
        return _stack(sims), _stack(samples)


# =============================================================================================
# a11: per-iteration cost lookup (reference: Effi_MVS_plus.py:253-303)
# =============================================================================================
class GetCost(nn.Module):
    def __init__(self):
        super().__init__()

    def make_lookup(self, b, pro, depth_interval, depth_max_cur_volume=0, depth_min_cur_volume=0, CostNum=4,
                    disp_range=None, **unused):
        """Closure used by the fused update block: normalised inverse depth [1,h,w] -> cost [2*CostNum,h,w]."""
        interval = depth_interval[b].reshape(1).contiguous()
        lo = depth_min_cur_volume[b] if depth_min_cur_volume.shape[0] > 1 else depth_min_cur_volume[0]
        hi = depth_max_cur_volume[b] if depth_max_cur_volume.shape[0] > 1 else depth_max_cur_volume[0]
        cur, reg = pro[-1], pro[0]

        def lookup(inv_depth, out=None):
            h, w = inv_depth.shape[-2:]
            n = h * w
            return ops.getcost(inv_depth, disp_range[b], interval, cur[b * n:(b + 1) * n], reg[b * n:(b + 1) * n],
                               lo, hi, CostNum, h, w, input_is_depth=False, out=out)

        def lookup_conv1x1(inv_depth, weight, bias, cout, out=None):
            """Same lookup with the encoder's convc1 (+ReLU) applied in the kernel: -> [cout,h,w]."""
            h, w = inv_depth.shape[-2:]
            n = h * w
            return ops.getcost_conv1x1(inv_depth, disp_range[b], interval, cur[b * n:(b + 1) * n], reg[b * n:(b + 1) * n],
                                       lo, hi, CostNum, h, w, weight, bias, cout, relu=True, out=out)

        def lookup_encoder_inputs(inv_depth, wc1, bc1, wd1, bd1, cout, out_c1=None, out_d1=None):
            """lookup_conv1x1 and the encoder's convd1 (7x7, +ReLU) of the same map in one launch."""
            h, w = inv_depth.shape[-2:]
            n = h * w
            return ops.encoder_inputs(inv_depth, disp_range[b], interval, cur[b * n:(b + 1) * n], reg[b * n:(b + 1) * n],
                                      lo, hi, CostNum, h, w, wc1, bc1, wd1, bd1, cout, out_c1, out_d1)

        def lookup_encoder_inputs_sr(inv_depth, wc1, bc1, wd1, bd1, cout, out_c1, out_d1):
            """lookup_encoder_inputs writing split-resident maps (ops.SRMap)."""
            h, w = inv_depth.shape[-2:]
            n = h * w
            return ops.encoder_inputs_sr(inv_depth, disp_range[b], interval, cur[b * n:(b + 1) * n], reg[b * n:(b + 1) * n],
                                         lo, hi, CostNum, h, w, wc1, bc1, wd1, bd1, cout, out_c1, out_d1)

        lookup.conv1x1 = lookup_conv1x1 if CostNum in (2, 3, 4) else None
        lookup.encoder_inputs = lookup_encoder_inputs if CostNum == 3 else None
        def lookup_encoder_pair_sr(inv_depth, wc1, bc1, wd1, bd1, hd, wc2, bc2, out_c2, wd2, bd2, out_d2):
            """lookup_encoder_inputs_sr + the convc2 | convd2 pair in one launch (ops.encoder_pair_gen_sr)."""
            h, w = inv_depth.shape[-2:]
            n = h * w
            return ops.encoder_pair_gen_sr(inv_depth, disp_range[b], interval, cur[b * n:(b + 1) * n], reg[b * n:(b + 1) * n],
                                           lo, hi, CostNum, h, w, wc1, bc1, wd1, bd1, hd, wc2, bc2, out_c2, wd2, bd2, out_d2, hd)

        lookup.encoder_inputs_sr = lookup_encoder_inputs_sr if CostNum == 3 else None
        lookup.encoder_pair_sr = lookup_encoder_pair_sr if CostNum == 3 else None
        return lookup

    @ops.on_tensor_device

    def forward(self, depth_values, pro, features, proj_matrices, depth_interval, depth_max, depth_min, view_weights,
                CostNum=4, Inverse=True, G=8, depth_max_cur_volume=0, depth_min_cur_volume=0, iter=1,
                inter_iter=[1, 1, 1, 1], disp_range=None):
        if not Inverse:
            raise NotImplementedError("GetCost: HIP path covers Inverse=True")
        interval = depth_interval * inter_iter[iter] if inter_iter[iter] != 1 else depth_interval
        B, _, h, w = depth_values.shape
        n = h * w
        if torch.is_grad_enabled() and (pro[0].requires_grad or pro[-1].requires_grad):
            from .. import autograd as A          # differentiable form: gradients to the two cached volumes
            planar = lambda v: v.reshape(B, h, w, v.shape[-1]).permute(0, 3, 1, 2)      # noqa: E731  ([B*h*w,1,1,D] -> [B,D,h,w])
            return A.getcost(planar(pro[-1]), planar(pro[0]), depth_values, None, interval.reshape(B), depth_min_cur_volume,
                             depth_max_cur_volume, CostNum, input_is_depth=True)
        outs = []
        for b in range(B):
            lo = depth_min_cur_volume[b] if depth_min_cur_volume.shape[0] == B else depth_min_cur_volume[0]
            hi = depth_max_cur_volume[b] if depth_max_cur_volume.shape[0] == B else depth_max_cur_volume[0]
            outs.append(ops.getcost(depth_values[b].contiguous(), None, interval[b].reshape(1).contiguous(),
                                    pro[-1][b * n:(b + 1) * n], pro[0][b * n:(b + 1) * n], lo, hi, CostNum, h, w,
                                    input_is_depth=True))
        return _stack(outs)


# =============================================================================================
# a17: the cascade (reference: Effi_MVS_plus.py:315-568)
# =============================================================================================
class Effi_MVS_plus(nn.Module):
    def __init__(self, args, refine=False, ndepths=48, depth_interals_ratio=[4, 2, 1], share_cr=False, CostNum=4,
                 inverse=True, stage_channel=True):
        super().__init__()
        self.refine = refine
        self.share_cr = share_cr
        self.ndepths = args.ndepths
        self.inverse = inverse
        self.depth_interals_ratio = depth_interals_ratio
        self.cost_num = 2
        self.seq_len = [int(e) for e in args.GRUiters.split(",")]
        self.args = args
        self.num_stage = 3
        self.CostNum = args.CostNum
        self.CostNum_ratio = [4, 2, 1]
        self.GetCost = GetCost()
        self.GetCost_initvolume = GetCost_initvolume()
        self.stage_channel = stage_channel
        self.hdim_stage = [48, 32, 16]
        self.cdim_stage = [12, 8, 4]
        self.context_feature = [60, 40, 20]
        self.depth_stage_nums = [int(e) for e in args.ndepths.split(",")]
        self.hdim = 32
        self.cdim = 32
        self.feat_ratio = [2, 2, 2]
        self.G = 1
        self.cost_dim_stage = [32, 16, 8]
        self.feature_in_channel = [8, 16, 32, 64]
        self.context_in_channel = [4, 8, 16, 32]
        if list(depth_interals_ratio) != [4, 2, 1] or not inverse or not stage_channel:
            raise NotImplementedError("Effi_MVS_plus: HIP path covers the shipped configuration "
                                      "(depth_interals_ratio=[4,2,1], inverse=True, stage_channel=True)")

        self.PixelwiseNet = PixelwiseNet2d()
        self.feature = P_1to8_FeatureNet_Fast(base_channels=4, in_channel=self.feature_in_channel,
                                              out_channel=self.cost_dim_stage, stage_channel=self.stage_channel)
        self.feature.channels_last_outputs = True     # the cost-volume kernels read channel-last features
        self.cnet_depth = P_1to8_FeatureNet_Fast(base_channels=4, in_channel=self.context_in_channel,
                                                 out_channel=self.context_feature, stage_channel=self.stage_channel)
        blocks = [BasicUpdateBlock(hidden_dim=self.hdim_stage[s], cost_dim=self.G * self.CostNum,
                                   ratio=self.feat_ratio[s], context_dim=self.cdim_stage[s], UpMask=True,
                                   Inverse=self.inverse, cost_num=self.cost_num) for s in range(3)]
        # the reference registers every block twice (attribute + ModuleList): both key families must exist
        self.update_block_depth1, self.update_block_depth2, self.update_block_depth3 = blocks
        self.update_block = nn.ModuleList(blocks)
        self.depthnet = DepthNet()
        self.CSP_R1 = cost_up_small(in_channels=self.G, base_channels=8)
        self.CSP_R2 = cost_up_small(in_channels=self.G, base_channels=8)
        self.CSP_R = nn.ModuleList([self.CSP_R1, self.CSP_R2])
        self.CSP_C1 = cost_up_small(in_channels=self.G, base_channels=8)
        self.CSP_C2 = cost_up_small(in_channels=self.G, base_channels=8)
        self.CSP_C = nn.ModuleList([self.CSP_C1, self.CSP_C2])
        self.cost_regularization = CostRegNet_2_sample_FPN3D_Fast(in_channels=self.G, base_channels=8)

    # -----------------------------------------------------------------------------------------
    def _hot_single(self, feats, ctx, pairs, disp_range, want_intermediates=False):
        """One sample, unbatched.  feats: per view {stageK: [C,h,w]}; ctx {stageK: [hd+cd,h,w]};
        pairs {stageK: [N,2,4,4]}; disp_range [384] ascending inverse depths."""
        D1 = self.depth_stage_nums[0]
        ops.mark("begin")
        hyp, misc = ops.stage1_hypotheses(disp_range, D1)      # misc: 3 intervals, depth_min_, depth_max_
        g_min, g_max = misc[3:4], misc[4:5]
        preds, inter = [], {}
        conf = None
        weights = reg_vol = cur_vol = None
        lo_prev, hi_prev = g_min, g_max          # depth range the PREVIOUS stage's volumes are sampled on

        keys = ["stage{}".format(s + 1) for s in range(self.num_stage)]
        # per-stage inputs that depend on nothing but the features / cameras / context pyramid: relative projections of all
        # stages in one launch, tanh / relu halves of all context maps (hidden state and context input of the update blocks)
        # in one launch
        if self.num_stage <= 4:
            rts = ops.compose_rel_proj_stages([pairs[k].contiguous() for k in keys])
        else:
            rts = [ops.compose_rel_proj(pairs[k]) for k in keys]

        table = feats if isinstance(feats, ops.ViewTable) else None     # maps read through a device pointer table (scan_eval.py)

        def geometry(s):
            if table is not None:
                return (), rts[s], table.shapes[s]
            maps = [f[keys[s]] for f in feats]
            return ops.to_nhwc(maps), rts[s], maps[0].shape

        # split-resident maps of the three update blocks (ops.SRMap): one allocation per stage, all borders zeroed by ONE launch, the
        # initial hidden states written in both forms by the launch that splits the context maps
        sr_maps = None
        use_sr = (ops.uses_sr() and self.num_stage <= 4 and self.CostNum == 3
                  and all(hd in (16, 32, 48) for hd in self.hdim_stage[:self.num_stage])
                  and all(ctx[k].shape[-1] % 4 == 0 for k in keys))

        # fp32 copy of each hidden state in the Q4 layout ([hd/4][h][w][4]: one 16-byte access per lane in the GRU epilogues) where
        # the stage's block takes the split-resident path and nothing else reads that copy
        state_q4 = [False] * self.num_stage

        def all_states():
            nonlocal sr_maps, state_q4
            cs = [ctx[k].contiguous() for k in keys]
            if use_sr:
                nm = BasicUpdateBlock.N_SR_MAPS
                sr_maps = [ops.sr_alloc(nm, self.hdim_stage[s], cs[s].shape[1], cs[s].shape[2], cs[s].device, clear=False)
                           for s in range(self.num_stage)]
                ops.sr_clear_border(sr_maps)
                state_q4 = [self.update_block[s].state_q4_ok(not want_intermediates) and ops.uses_sr(cs[s].shape[1] * cs[s].shape[2])
                            for s in range(self.num_stage)]
                return dict(enumerate(ops.split_tanh_relu_stages_sr(cs, self.hdim_stage[:self.num_stage], self.cdim_stage[:self.num_stage],
                                                                    [m[-1] for m in sr_maps], q4=state_q4)))
            if self.num_stage <= 4:
                return dict(enumerate(ops.split_tanh_relu_stages(cs, self.hdim_stage[:self.num_stage], self.cdim_stage[:self.num_stage])))
            return {s: ops.split_tanh_relu(cs[s], self.hdim_stage[s], self.cdim_stage[s]) for s in range(self.num_stage)}

        geo = {0: geometry(0)}
        with ops.Branch() as prep_branch:        # what the stage-1 cost volume does not need (side stream when branches are on)
            st = all_states()
            for s in range(1, self.num_stage):
                geo[s] = geometry(s)
        prep_joined = False
        tail_branch = None
        inv_next = None                          # normalised inverse depth the next update block starts from
        for s in range(self.num_stage):
            if s > 0:
                ops.mark("stage{}".format(s))              # end of the previous stage
            nhwc, rt, (_, h, w) = geo[s]
            if s == 0:
                if table is not None:
                    sim_views, entropy = ops.warpcorr_views_tbl(table, 0, rt, hyp, D1, x3=bool(ops.option("warp_x3")))
                else:
                    sim_views, entropy = ops.warpcorr_views(nhwc[0], nhwc[1:], rt, hyp, D1, x3=bool(ops.option("warp_x3")))
                weights = self.PixelwiseNet.run(entropy)
                cur_vol = ops.view_aggregate(sim_views, weights)
                reg_vol = self.cost_regularization.run(cur_vol.unsqueeze(0))[0][0]
                depth, c, inv_next = ops.softmax_regress_conf(reg_vol, hyp, disp_range)     # + depth_to_inv of it (:538)
                with ops.Branch() as tail_branch:      # the confidence map is only an output: off the critical path
                    conf = ops.upsample_nearest(c.unsqueeze(0), 4)[0]
                preds.append(depth)
                lo_cur, hi_cur = g_min, g_max
            else:
                D = self.depth_stage_nums[s]
                if table is not None:
                    sim, samples = ops.warpcorr_dyn_tbl(table, s, rt, preds[-1], misc[s:s + 1], weights, D)
                else:
                    sim, samples = ops.warpcorr_dyn(nhwc[0], nhwc[1:], rt, preds[-1], misc[s:s + 1], weights, D)
                x = sim.unsqueeze(0)
                # the two cross-scale blocks are independent chains of 5 kernels on volumes of the same shape
                csp_r, csp_c = self.CSP_R[s - 1], self.CSP_C[s - 1]
                if cur_vol.shape == reg_vol.shape and cur_vol.dim() == 3 and cost_up_small.pairable(csp_r, csp_c, x, w // 2):
                    # ... so each layer of both is ONE launch (no second stream, no fork / join bubbles)
                    prior, prior_c = ops.vol_lookup1d_pair(reg_vol, cur_vol, samples, lo_prev, hi_prev, h // 2, w // 2)
                    (reg_new, _), (cur_new, _) = cost_up_small.run_pair(csp_r, csp_c, x, prior.unsqueeze(0), prior_c.unsqueeze(0))
                    reg_vol, cur_vol = reg_new[0], cur_new[0]
                else:
                    with ops.Branch() as br:           # general shapes: CSP_C goes to the side stream
                        prior_c = ops.vol_lookup1d(cur_vol, samples, lo_prev, hi_prev, h // 2, w // 2)
                        cur_new = csp_c.run(x, prior_c.unsqueeze(0))[0][0]
                    prior = ops.vol_lookup1d(reg_vol, samples, lo_prev, hi_prev, h // 2, w // 2)
                    reg_vol = csp_r.run(x, prior.unsqueeze(0))[0][0]
                    br.join(cur_new)
                    cur_vol = cur_new
                lo_cur, hi_cur = samples[D - 1], samples[0]      # depth_min2 / depth_max2 (:508-509)
            if want_intermediates:
                inter["view_weights"] = weights
                inter["reg_volume{}".format(s + 1)] = reg_vol
                inter["cur_volume{}".format(s + 1)] = cur_vol
            if not prep_joined:                   # first use of the preparation's results: the stage-1 update block
                prep_branch.join(*[t_ for k in st for t_ in st[k]], *[t_ for k in range(1, self.num_stage) for t_ in (list(geo[k][0]) + [geo[k][1]])])
                prep_joined = True
            hidden, inp = st[s]
            inv_cur = (inv_next if inv_next is not None else ops.depth_to_inv(preds[-1], disp_range)).unsqueeze(0)
            cur_c, reg_c, lo_c, hi_c, itv = cur_vol, reg_vol, lo_cur, hi_cur, misc[s:s + 1]

            def lookup(inv_depth, out=None, cur_c=cur_c, reg_c=reg_c, lo_c=lo_c, hi_c=hi_c, itv=itv, h=h, w=w):
                return ops.getcost(inv_depth, disp_range, itv, cur_c, reg_c, lo_c, hi_c, self.CostNum, h, w, out=out)

            def lookup_conv1x1(inv_depth, weight, bias, cout, out=None, cur_c=cur_c, reg_c=reg_c, lo_c=lo_c, hi_c=hi_c,
                               itv=itv, h=h, w=w):
                return ops.getcost_conv1x1(inv_depth, disp_range, itv, cur_c, reg_c, lo_c, hi_c, self.CostNum, h, w,
                                           weight, bias, cout, relu=True, out=out)

            def lookup_encoder_inputs(inv_depth, wc1, bc1, wd1, bd1, cout, out_c1=None, out_d1=None, cur_c=cur_c, reg_c=reg_c,
                                      lo_c=lo_c, hi_c=hi_c, itv=itv, h=h, w=w):
                return ops.encoder_inputs(inv_depth, disp_range, itv, cur_c, reg_c, lo_c, hi_c, self.CostNum, h, w,
                                          wc1, bc1, wd1, bd1, cout, out_c1, out_d1)

            def lookup_encoder_inputs_sr(inv_depth, wc1, bc1, wd1, bd1, cout, out_c1, out_d1, cur_c=cur_c, reg_c=reg_c,
                                         lo_c=lo_c, hi_c=hi_c, itv=itv, h=h, w=w):
                return ops.encoder_inputs_sr(inv_depth, disp_range, itv, cur_c, reg_c, lo_c, hi_c, self.CostNum, h, w,
                                             wc1, bc1, wd1, bd1, cout, out_c1, out_d1)

            lookup.conv1x1 = lookup_conv1x1 if self.CostNum in (2, 3, 4) else None
            lookup.encoder_inputs = lookup_encoder_inputs if self.CostNum == 3 else None
            def lookup_encoder_pair_sr(inv_depth, wc1, bc1, wd1, bd1, hd, wc2, bc2, out_c2, wd2, bd2, out_d2, cur_c=cur_c, reg_c=reg_c,
                                       lo_c=lo_c, hi_c=hi_c, itv=itv, h=h, w=w):
                return ops.encoder_pair_gen_sr(inv_depth, disp_range, itv, cur_c, reg_c, lo_c, hi_c, self.CostNum, h, w,
                                               wc1, bc1, wd1, bd1, hd, wc2, bc2, out_c2, wd2, bd2, out_d2, hd)

            lookup.encoder_inputs_sr = lookup_encoder_inputs_sr if self.CostNum == 3 else None
            lookup.encoder_pair_sr = lookup_encoder_pair_sr if self.CostNum == 3 else None
            _, masks, invs, depths = self.update_block[s].run_fused(hidden, lookup, inv_cur, inp, self.seq_len[s],
                                                                     disp_range, fuse_upsample=not want_intermediates,
                                                                     sr_maps=None if sr_maps is None else sr_maps[s],
                                                                     net_sr_ready=sr_maps is not None, net_owned=True, net_q4=state_q4[s])
            preds.extend(d[0] for d in depths)
            if isinstance(masks[-1], tuple):      # mask head + upsampling ran as one kernel
                up_depth, inv_next = masks[-1]
            else:
                _, up_depth, inv_next = ops.convex_upsample2x(invs[-1], masks[-1], disp_range, want_inv=False, want_depth_inv=True)
            preds.append(up_depth)
            lo_prev, hi_prev = lo_cur, hi_cur
        ops.mark("stage{}".format(self.num_stage))
        if tail_branch is not None:
            tail_branch.join(conf)
        out = {"depth": preds, "photometric_confidence": conf}
        if want_intermediates:
            out["intermediates"] = inter
        return out

    @ops.on_tensor_device

    def forward_hot(self, features, cnet_depth, proj_matrices, depth_values, want_intermediates=False):
        """The accelerated path: everything of ``forward`` after the FPN (reference: Effi_MVS_plus.py:437-568).

        features: list over views of {"stageK": [B,C,h,w]}; cnet_depth: {"stageK": [B,hd+cd,h,w]};
        proj_matrices: {"stageK": [B,N,2,4,4]}; depth_values [B,384].  ``features`` may also be an ``ops.ViewTable`` (B = 1,
        inference): the maps of this item's views behind a device pointer table, as the evaluation-set runner passes them.
        """
        if self.training:
            from .. import train_path
            return train_path.hot_path(self, features, cnet_depth, proj_matrices, depth_values)
        B = depth_values.shape[0]
        if isinstance(features, ops.ViewTable) and B != 1:
            raise ValueError("forward_hot: a ViewTable describes ONE sample (B = 1)")
        outs = []
        for b in range(B):
            feats = features if isinstance(features, ops.ViewTable) else [{k: v[b] for k, v in f.items()} for f in features]
            ctx = {k: v[b] for k, v in cnet_depth.items()}
            pairs = {k: v[b].contiguous() for k, v in proj_matrices.items()}
            outs.append(self._hot_single(feats, ctx, pairs, depth_values[b].contiguous(), want_intermediates))
        res = {"depth": [_stack([o["depth"][i] for o in outs]) for i in range(len(outs[0]["depth"]))],
               "photometric_confidence": _stack([o["photometric_confidence"] for o in outs])}
        if want_intermediates:
            res["intermediates"] = {k: _stack([o["intermediates"][k] for o in outs]) for k in outs[0]["intermediates"]}
        return res

    @ops.on_tensor_device

    def forward(self, imgs, proj_matrices, depth_values):
        # kept for callers that use it like the reference does (Effi_MVS_plus.py:423)
        disp_min = depth_values[:, 0, None, None, None]
        disp_max = depth_values[:, -1, None, None, None]
        self.scale_inv_depth = partial(disp_to_depth, min_depth=1. / disp_max, max_depth=1. / disp_min)
        self.scale_inv_depth.effi_disp_range = depth_values
        # the N + 1 pyramid passes are independent: with branches on, odd views and the context net go to the side stream (their coarse layers
        # do not fill the chip on their own)
        n_views = imgs.size(1)
        if self.training:
            # the reference's order (:432-435): BatchNorm running statistics are an exponential average, so the order of the N + 1
            # pyramid passes is part of the result
            features = [self.feature(imgs[:, v]) for v in range(n_views)]
            return self.forward_hot(features, self.cnet_depth(imgs[:, 0]), proj_matrices, depth_values)
        features = [None] * n_views
        with ops.Branch() as br:
            for v in range(1, n_views, 2):
                features[v] = self.feature(imgs[:, v])
            cnet_depth = self.cnet_depth(imgs[:, 0])
        for v in range(0, n_views, 2):
            features[v] = self.feature(imgs[:, v])
        br.join(*[t_ for v in range(1, n_views, 2) for t_ in features[v].values()], *cnet_depth.values())
        return self.forward_hot(features, cnet_depth, proj_matrices, depth_values)
