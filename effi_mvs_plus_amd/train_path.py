"""Scope row n2: the cost-volume path in TRAINING mode (``model.train()`` -> forward -> ``loss.backward()``, the reference's
train.py:229-263), composed from the differentiable operators of ``effi_mvs_plus_amd.autograd`` -- every convolution, BatchNorm
(batch statistics), activation / gating, lookup, soft-argmin, warp and upsampling runs forward and backward on the HIP kernels.

The functions take the drop-in modules of ``effi_mvs_plus_amd.models`` (they own the parameters and BatchNorm buffers) and batched
tensors, and follow the reference's op sequence -- including its detach points (update.py:121, Effi_MVS_plus.py:43,495) -- so that the
gradients equal torch autograd through the reference (checked against the training-mode oracle in tests/test_gpu_train.py).
Inference keeps its own fused launches (``Effi_MVS_plus.forward_hot``); nothing here is used in eval mode.
"""
import torch

from . import autograd as A
from . import ops

RELU, SIGMOID, TANH, NONE = ops.ACT_RELU, ops.ACT_SIGMOID, ops.ACT_TANH, ops.ACT_NONE


# ---- 3-D blocks (models/module.py:124-209,435-463,501-516) ------------------------------------------------------
def conv3d_block(mod, xs):
    """``Conv3d`` wrapper: conv (bias iff no BN) -> BatchNorm3d (batch statistics) -> ReLU."""
    if mod.conv.bias is not None:
        raise NotImplementedError("Conv3d with bias (bn=False) is not used on the path")
    y = A.conv3d(xs, mod.conv.weight, mod.conv.stride)
    if mod.bn is not None:
        return A.batch_norm_train(y, mod.bn, mod.relu)
    return A.activation(y, RELU) if mod.relu else y


def deconv3d_block(mod, x):
    if mod.conv.bias is not None:
        raise NotImplementedError("Deconv3d with bias (bn=False) is not used on the path")
    y = A.deconv3d(x, mod.conv.weight, mod.conv.stride)
    if mod.bn is not None:
        return A.batch_norm_train(y, mod.bn, mod.relu)
    return A.activation(y, RELU) if mod.relu else y


def cost_regnet(mod, x):
    """CostRegNet_2_sample_FPN3D_Fast.forward -> (prob [B,1,D,h,w], pro [B,8,D,h,w])  (models/module.py:450-463)."""
    c1 = conv3d_block(mod.conv1, [conv3d_block(mod.conv0, [x])])
    c3 = conv3d_block(mod.conv3, [conv3d_block(mod.conv2, [c1])])
    y = conv3d_block(mod.conv5, [conv3d_block(mod.conv4, [c3])])
    y = c3 + deconv3d_block(mod.conv6, y)
    pro = c1 + deconv3d_block(mod.conv7, y)
    prob = A.conv3d([pro], mod.prob.weight, 1)
    return prob, pro


def cost_up_small(mod, x, prior):
    """cost_up_small.forward -> (conv2, conv1)  (models/module.py:509-516); the concatenation is read in place."""
    c0 = conv3d_block(mod.conv0, [x])
    pc = conv3d_block(mod.conv_cost, [prior])
    c1 = conv3d_block(mod.conv1, [c0, pc])
    return deconv3d_block(mod.conv2, c1), c1


# ---- feature / context pyramid (scope row n1 in training; models/module.py:32-75,346-412) -----------------------------------
def fpn_block(blk, x):
    """``Conv2d`` wrapper of the pyramid: conv (3x3 stride 1 or 5x5 stride 2, no bias) -> BatchNorm2d (batch statistics) -> ReLU."""
    conv = blk.conv
    if conv.bias is not None or blk.bn is None or not isinstance(blk.bn, torch.nn.BatchNorm2d):
        raise NotImplementedError("feature pyramid (training): conv without bias + BatchNorm2d blocks, as P_1to8_FeatureNet_Fast builds them")
    if conv.kernel_size == (5, 5) and conv.stride == (2, 2) and conv.padding == (2, 2):
        y = A.conv2d_k5s2(x, conv.weight)
    elif conv.kernel_size == (3, 3) and conv.stride == (1, 1) and conv.padding == (1, 1):
        y = A.conv2d([x], conv.weight, None, NONE)
    else:
        raise NotImplementedError("feature pyramid (training): 3x3/s1/p1 and 5x5/s2/p2 blocks")
    return A.batch_norm_train(y, blk.bn, blk.relu)


def feature_pyramid(fpn, x):
    """P_1to8_FeatureNet_Fast.forward in training mode (models/module.py:392-412) on the differentiable HIP operators; the
    nearest-neighbour upsampling + add of the top-down path is tensor glue (torch)."""
    import torch.nn.functional as F
    for blk in fpn.conv0:
        x = fpn_block(blk, x)
    levels = []
    for seq in (fpn.conv1, fpn.conv2, fpn.conv3):
        for blk in seq:
            x = fpn_block(blk, x)
        levels.append(x)
    l1, l2, top = levels
    out = {"stage1": A.conv2d([top], fpn.out1.weight, fpn.out1.bias, NONE)}
    top = F.interpolate(top, scale_factor=2, mode="nearest") + A.conv2d([l2], fpn.inner1.weight, fpn.inner1.bias, NONE)
    out["stage2"] = A.conv2d([top], fpn.out2.weight, fpn.out2.bias, NONE)
    top = F.interpolate(top, scale_factor=2, mode="nearest") + A.conv2d([l1], fpn.inner2.weight, fpn.inner2.bias, NONE)
    out["stage3"] = A.conv2d([top], fpn.out3.weight, fpn.out3.bias, NONE)
    return out


# ---- view-weight net (models/Effi_MVS_plus.py:361-362) ---------------------------------------------------------------
def pixelwise_net(seq, entropy):
    x = entropy
    for i in range(3):
        x = A.batch_norm_train(A.conv2d([x], seq[i].conv.weight, None, NONE), seq[i].bn, True)
    return A.conv2d([x], seq[3].weight, seq[3].bias, SIGMOID)


# ---- GRU update block (models/update.py) --------------------------------------------------------------------------------
def projection_input(enc, disp, cost, context):
    cor = A.conv2d([cost], enc.convc1.weight, enc.convc1.bias, RELU)
    cor = A.conv2d([cor], enc.convc2.weight, enc.convc2.bias, RELU)
    dfm = A.conv2d([disp], enc.convd1.weight, enc.convd1.bias, RELU)
    dfm = A.conv2d([dfm], enc.convd2.weight, enc.convd2.bias, RELU)
    x = A.conv2d([cor, dfm], enc.convd.weight, enc.convd.bias, NONE)
    x = A.conv2d([x, context], enc.convc.weight, enc.convc.bias, RELU)
    return A.dropout2d(x, enc.dropout.p) if enc.dropout is not None else x          # update.py:97-98 (training)


def gru_gate_weights(gru):
    """convz | convr as ONE convolution (they read the same input, update.py:43-46): weights and biases concatenated along the output
    channels -- once per update block, so the three iterations share the packed operands and the gradient comes back through one
    slice.  Forward, input gradient and both weight gradients then run once instead of twice."""
    return torch.cat([gru.convz.weight, gru.convr.weight]), torch.cat([gru.convz.bias, gru.convr.bias])


def conv_gru(gru, h, *xs, gates=None):
    xs = list(xs)
    wzr, bzr = gates if gates is not None else gru_gate_weights(gru)
    z, r = A.split_channels(A.conv2d([h] + xs, wzr, bzr, SIGMOID), gru.convz.out_channels)
    q = A.conv2d([A._Mul.apply(r, h)] + xs, gru.convq.weight, gru.convq.bias, TANH)
    return A._GruCombine.apply(z, h, q)


def depth_head(head, x):
    out = A.conv2d([A.conv2d([x], head.conv1.weight, head.conv1.bias, RELU)], head.conv2.weight, head.conv2.bias, NONE)
    if head.dropout is not None:
        out = A.dropout2d(out, head.dropout.p)                                      # update.py:22-23 (training)
    return A.activation(out, TANH)


def mask_head(block, net):
    hid = A.conv2d([net], block.mask[0].weight, block.mask[0].bias, RELU)
    return 0.25 * A.conv2d([hid], block.mask[2].weight, block.mask[2].bias, NONE)   # update.py:136-137


def update_block(block, net, cost_fn, inv_depth, context, seq_len):
    """BasicUpdateBlock.forward (models/update.py:114-141): ``cost_fn(inv_depth, i)`` -> [B,2*nq,h,w]."""
    inv_list, mask_list = [], []
    gates = gru_gate_weights(block.depth_gru)
    for i in range(seq_len):
        inv_depth = inv_depth.detach()                                              # update.py:121
        x = projection_input(block.encoder, inv_depth, cost_fn(inv_depth, i), context)
        net = conv_gru(block.depth_gru, net, x, gates=gates)
        inv_depth = inv_depth + depth_head(block.depth_head, net)
        inv_list.append(inv_depth)
        mask_list.append(mask_head(block, net) if (block.UpMask and i == seq_len - 1) else inv_depth)
    return net, mask_list, inv_list


# ---- the cascade (models/Effi_MVS_plus.py:407-568 after the FPN) ------------------------------------------------------------
def depthnet(pixelwise_seq, cost_reg, feats, pairs, hyp):
    """DepthNet.forward (models/Effi_MVS_plus.py:14-89): feats list over views of [B,C,h,w]; pairs [B,N,2,4,4]; hyp [B,D] or
    [B,D,h,w] -> dict like the reference's."""
    B = feats[0].shape[0]
    sims, ents = [], []
    for b in range(B):
        s, e = A.warp_correlate(feats[0][b], [f[b] for f in feats[1:]], pairs[b], hyp[b], with_entropy=True)
        sims.append(s), ents.append(e)
    sim_views, entropy = A._stack(sims), A._stack(ents)                       # [B,S,D,h,w], [B,S,h,w] (detached, :43)
    S, h, w = entropy.shape[1:]
    # one call per source view, as in the reference's loop (:32-46): each call has its own batch statistics
    weights = torch.cat([pixelwise_net(pixelwise_seq, entropy[:, v:v + 1]) for v in range(S)], dim=1)
    volume = A.view_aggregate(sim_views, weights)                                   # [B,D,h,w]
    prob_pre, _ = cost_regnet(cost_reg, volume.unsqueeze(1))
    prob_pre = prob_pre.squeeze(1)
    depth, conf = A.soft_argmin(prob_pre, hyp)
    return {"depth": depth, "photometric_confidence": conf, "view_weights": weights, "reg_volume": prob_pre,
            "volume": volume.unsqueeze(1)}


def hot_path(model, features, cnet_depth, proj_matrices, depth_values):
    """Training-mode ``forward_hot``: same interface and outputs."""
    B, n_range = depth_values.shape
    lo, hi = depth_values[:, 0], depth_values[:, -1]
    if B > 1 and getattr(model, "static_depth_range", None) is None and (not torch.equal(lo, lo[:1].expand_as(lo))
                                                                         or not torch.equal(hi, hi[:1].expand_as(hi))):
        raise NotImplementedError("training path: the samples of a batch must share their depth range (DTU training does)")
    static_range = getattr(model, "static_depth_range", None)
    if static_range is not None:        # graph capture (train_graph.GraphedTrainStep): the range is a constant of the captured step
        lo_f, hi_f = static_range
    else:
        lo_f, hi_f = float(lo[0]), float(hi[0])
    d_nums = model.depth_stage_nums
    base_itv = (hi - lo) / n_range                                                  # :424
    hyp1, _ = zip(*[ops.stage1_hypotheses(depth_values[b].contiguous(), d_nums[0]) for b in range(B)])
    hyp1 = A._stack(list(hyp1))                                                        # [B,D1] depths
    depth_min_, depth_max_ = 1.0 / hi.view(B, 1, 1, 1), 1.0 / lo.view(B, 1, 1, 1)
    keys = ["stage{}".format(s + 1) for s in range(model.num_stage)]

    hidden, inp = [], []
    for s, k in enumerate(keys):
        hcat, ccat = torch.split(cnet_depth[k], [model.hdim_stage[s], model.cdim_stage[s]], dim=1)
        hidden.append(A.activation(hcat, TANH))
        inp.append(A.activation(ccat, RELU))

    preds, conf = [], None
    weights = reg_vol = cur_vol = None
    dmin_prev, dmax_prev = depth_min_, depth_max_
    for s, k in enumerate(keys):
        feats = [f[k] for f in features]
        pairs = proj_matrices[k]
        h, w = feats[0].shape[-2:]
        if s == 0:
            out = depthnet(model.PixelwiseNet, model.cost_regularization, feats, pairs, hyp1)
            conf = A._stack([ops.upsample_nearest(out["photometric_confidence"][b:b + 1].contiguous(), 4)[0] for b in range(B)])   # :478-480
            weights, reg_vol, cur_vol = out["view_weights"], out["reg_volume"], out["volume"].squeeze(1)
            preds.append(out["depth"])
            cur_depth = out["depth"].unsqueeze(1)
            dmin_cur, dmax_cur = depth_min_, depth_max_
        else:
            cur_depth = preds[-1].unsqueeze(1).detach()                             # :494-495
            D = d_nums[s]
            itv = base_itv * model.depth_interals_ratio[s]
            sims, smps = [], []
            for b in range(B):
                sm, sp = A.warp_correlate_dyn(feats[0][b], [f[b] for f in feats[1:]], weights[b], pairs[b], cur_depth[b, 0], itv[b], D)
                sims.append(sm), smps.append(sp)
            sim, samples = A._stack(sims), A._stack(smps)                     # [B,D,h,w]
            dmax_cur, dmin_cur = samples[:, 0:1], samples[:, -1:]                   # :508-509
            x5 = sim.unsqueeze(1)
            prior = A.vol_lookup(reg_vol, samples, dmin_prev, dmax_prev)            # queries read nearest-downsampled (:510)
            reg_vol = cost_up_small(model.CSP_R[s - 1], x5, prior.unsqueeze(1))[0].squeeze(1)
            prior = A.vol_lookup(cur_vol, samples, dmin_prev, dmax_prev)
            cur_vol = cost_up_small(model.CSP_C[s - 1], x5, prior.unsqueeze(1))[0].squeeze(1)
        # depth_to_disp (:538); the update block detaches it first thing (update.py:121), so no gradient leaves through it
        inv_cur = A._stack([ops.depth_to_inv(cur_depth[b].detach().contiguous(), depth_values[b].contiguous()) for b in range(B)])
        itv_s = base_itv * model.depth_interals_ratio[s]

        def cost_fn(inv, i, cur_vol=cur_vol, reg_vol=reg_vol, itv_s=itv_s, dmin_cur=dmin_cur, dmax_cur=dmax_cur):
            return A.getcost(cur_vol, reg_vol, inv, depth_values, itv_s, dmin_cur, dmax_cur, model.CostNum)

        _, masks, invs = update_block(model.update_block[s], hidden[s], cost_fn, inv_cur, inp[s], model.seq_len[s])
        for inv_i in invs:
            preds.append(A.inv_to_depth(inv_i, lo_f, hi_f).squeeze(1))
        up = A.convex_upsample(invs[-1], masks[-1]).unsqueeze(1)
        preds.append(A.inv_to_depth(up, lo_f, hi_f).squeeze(1))
        dmin_prev, dmax_prev = dmin_cur, dmax_cur
    return {"depth": preds, "photometric_confidence": conf}
