"""CPU oracle for the Effi-MVS+ cost-volume hot path.  TEST INFRASTRUCTURE ONLY.

This file restates, in plain functional torch (CPU, any float dtype), the arithmetic of the
reference's hot path so that the HIP kernels can be checked against it.  It is NOT part of the
product: only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import it, and only as the checker / the timed CPU baseline.  The product path
(``effi_mvs_plus_amd``) never imports it and fails loudly when the HIP library is missing.

Parity pinning: the reference (bdwsq1996/Effi-MVS-plus) ships no tests or golden vectors
(SURVEY.md section 4), so this oracle is pinned against the reference ITSELF, imported on CPU in the
build container (``tests/golden/make_golden.py`` -> ``tests/golden/*.npz``,
``tests/test_oracle_vs_reference.py``).  The arithmetic that is not spelled out below lives in
PyTorch itself (``F.grid_sample``, ``conv3d``, ``conv_transpose3d``, ``batch_norm``, ``softmax``,
``inverse``), version unpinned by the reference; "the reference" therefore means these ops as
implemented by the torch build in this image (2.10 CPU).

Style: weights come in as a flat ``state_dict`` (the reference's key names) and every function is
a pure function of tensors; nothing here is an ``nn.Module``.  All ``file:line`` citations are
relative to the reference repository root.  Eval-mode semantics by default (BatchNorm uses running
statistics, Dropout2d is the identity); ``with training(dropout_p):`` switches to the reference's
``model.train()`` semantics (batch statistics + running-stat update with momentum 0.1, Dropout2d at
update.py:22-23,97-98) so that torch autograd through these functions is the training-path checker
of scope row n2 (pinned against the imported reference in tests/test_oracle_vs_reference.py).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

BN_EPS = 1e-5  # nn.BatchNorm2d / nn.BatchNorm3d default, models/module.py:148,191,217

# The reference's hot path holds two value-neutral checks that force a device->host synchronisation each time they run on a
# GPU: the NaN probe of homo_warping_new (models/module.py:331-332, 12 per view) and the torch.unique assert of
# bilinear_sampler (models/Effi_MVS_plus.py:109, 22 per view: a device sort + .numel() on its result).  They never change a
# value, so the oracle omits them by default; ``with literal_syncs():`` re-enables them op for op -- bench.py's "literal
# reference-style PyTorch-ROCm baseline"  times the path that way.  SYNC_COUNT counts them.
LITERAL_SYNCS = False
SYNC_COUNT = 0

# model.train() semantics (train.py:229-263): BatchNorm on batch statistics (updating the running ones in ``sd`` in place),
# Dropout2d(p) where the reference has it.
TRAINING = False
DROPOUT_P = 0.1      # nn.Dropout2d(p=0.1), models/update.py:18,84
BN_MOMENTUM = 0.1    # Conv3d / Deconv3d / Conv2d wrappers pass bn_momentum=0.1 (module.py:35,127,171); nn.BatchNorm2d default too


class training:
    def __init__(self, dropout_p=0.1):
        self.p = dropout_p

    def __enter__(self):
        global TRAINING, DROPOUT_P
        self._before = (TRAINING, DROPOUT_P)
        TRAINING, DROPOUT_P = True, self.p
        return self

    def __exit__(self, *exc):
        global TRAINING, DROPOUT_P
        TRAINING, DROPOUT_P = self._before
        return False


def _dropout2d(x):
    """``self.dropout(x)`` under ``if self.training`` (models/update.py:22-23,97-98)."""
    if TRAINING and DROPOUT_P > 0:
        return F.dropout2d(x, DROPOUT_P, True)
    return x


class literal_syncs:
    def __enter__(self):
        global LITERAL_SYNCS, SYNC_COUNT
        self._before = LITERAL_SYNCS
        LITERAL_SYNCS, SYNC_COUNT = True, 0
        return self

    def __exit__(self, *exc):
        global LITERAL_SYNCS
        LITERAL_SYNCS = self._before
        return False


# --------------------------------------------------------------------------------------------
# small layer helpers
# --------------------------------------------------------------------------------------------
def _bn(x, sd, prefix):
    """Batch norm (models/module.py:148-157,217-220): running statistics in eval mode; batch statistics in training mode,
    where the running statistics in ``sd`` are updated in place exactly as nn.BatchNorm does."""
    if TRAINING:
        if prefix + ".num_batches_tracked" in sd:
            sd[prefix + ".num_batches_tracked"] += 1
        return F.batch_norm(x, sd[prefix + ".running_mean"], sd[prefix + ".running_var"],
                            sd[prefix + ".weight"], sd[prefix + ".bias"], True, BN_MOMENTUM, BN_EPS)
    return F.batch_norm(x, sd[prefix + ".running_mean"], sd[prefix + ".running_var"],
                        sd[prefix + ".weight"], sd[prefix + ".bias"], False, 0.0, BN_EPS)


def conv3d_block(x, sd, prefix, stride=1, padding=1, relu=True):
    """``Conv3d`` wrapper: conv (no bias) -> BN -> ReLU.  models/module.py:124-160."""
    y = F.conv3d(x, sd[prefix + ".conv.weight"], None, stride=stride, padding=padding)
    y = _bn(y, sd, prefix + ".bn")
    return F.relu(y) if relu else y


def deconv3d_block(x, sd, prefix, stride, padding, output_padding, relu=True):
    """``Deconv3d`` wrapper: transposed conv (no bias) -> BN -> ReLU.  models/module.py:168-203."""
    y = F.conv_transpose3d(x, sd[prefix + ".conv.weight"], None, stride=stride, padding=padding,
                           output_padding=output_padding)
    y = _bn(y, sd, prefix + ".bn")
    return F.relu(y) if relu else y


def conv_bn_relu_2d(x, sd, prefix):
    """``ConvBnReLU``: 3x3 conv (no bias, pad 1) -> BN2d -> ReLU.  models/module.py:213-220."""
    y = F.conv2d(x, sd[prefix + ".conv.weight"], None, stride=1, padding=1)
    return F.relu(_bn(y, sd, prefix + ".bn"))


def conv2d(x, sd, prefix, padding):
    """Plain ``nn.Conv2d`` with bias (models/update.py:14-15,36-38,73-81,109-112)."""
    return F.conv2d(x, sd[prefix + ".weight"], sd.get(prefix + ".bias"), padding=padding)


# --------------------------------------------------------------------------------------------
# a1: homography warp
# --------------------------------------------------------------------------------------------
def compose_projection(pair):
    """K . [R|t] written into the top 3x4 of the extrinsic.  models/Effi_MVS_plus.py:34-37,217-220.

    pair: [B, 2, 4, 4] with [:,0] = extrinsic 4x4 and [:,1,:3,:3] = intrinsic K.
    """
    out = pair[:, 0].clone()
    out[:, :3, :4] = torch.matmul(pair[:, 1, :3, :3], pair[:, 0, :3, :4])
    return out


def relative_projection(src_proj, ref_proj):
    """proj = src_proj . ref_proj^-1 -> (rot [B,3,3], trans [B,3,1]).  models/module.py:314-316."""
    proj = torch.matmul(src_proj, torch.inverse(ref_proj))
    return proj[:, :3, :3], proj[:, :3, 3:4]


def warp_grid(rot, trans, depth_values, height, width):
    """Normalised sampling grid [B, D, H*W, 2] for every (depth, pixel).  models/module.py:318-339.

    The reference's NaN probe (:331-332) only mutates proj_xyz after proj_xy was formed and never
    changes the result; it is omitted.
    """
    batch = rot.shape[0]
    num_depth = depth_values.shape[1]
    dt, dev = rot.dtype, rot.device
    ys, xs = torch.meshgrid(torch.arange(0, height, dtype=dt, device=dev),
                            torch.arange(0, width, dtype=dt, device=dev), indexing="ij")
    xyz = torch.stack((xs.reshape(-1), ys.reshape(-1), torch.ones(height * width, dtype=dt, device=dev)))
    xyz = xyz.unsqueeze(0).repeat(batch, 1, 1)                                  # [B,3,HW]
    rot_xyz = torch.matmul(rot, xyz)                                            # [B,3,HW]
    rot_depth_xyz = rot_xyz.unsqueeze(2).repeat(1, 1, num_depth, 1) * depth_values.reshape(batch, 1, num_depth, -1)
    proj_xyz = rot_depth_xyz + trans.view(batch, 3, 1, 1)                       # [B,3,D,HW]
    z = proj_xyz[:, 2:3]
    z = torch.where(z == 0, z + 1e-8, z)                                        # :328-329
    proj_xy = proj_xyz[:, :2] / z
    if LITERAL_SYNCS:                                                           # :331-332, after proj_xy exists: no effect on it
        global SYNC_COUNT
        SYNC_COUNT += 1
        if z.mean() != z.mean():
            z = z + 1e8
    gx = proj_xy[:, 0] / ((width - 1) / 2) - 1
    gy = proj_xy[:, 1] / ((height - 1) / 2) - 1
    return torch.stack((gx, gy), dim=3)                                         # [B,D,HW,2]


def homo_warping_new(src_fea, src_proj, ref_proj, depth_values):
    """Warp src features onto the reference view's depth hypotheses.  models/module.py:303-344.

    src_fea [B,C,H,W]; src_proj/ref_proj [B,4,4] (already K.[R|t]); depth_values [B,D] or [B,D,H,W]
    -> [B, C, D*H, W] (bilinear, zeros padding, align_corners=True).
    """
    batch, _, height, width = src_fea.shape
    num_depth = depth_values.shape[1]
    rot, trans = relative_projection(src_proj, ref_proj)
    if depth_values.dim() == 2:
        depth_values = depth_values.view(batch, num_depth, 1, 1).expand(batch, num_depth, height, width)
    grid = warp_grid(rot, trans, depth_values, height, width)
    return F.grid_sample(src_fea, grid.view(batch, num_depth * height, width, 2), mode="bilinear",
                         padding_mode="zeros", align_corners=True)


# --------------------------------------------------------------------------------------------
# a7: hypothesis sampling (in inverse depth)
# --------------------------------------------------------------------------------------------
def cur_depth_range_samples(cur, ndepth, interval):
    """Per-pixel hypotheses around ``cur`` [B,H,W] -> [B,D,H,W].  models/module.py:554-570."""
    lo = (cur - ndepth // 2 * interval).clamp(min=1e-4)
    hi = (cur + ndepth // 2 * interval).clamp(min=1e-4, max=1e4)
    step = (hi - lo) / (ndepth - 1)
    ar = torch.arange(0, ndepth, device=cur.device, dtype=cur.dtype).reshape(1, -1, 1, 1)
    return (lo.unsqueeze(1) + ar * step.unsqueeze(1)).clamp(min=1e-5)


def depth_range_samples(cur, ndepth, interval, shape):
    """``get_depth_range_samples``.  2-D input: linspace between first/last entry, repeated over the
    map; 3-D input: per-pixel hypotheses.  models/module.py:572-591."""
    if cur.dim() == 2:
        lo, hi = cur[:, 0], cur[:, -1]
        step = (hi - lo) / (ndepth - 1)
        ar = torch.arange(0, ndepth, device=cur.device, dtype=cur.dtype).reshape(1, -1)
        s = lo.unsqueeze(1) + ar * step.unsqueeze(1)
        return s.unsqueeze(-1).unsqueeze(-1).repeat(1, 1, shape[1], shape[2])
    return cur_depth_range_samples(cur, ndepth, interval)


# --------------------------------------------------------------------------------------------
# a10: inverse-depth <-> normalised coordinate, 1-D volume lookup
# --------------------------------------------------------------------------------------------
def disp_to_depth(disp, min_depth, max_depth):
    """models/Effi_MVS_plus.py:138-148 -> (scaled_disp, depth)."""
    min_disp = 1 / max_depth
    max_disp = 1 / min_depth
    scaled = (min_disp + (max_disp - min_disp) * disp).clamp(min=1e-4)
    return scaled, 1 / scaled


def depth_to_disp(depth, min_depth, max_depth):
    """models/Effi_MVS_plus.py:151-164."""
    scaled = 1 / depth
    min_disp = 1 / max_depth
    max_disp = 1 / min_depth
    return (scaled - min_disp) / ((max_disp - min_disp) + 1e-10)


def volume_lookup_1d(pro, depth_sample, depth_min, depth_max):
    """``pro_bilinear_sampler``: linear interpolation of a per-pixel D-vector at d query depths,
    zeros outside [0, D-1].  models/Effi_MVS_plus.py:102-134 (the torch.unique assert at :109 is a
    precondition check with no effect on the value and is omitted).

    pro [B*h*w,1,1,Dp]; depth_sample [B,d,h,w]; depth_min/max broadcastable to [B,1,h,w].
    """
    dp = pro.shape[-1]
    b, d, h, w = depth_sample.shape
    t = depth_to_disp(depth_sample, depth_min, depth_max) * (dp - 1)
    x0 = t.permute(0, 2, 3, 1).reshape(b * h * w, 1, d, 1)
    xg = 2 * x0 / (dp - 1) - 1
    y0 = torch.zeros_like(x0)
    if LITERAL_SYNCS:                                                           # models/Effi_MVS_plus.py:109
        global SYNC_COUNT
        SYNC_COUNT += 1
        assert torch.unique(y0).numel() == 1 and pro.shape[-2] == 1
    grid = torch.cat([xg, y0], dim=-1)
    out = F.grid_sample(pro, grid, align_corners=True)
    return out.reshape(b, h, w, -1).permute(0, 3, 1, 2)


def volume_lookup_1d_explicit(vol, depth_sample, depth_min, depth_max):
    """Same lookup written out as an explicit lerp on a [B,Dp,h,w] volume (cross-check of the
    grid_sample form above; used to document the semantics the HIP kernel implements)."""
    dp = vol.shape[1]
    t = depth_to_disp(depth_sample, depth_min, depth_max) * (dp - 1)
    g = 2 * t / (dp - 1) - 1
    ix = ((g + 1) / 2) * (dp - 1)
    i0 = torch.floor(ix)
    w1 = ix - i0
    w0 = (i0 + 1) - ix
    i0l = i0.long()
    i1l = i0l + 1
    v0 = torch.gather(vol, 1, i0l.clamp(0, dp - 1)) * ((i0l >= 0) & (i0l <= dp - 1))
    v1 = torch.gather(vol, 1, i1l.clamp(0, dp - 1)) * ((i1l >= 0) & (i1l <= dp - 1))
    return v0 * w0 + v1 * w1


# --------------------------------------------------------------------------------------------
# a3/a4/a9: weight net, 3-D regulariser, cross-scale propagation block
# --------------------------------------------------------------------------------------------
def pixelwise_net(sd, prefix, entropy):
    """View-weight net.  models/Effi_MVS_plus.py:361-362."""
    x = conv_bn_relu_2d(entropy, sd, prefix + ".0")
    x = conv_bn_relu_2d(x, sd, prefix + ".1")
    x = conv_bn_relu_2d(x, sd, prefix + ".2")
    x = F.conv2d(x, sd[prefix + ".3.weight"], sd[prefix + ".3.bias"])
    return torch.sigmoid(x)


def cost_regnet(sd, prefix, x):
    """``CostRegNet_2_sample_FPN3D_Fast.forward`` -> (prob, pro).  models/module.py:435-463."""
    conv1 = conv3d_block(conv3d_block(x, sd, prefix + ".conv0"), sd, prefix + ".conv1")
    conv3 = conv3d_block(conv3d_block(conv1, sd, prefix + ".conv2", stride=2), sd, prefix + ".conv3")
    y = conv3d_block(conv3d_block(conv3, sd, prefix + ".conv4", stride=2), sd, prefix + ".conv5")
    y = conv3 + deconv3d_block(y, sd, prefix + ".conv6", 2, 1, 1)
    pro = conv1 + deconv3d_block(y, sd, prefix + ".conv7", 2, 1, 1)
    prob = F.conv3d(pro, sd[prefix + ".prob.weight"], None, stride=1, padding=1)
    return prob, pro


def cost_up_small(sd, prefix, x, prior):
    """``cost_up_small.forward`` -> (conv2, conv1).  models/module.py:501-516."""
    c0 = conv3d_block(x, sd, prefix + ".conv0", stride=(1, 2, 2))
    pc = conv3d_block(prior, sd, prefix + ".conv_cost")
    c1 = conv3d_block(torch.cat([c0, pc], dim=1), sd, prefix + ".conv1")
    c2 = deconv3d_block(c1, sd, prefix + ".conv2", (1, 2, 2), 1, (0, 1, 1))
    return c2, c1


def depth_regression(p, depth_values):
    """models/module.py:518-524."""
    if depth_values.dim() <= 2:
        depth_values = depth_values.view(*depth_values.shape, 1, 1)
    return torch.sum(p * depth_values, 1)


# --------------------------------------------------------------------------------------------
# a2: stage-1 volume
# --------------------------------------------------------------------------------------------
def depthnet(sd, features, proj_matrices, depth_values, num_depth, regnet_prefix="cost_regularization",
             pixelwise_prefix="PixelwiseNet", G=1):
    """``DepthNet.forward`` (eval mode).  models/Effi_MVS_plus.py:14-89.  ``pixelwise_prefix=None`` = the reference's
    ``pixel_wise_net=None`` branch (:55-58,70): plain mean of the views' similarities, ``view_weights`` stays the empty list."""
    projs = torch.unbind(proj_matrices, 1)
    assert len(features) == len(projs)
    assert depth_values.shape[1] == num_depth
    ref, srcs = features[0], features[1:]
    B, C, H, W = ref.shape
    ref_g = ref.view(B, G, C // G, H, W)
    ref_new = compose_projection(projs[0])
    sim_sum, w_sum, view_weights = 0, 0, []
    for src, sp in zip(srcs, projs[1:]):
        warped = homo_warping_new(src, compose_projection(sp), ref_new, depth_values)
        warped = warped.view(B, G, C // G, num_depth, H, W)
        sim = (warped * ref_g.unsqueeze(3)).mean(2)                              # [B,G,D,H,W]
        if pixelwise_prefix is None:
            sim_sum = sim_sum + sim                                              # :55-58
            continue
        p = F.softmax(sim.squeeze(1).detach(), dim=1)                            # :43 (detached: no gradient through the entropy)
        entropy = (-p * torch.log(p + 1e-7)).sum(dim=1, keepdim=True)
        vw = pixelwise_net(sd, pixelwise_prefix, entropy)                        # [B,1,H,W]
        view_weights.append(vw)
        sim_sum = sim_sum + sim * vw.unsqueeze(1)
        w_sum = w_sum + vw.unsqueeze(1)
    if pixelwise_prefix is None:
        similarity = sim_sum / (len(features) - 1)                               # :70
    else:
        view_weights = torch.cat(view_weights, dim=1)
        similarity = sim_sum / (w_sum + 1e-6)
    prob_pre, _ = cost_regnet(sd, regnet_prefix, similarity)
    prob_pre = prob_pre.squeeze(1)
    prob = F.softmax(prob_pre, dim=1)
    depth = depth_regression(prob, depth_values)
    sum4 = 4 * F.avg_pool3d(F.pad(prob.unsqueeze(1), pad=(0, 0, 0, 0, 1, 2)), (4, 1, 1), stride=1, padding=0).squeeze(1)
    idx = depth_regression(prob, torch.arange(num_depth, device=prob.device, dtype=prob.dtype)).long()
    idx = idx.clamp(min=0, max=num_depth - 1)
    conf = torch.gather(sum4, 1, idx.unsqueeze(1)).squeeze(1)
    return {"depth": depth, "photometric_confidence": conf, "view_weights": view_weights,
            "reg_volume": prob_pre, "volume": similarity}


# --------------------------------------------------------------------------------------------
# a8/a11: stage-2/3 dynamic volume, per-iteration cost lookup
# --------------------------------------------------------------------------------------------
def getcost_initvolume(depth_values, features, proj_matrices, depth_interval, view_weights, cost_num, G=1):
    """``GetCost_initvolume.forward`` (Inverse=True, iter=1, inter_iter all ones).
    models/Effi_MVS_plus.py:184-251 -> (similarity [B,G*D,h,w], depth_range_samples [B,D,h,w])."""
    projs = torch.unbind(proj_matrices, 1)
    ref, srcs = features[0], features[1:]
    B, C, H, W = ref.shape
    inv = 1.0 / depth_values
    samples = depth_range_samples(inv.squeeze(1), cost_num, depth_interval.squeeze(1), [B, H, W])
    samples = 1.0 / samples
    ref_g = ref.view(B, G, C // G, H, W)
    ref_new = compose_projection(projs[0])
    sim_sum, w_sum = 0, 0
    for i, (src, sp) in enumerate(zip(srcs, projs[1:])):
        warped = homo_warping_new(src, compose_projection(sp), ref_new, samples)
        warped = warped.view(B, G, C // G, cost_num, H, W)
        sim = (warped * ref_g.unsqueeze(3)).mean(2)
        vw = view_weights[:, i].unsqueeze(1)
        sim_sum = sim_sum + sim * vw.unsqueeze(1)
        w_sum = w_sum + vw.unsqueeze(1)
    similarity = sim_sum / (w_sum + 1e-6)
    return similarity.view(B, G * cost_num, H, W), samples


def getcost(depth_values, pro, depth_interval, cost_num, depth_max_cur_volume, depth_min_cur_volume, shape):
    """``GetCost.forward`` (Inverse=True): 3 hypotheses around the current estimate, looked up in the
    cached cur/reg volumes.  models/Effi_MVS_plus.py:257-303 -> [B, 2*cost_num, h, w]."""
    inv = 1.0 / depth_values
    samples = depth_range_samples(inv.squeeze(1), cost_num, depth_interval.squeeze(1), shape)
    samples = 1.0 / samples
    a = volume_lookup_1d(pro[-1], samples, depth_min_cur_volume, depth_max_cur_volume)
    b = volume_lookup_1d(pro[0], samples, depth_min_cur_volume, depth_max_cur_volume)
    return torch.cat([a, b], dim=1)


# --------------------------------------------------------------------------------------------
# a12-a16: GRU update block, mask head, convex upsampling
# --------------------------------------------------------------------------------------------
def projection_input(sd, prefix, disp, cost, context):
    """``ProjectionInput.forward`` (eval).  models/update.py:86-99."""
    cor = F.relu(conv2d(cost, sd, prefix + ".convc1", 0))
    cor = F.relu(conv2d(cor, sd, prefix + ".convc2", 1))
    dfm = F.relu(conv2d(disp, sd, prefix + ".convd1", 3))
    dfm = F.relu(conv2d(dfm, sd, prefix + ".convd2", 1))
    x = conv2d(torch.cat([cor, dfm], dim=1), sd, prefix + ".convd", 1)
    x = conv2d(torch.cat([x, context], dim=1), sd, prefix + ".convc", 0)
    return _dropout2d(F.relu(x))                                      # update.py:95-98


def conv_gru(sd, prefix, h, x):
    """``ConvGRU.forward``.  models/update.py:40-49."""
    hx = torch.cat([h, x], dim=1)
    z = torch.sigmoid(conv2d(hx, sd, prefix + ".convz", 1))
    r = torch.sigmoid(conv2d(hx, sd, prefix + ".convr", 1))
    q = torch.tanh(conv2d(torch.cat([r * h, x], dim=1), sd, prefix + ".convq", 1))
    return (1 - z) * h + z * q


def depth_head(sd, prefix, x):
    """``DepthHead.forward`` (eval, act_fn=tanh).  models/update.py:20-27."""
    return torch.tanh(_dropout2d(conv2d(F.relu(conv2d(x, sd, prefix + ".conv1", 1)), sd, prefix + ".conv2", 1)))   # :21-27


def mask_head(sd, prefix, net):
    """0.25 * mask(net).  models/update.py:109-112,136-138."""
    return 0.25 * conv2d(F.relu(conv2d(net, sd, prefix + ".0", 1)), sd, prefix + ".2", 0)


def update_block(sd, prefix, net, depth_cost_func, inv_depth, context, seq_len, scale_inv_depth):
    """``BasicUpdateBlock.forward`` (eval, UpMask=True).  models/update.py:114-141."""
    inv_list, mask_list = [], []
    for i in range(seq_len):
        inv_depth = inv_depth.detach()                                  # update.py:121
        cost = depth_cost_func(scale_inv_depth(inv_depth)[1], i)
        x = projection_input(sd, prefix + ".encoder", inv_depth, cost, context)
        net = conv_gru(sd, prefix + ".depth_gru", net, x)
        inv_depth = inv_depth + depth_head(sd, prefix + ".depth_head", net)
        inv_list.append(inv_depth)
        mask_list.append(mask_head(sd, prefix + ".mask", net) if i == seq_len - 1 else inv_depth)
    return net, mask_list, inv_list


def upsample_depth(depth, mask, ratio=2):
    """Convex-combination upsampling.  models/Effi_MVS_plus.py:167-178."""
    N, _, H, W = depth.shape
    mask = torch.softmax(mask.view(N, 1, 9, ratio, ratio, H, W), dim=2)
    nb = F.unfold(depth, [3, 3], padding=1).view(N, 1, 9, 1, 1, H, W)
    up = torch.sum(mask * nb, dim=2).permute(0, 1, 4, 2, 5, 3)
    return up.reshape(N, ratio * H, ratio * W)


# --------------------------------------------------------------------------------------------
# out-of-scope neighbour needed to drive the path end to end: the FPN feature / context nets
# --------------------------------------------------------------------------------------------
def _conv2d_bn_relu(x, sd, prefix, stride, padding):
    """``Conv2d`` wrapper (bn=True, relu=True).  models/module.py:32-69."""
    y = F.conv2d(x, sd[prefix + ".conv.weight"], None, stride=stride, padding=padding)
    return F.relu(_bn(y, sd, prefix + ".bn"))


def feature_net(sd, prefix, x):
    """``P_1to8_FeatureNet_Fast.forward`` (stage_channel=True).  models/module.py:392-412."""
    c0 = _conv2d_bn_relu(_conv2d_bn_relu(x, sd, prefix + ".conv0.0", 1, 1), sd, prefix + ".conv0.1", 1, 1)
    lv = [c0]
    for name in ("conv1", "conv2", "conv3"):
        y = _conv2d_bn_relu(lv[-1], sd, f"{prefix}.{name}.0", 2, 2)
        y = _conv2d_bn_relu(y, sd, f"{prefix}.{name}.1", 1, 1)
        y = _conv2d_bn_relu(y, sd, f"{prefix}.{name}.2", 1, 1)
        lv.append(y)
    _, c1, c2, c3 = lv
    out = {"stage1": F.conv2d(c3, sd[prefix + ".out1.weight"])}
    f = F.interpolate(c3, scale_factor=2, mode="nearest") + F.conv2d(c2, sd[prefix + ".inner1.weight"], sd[prefix + ".inner1.bias"])
    out["stage2"] = F.conv2d(f, sd[prefix + ".out2.weight"], padding=1)
    f = F.interpolate(f, scale_factor=2, mode="nearest") + F.conv2d(c1, sd[prefix + ".inner2.weight"], sd[prefix + ".inner2.bias"])
    out["stage3"] = F.conv2d(f, sd[prefix + ".out3.weight"], padding=1)
    return out


# --------------------------------------------------------------------------------------------
# a17: the cascade
# --------------------------------------------------------------------------------------------
HDIM = (48, 32, 16)      # models/Effi_MVS_plus.py:337
CDIM = (12, 8, 4)        # :338
INTERVAL_RATIO = (4, 2, 1)  # :316 (depth_interals_ratio default)


def hot_path(sd, features, context, proj_matrices, depth_values, ndepths=(48, 8, 8), gru_iters=(3, 3, 3),
             cost_num=3, return_intermediates=False):
    """Stage loop of ``Effi_MVS_plus.forward`` from the point where per-view features and the
    context pyramid exist.  models/Effi_MVS_plus.py:409-424,437-568.

    features: list over views of {"stage1..3": [B,C,h,w]}; context: {"stage1..3": [B,hd+cd,h,w]}.
    """
    disp_min = depth_values[:, 0, None, None, None]
    disp_max = depth_values[:, -1, None, None, None]
    depth_max_ = 1.0 / disp_min
    depth_min_ = 1.0 / disp_max
    depth_max, depth_min = depth_max_, depth_min_
    depth_max2, depth_min2 = depth_max_, depth_min_

    def scale_inv(d):
        return disp_to_depth(d, depth_min_, depth_max_)

    depth_interval = (disp_max - disp_min) / depth_values.size(1)
    hidden, inp = [], []
    for s in range(3):
        h, c = torch.split(context[f"stage{s + 1}"], [HDIM[s], CDIM[s]], dim=1)
        hidden.append(torch.tanh(h))
        inp.append(torch.relu(c))

    preds, inter = [], {}
    conf = None
    view_weights = init_volume = reg_volume = cur_volume = None
    for s in range(3):
        feats = [f[f"stage{s + 1}"] for f in features]
        pm = proj_matrices[f"stage{s + 1}"]
        ref = feats[0]
        B, _, H, W = ref.shape
        if s == 0:
            samples = depth_range_samples(depth_values, ndepths[0], INTERVAL_RATIO[0] * depth_interval, [B, H, W])
            samples = 1.0 / samples
            out = depthnet(sd, feats, pm, samples, ndepths[0])
            conf = F.interpolate(out["photometric_confidence"].unsqueeze(1), [H * 4, W * 4], mode="nearest").squeeze(1)
            view_weights = out["view_weights"]
            init_volume = out["volume"]
            cur_volume = init_volume.squeeze(1)
            reg_volume = out["reg_volume"]
            cur_depth = out["depth"].unsqueeze(1)
            preds = [out["depth"]]
        else:
            cur_depth = preds[-1].unsqueeze(1).detach()                # models/Effi_MVS_plus.py:494-495
            view_weights = F.interpolate(view_weights, scale_factor=2, mode="nearest")
            cur_volume, samples_ = getcost_initvolume(cur_depth, feats, pm, depth_interval * INTERVAL_RATIO[s],
                                                      view_weights, ndepths[s])
            depth_max2 = samples_[:, 0:1]
            depth_min2 = samples_[:, -1:]
            D = samples_.shape[1]
            low = F.interpolate(samples_.unsqueeze(1), size=[D, H // 2, W // 2], mode="nearest").squeeze(1)
            x5 = cur_volume.view(B, 1, D, H, W)
            dp = ndepths[s - 1]
            pro = reg_volume.permute(0, 2, 3, 1).reshape(B * (H // 2) * (W // 2), 1, 1, dp)
            prior = volume_lookup_1d(pro, low, depth_min, depth_max)
            reg_volume = cost_up_small(sd, f"CSP_R.{s - 1}", x5, prior.unsqueeze(1))[0].squeeze(1)
            iv = init_volume.squeeze(1).permute(0, 2, 3, 1).reshape(B * (H // 2) * (W // 2), 1, 1, dp)
            prior = volume_lookup_1d(iv, low, depth_min, depth_max)
            init_volume = cost_up_small(sd, f"CSP_C.{s - 1}", x5, prior.unsqueeze(1))[0]
            cur_volume = init_volume.squeeze(1)
            depth_max, depth_min = depth_max2, depth_min2
        if return_intermediates:
            inter[f"view_weights{s + 1}"] = view_weights
            inter[f"reg_volume{s + 1}"] = reg_volume
            inter[f"cur_volume{s + 1}"] = cur_volume
        inv_cur = depth_to_disp(cur_depth, depth_min_, depth_max_)
        Bv, Dv, Hv, Wv = reg_volume.shape
        pro = [reg_volume.permute(0, 2, 3, 1).reshape(Bv * Hv * Wv, 1, 1, Dv),
               cur_volume.permute(0, 2, 3, 1).reshape(Bv * Hv * Wv, 1, 1, cur_volume.shape[1])]
        interval_s = depth_interval * INTERVAL_RATIO[s]
        dmx, dmn = depth_max2, depth_min2

        def cost_func(depth, it, pro=pro, interval_s=interval_s, dmx=dmx, dmn=dmn, shape=[B, H, W]):
            return getcost(depth, pro, interval_s, cost_num, dmx, dmn, shape)

        _, masks, invs = update_block(sd, f"update_block.{s}", hidden[s], cost_func, inv_cur, inp[s],
                                      gru_iters[s], scale_inv)
        for inv_i in invs:
            preds.append(scale_inv(inv_i)[1].squeeze(1))
        up = upsample_depth(invs[-1], masks[-1], ratio=2).unsqueeze(1)
        preds.append(scale_inv(up)[1].squeeze(1))
    res = {"depth": preds, "photometric_confidence": conf}
    if return_intermediates:
        res["intermediates"] = inter
    return res


def mvs_loss(inputs, depth_gt_ms, mask_ms, dloss, loss_rate=0.9):
    """Smooth-L1 over the cascade's outputs, geometric weights.  models/module.py:526-552 -> (total, {l_i})."""
    total = torch.tensor(0.0, dtype=torch.float32, device=mask_ms["stage1"].device)
    per = {}
    n = len(inputs)
    for i, est in enumerate(inputs):
        key = "stage{}".format(dloss[i])
        mask = mask_ms[key] > 0.5
        li = F.smooth_l1_loss(est[mask], depth_gt_ms[key][mask], reduction="mean")
        per["l{}".format(i)] = li
        total = total + (1.0 if i == 0 else loss_rate ** (n - i - 1)) * li
    return total, per


def full_forward(sd, imgs, proj_matrices, depth_values, **kw):
    """``Effi_MVS_plus.forward`` including the (out-of-scope) FPN nets.  models/Effi_MVS_plus.py:407-568."""
    features = [feature_net(sd, "feature", imgs[:, v]) for v in range(imgs.size(1))]
    context = feature_net(sd, "cnet_depth", imgs[:, 0])
    return hot_path(sd, features, context, proj_matrices, depth_values, **kw)


# =============================================================================================
# n3: dynamic geometric-consistency filter + depth averaging of the Tanks-and-Temples driver
# (reference: misc/fusion.py:8-46,117-181 and the tensor part of test_tank.py:466-512).  Device agnostic
# (the reference's get_pixel_grids hard-codes .cuda()); same operations in the same order, fp32.
# =============================================================================================
def fusion_pixel_grids(height, width, device=None):
    """misc/fusion.py:8-13 -> [h,w,3,1] pixel centres (x+0.5, y+0.5, 1)."""
    x = (torch.arange(width, dtype=torch.float32, device=device) + 0.5).repeat(height, 1)
    y = (torch.arange(height, dtype=torch.float32, device=device) + 0.5).repeat(width, 1).t()
    return torch.stack([x, y, torch.ones_like(x)], dim=-1).unsqueeze(-1)


# The reference writes every per-pixel 3x3 / 4x4 transform as a broadcast ``@`` ([N,1,1,r,c] @ [1,h,w,c,1]), which PyTorch
# lowers to ONE batched GEMM with N*h*w batches of an (r x c)(c x 1) product -- 18.9 M batches at 1600x1184 with 10 source
# views.  On CPU that is fine (and is what the golden vectors pin).  On PyTorch-ROCm the round-1 bench aborted with a GPU memory
# access fault inside that call ("the fault of round 1"), so the GPU baseline leg of bench.py evaluates the
# same products element-wise (``with elementwise_mm():``): sum_c A[..., r, c] * x[..., c, 0], no batched GEMM.
ELEMENTWISE_MM = False


class elementwise_mm:
    def __enter__(self):
        global ELEMENTWISE_MM
        self._before = ELEMENTWISE_MM
        ELEMENTWISE_MM = True
        return self

    def __exit__(self, *exc):
        global ELEMENTWISE_MM
        ELEMENTWISE_MM = self._before
        return False


def _mm(a, x):
    """a [..., r, c] @ x [..., c, 1] with broadcasting over the leading dimensions."""
    if not ELEMENTWISE_MM:
        return a @ x
    return (a * x.transpose(-1, -2)).sum(dim=-1, keepdim=True)


def fusion_idx_img2cam(idx_img_homo, depth, cam):
    """misc/fusion.py:23-28."""
    idx_cam = _mm(cam[:, 1:2, :3, :3].unsqueeze(1).inverse(), idx_img_homo)
    idx_cam = idx_cam / (idx_cam[..., -1:, :] + 1e-9) * depth.permute(0, 2, 3, 1).unsqueeze(4)
    return torch.cat([idx_cam, torch.ones_like(idx_cam[..., -1:, :])], dim=-2)


def fusion_idx_cam2world(idx_cam_homo, cam):
    """misc/fusion.py:31-34."""
    w = _mm(cam[:, 0:1, ...].unsqueeze(1).inverse(), idx_cam_homo)
    return w / (w[..., -1:, :] + 1e-9)


def fusion_idx_world2cam(idx_world_homo, cam):
    """misc/fusion.py:37-40."""
    c = _mm(cam[:, 0:1, ...].unsqueeze(1), idx_world_homo)
    return c / (c[..., -1:, :] + 1e-9)


def fusion_idx_cam2img(idx_cam_homo, cam):
    """misc/fusion.py:43-47."""
    idx_cam = idx_cam_homo[..., :3, :] / (idx_cam_homo[..., 3:4, :] + 1e-9)
    img = _mm(cam[:, 1:2, :3, :3].unsqueeze(1), idx_cam)
    return img / (img[..., -1:, :] + 1e-9)


def fusion_get_reproj_dynamic(ref_depth, srcs_depth, ref_cam, srcs_cam):
    """misc/fusion.py:117-156: ref pixel -> src view (bilinear depth sample) -> back to the ref view.
    ref_depth [n,1,h,w], srcs_depth [n,v,1,h,w], cams [n,(v,)2,4,4] -> (reproj_xyd [n,v,3,h,w], ref_idx_cam, src2ref_idx_cam)."""
    n, v, _, h, w = srcs_depth.size()
    srcs_depth_f = srcs_depth.reshape(n * v, 1, h, w)
    srcs_cam_f = srcs_cam.reshape(n * v, 2, 4, 4)
    ref_cam_r = ref_cam.unsqueeze(1).repeat(1, v, 1, 1, 1).view(n * v, 2, 4, 4)
    ref_depth_f = ref_depth.unsqueeze(1).repeat(1, v, 1, 1, 1).view(n * v, 1, h, w)
    idx_img = fusion_pixel_grids(h, w, ref_depth.device).unsqueeze(0)
    ref_idx_cam = fusion_idx_img2cam(idx_img, ref_depth_f, ref_cam_r)
    ref_idx_world = fusion_idx_cam2world(ref_idx_cam, ref_cam_r)
    ref2src_idx_cam = fusion_idx_world2cam(ref_idx_world, srcs_cam_f)
    ref2src_idx_img = fusion_idx_cam2img(ref2src_idx_cam, srcs_cam_f)
    warp_coord = ref2src_idx_img[..., :2, 0]
    px = warp_coord[..., 0] / ((w - 1) / 2) - 1
    py = warp_coord[..., 1] / ((h - 1) / 2) - 1
    warped = F.grid_sample(srcs_depth_f, torch.stack((px, py), dim=-1), mode="bilinear", padding_mode="zeros", align_corners=True)
    warp_homo = torch.cat([warp_coord, torch.ones_like(warp_coord[..., -1:])], dim=-1).unsqueeze(-1)
    src_idx_cam = fusion_idx_img2cam(warp_homo, warped, srcs_cam_f)
    src_idx_world = fusion_idx_cam2world(src_idx_cam, srcs_cam_f)
    src2ref_idx_cam = fusion_idx_world2cam(src_idx_world, ref_cam_r)
    reproj_depth = src2ref_idx_cam[:, :, :, 2, 0].clone()
    src2ref_img = fusion_idx_cam2img(src2ref_idx_cam, ref_cam_r)
    xyd = torch.cat([src2ref_img[..., :2, 0], reproj_depth.unsqueeze(-1)], dim=-1).permute(0, 3, 1, 2)
    return xyd.reshape(n, v, 3, h, w), ref_idx_cam, src2ref_idx_cam


def fusion_vis_filter_dynamic(ref_depth, reproj_xyd, dist_base=4, rel_diff_base=1300, thres_view=2, relative=False):
    """misc/fusion.py:159-181 -> (masks [n,v,v+1-thres_view,h,w] bool, mask = masks of the loosest threshold [n,v,1,h,w])."""
    n, v, _, h, w = reproj_xyd.size()
    xy = fusion_pixel_grids(h, w, reproj_xyd.device).permute(3, 2, 0, 1).unsqueeze(1)[:, :, :2]
    corrd_diff = (reproj_xyd[:, :, :2, :, :] - xy).norm(dim=2, keepdim=True)
    depth_diff = (ref_depth.unsqueeze(1) - reproj_xyd[:, :, 2:, :, :]).abs()
    if relative:
        depth_diff = depth_diff / ref_depth.unsqueeze(1)
    steps = torch.arange(thres_view, v + 1, device=reproj_xyd.device).reshape(1, 1, -1, 1, 1).repeat(n, v, 1, 1, 1)
    masks = torch.min(corrd_diff < steps / dist_base, depth_diff < steps / rel_diff_base)
    return masks, masks[:, :, -1:, :, :]


def fusion_dynamic_filter(ref_depth, src_depths, ref_cam, src_cams, ref_conf, prob_threshold, dh_view_num, dist_filter,
                          depth_filter, relative=False):
    """Tensor part of ``dynamic_filter_depth`` (test_tank.py:466-512) for one reference view.
    ref_depth [n,1,h,w]; src_depths [n,v,1,h,w]; ref_conf [n,H,W] -> dict(depth [n,1,h,w] averaged depth, geo_mask / prob_mask /
    mask [n,1,h,w] bool, points [n,3,h,w] world coordinates of the averaged depth)."""
    n, v, _, h, w = src_depths.size()
    dy_range = v + 1
    conf = F.interpolate(ref_conf.unsqueeze(1), size=[h, w], mode="nearest")
    prob_mask = (conf > prob_threshold).squeeze(1)
    reproj_xyd, _, _ = fusion_get_reproj_dynamic(ref_depth, src_depths, ref_cam, src_cams)
    vis_masks, vis_mask = fusion_vis_filter_dynamic(ref_depth, reproj_xyd, dist_base=dist_filter, rel_diff_base=depth_filter,
                                                    thres_view=dh_view_num, relative=relative)
    reproj_depth = reproj_xyd[:, :, -1].clone()
    reproj_depth[~vis_mask.squeeze(2)] = 0
    geo_mask_sums = vis_masks.sum(dim=1)
    geo_mask_sum = vis_mask.sum(dim=1)
    depth_avg = (torch.sum(reproj_depth, dim=1, keepdim=True) + ref_depth) / (geo_mask_sum + 1)
    geo_mask = geo_mask_sum >= dy_range
    for i in range(dh_view_num, dy_range):
        geo_mask = torch.logical_or(geo_mask, geo_mask_sums[:, i - dh_view_num] >= i)
    mask = torch.min(prob_mask, geo_mask)
    idx_img = fusion_pixel_grids(h, w, ref_depth.device).unsqueeze(0)
    idx_cam = fusion_idx_img2cam(idx_img, depth_avg, ref_cam)
    points = fusion_idx_cam2world(idx_cam, ref_cam)[..., :3, 0].permute(0, 3, 1, 2)
    return {"depth": depth_avg, "geo_mask": geo_mask.reshape(n, 1, h, w), "prob_mask": prob_mask.reshape(n, 1, h, w),
            "mask": mask.reshape(n, 1, h, w), "points": points, "reproj_xyd": reproj_xyd}
