"""Drop-in mirror of the reference's ``models/module.py`` for the cost-volume hot path.

Same public names, constructor/forward signatures and state-dict keys as the reference
(``models/module.py`` in bdwsq1996/Effi-MVS-plus), so its checkpoints load with ``strict=True`` and its
drivers call these classes unchanged; the arithmetic runs in the gfx950 kernels behind
``include/effi_mvs_hip.h``.  Inference only (eval mode, fp32, CUDA tensors): anything else raises --
there is no eager fallback on this path.
"""
from __future__ import annotations


import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops, packing


def _require_eval(mod):
    """The fused single-sample launches (``run`` methods) are the inference form of a module; in training mode ``forward`` goes
    through ``effi_mvs_plus_amd.train_path`` (differentiable operators, BatchNorm on batch statistics) instead."""
    if mod.training:
        raise NotImplementedError(
            f"{type(mod).__name__}.run is the fused inference launch; in training mode call the module itself (its forward takes "
            "the differentiable path of effi_mvs_plus_amd.train_path), or model.eval() for inference.")


def _stack(ts):
    """Re-attach the batch dimension (a view when B == 1)."""
    return ts[0].unsqueeze(0) if len(ts) == 1 else torch.stack(ts)


def _triple(v):
    return tuple(v) if isinstance(v, (tuple, list)) else (v, v, v)


# =============================================================================================
# 3-D conv wrappers (reference: models/module.py:124-166 Conv3d, :168-209 Deconv3d)
# =============================================================================================
class Conv3d(nn.Module):
    """nn.Conv3d (bias iff no BN) -> BatchNorm3d -> ReLU, kernel 3 / padding 1, stride 1 or 2 per axis.

    State-dict keys: ``conv.weight`` [, ``conv.bias``], ``bn.{weight,bias,running_mean,running_var,
    num_batches_tracked}`` -- identical to the reference.
    """

    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, relu=True, bn=True, bn_momentum=0.1,
                 init_method="xavier", **kwargs):
        super().__init__()
        self.out_channels = out_channels
        self.kernel_size = kernel_size
        self.stride = stride
        self.conv = nn.Conv3d(in_channels, out_channels, kernel_size, stride=stride, bias=(not bn), **kwargs)
        self.bn = nn.BatchNorm3d(out_channels, momentum=bn_momentum) if bn else None
        self.relu = relu
        self._cache = packing.PackCache()
        self._cache_planes = packing.PackCache()
        self._cache_planes_x3 = packing.PackCache()
        self._cache_roll = packing.PackCache()
        self._cache_s2_x3 = packing.PackCache()

    def _packed(self):
        t = [self.conv.weight, self.conv.bias]
        if self.bn is not None:
            t += [self.bn.weight, self.bn.bias, self.bn.running_mean, self.bn.running_var]
        return self._cache.get(t, lambda: packing.pack_conv3d(self.conv, self.bn))

    def run(self, srcs, skip=None):
        """srcs: list of unbatched planar [Ci,D,h,w] tensors (channel concatenation) -> [cout,Do,ho,wo]."""
        _require_eval(self)
        if _triple(self.conv.kernel_size) != (3, 3, 3) or _triple(self.conv.padding) != (1, 1, 1):
            raise NotImplementedError("Conv3d: only kernel 3 / padding 1 is instantiated on the HIP path")
        t = None
        cin_total = sum(x.shape[0] for x in srcs)
        if (skip is None and _triple(self.conv.stride) == (1, 1, 1) and 1 < self.out_channels <= 32 and len(srcs) <= 2
                and cin_total in (8, 16) and srcs[0].shape[0] % 8 == 0 and srcs[0].shape[-1] % 4 == 0
                and ops.uses_split() and ops.option("roll") != 0):
            # 8 / 16 input channels: rolling window of input planes in LDS, each plane fetched once
            t = [self.conv.weight, self.conv.bias]
            if self.bn is not None:
                t += [self.bn.weight, self.bn.bias, self.bn.running_mean, self.bn.running_var]
            wp, bp = self._cache_roll.get(t, lambda: packing.pack_conv3d_roll_bf16x3(self.conv, self.bn))
            return ops.conv3d_k3s1_roll(srcs, wp, bp, self.out_channels, relu=self.relu)
        if (skip is None and _triple(self.conv.stride) == (1, 1, 1) and 1 < self.out_channels <= 32
                and self.conv.in_channels >= 8 and self.conv.in_channels % 8 == 0 and ops.uses_split()
                and (srcs[0].shape[-1] % 4 == 0 or ops.option("conv3d_unaligned_split") != 0)):
            # stride-1 layers with >= 8 input channels: z-batched 2-D convolutions on the bf16 matrix cores in split
            # precision (single-channel inputs stay on the vector kernel: 3 of 16 K-slots used, measured slower)
            t = [self.conv.weight, self.conv.bias]
            if self.bn is not None:
                t += [self.bn.weight, self.bn.bias, self.bn.running_mean, self.bn.running_var]
            wp, bp = self._cache_planes_x3.get(t, lambda: packing.pack_conv3d_planes_bf16x3(self.conv, self.bn))
            return ops.conv3d_k3s1_bf16x3(srcs, wp, bp, self.out_channels, relu=self.relu)
        if (len(srcs) == 1 and skip is None and _triple(self.conv.stride) == (1, 1, 1) and self.out_channels in (16, 32)
                and self.conv.in_channels >= 8):
            # low-resolution U-Net levels: z-batched 2-D convolutions on the matrix cores
            t = [self.conv.weight, self.conv.bias]
            if self.bn is not None:
                t += [self.bn.weight, self.bn.bias, self.bn.running_mean, self.bn.running_var]
            wp, bp = self._cache_planes.get(t, lambda: packing.pack_conv3d_planes(self.conv, self.bn))
            return ops.conv3d_k3s1_mfma(srcs[0], wp, bp, self.out_channels, relu=self.relu)
        if (len(srcs) == 1 and skip is None and _triple(self.conv.stride) == (2, 2, 2) and self.out_channels <= 64
                and self.conv.in_channels >= 8 and srcs[0].shape[-1] % 4 == 0 and ops.uses_split()
                and ops.option("conv3d_s2_split") != 0):
            # down-sampling U-Net levels in split precision (column parities de-interleaved in LDS, as the pyramid's 5x5 stride-2 layers)
            t = [self.conv.weight, self.conv.bias]
            if self.bn is not None:
                t += [self.bn.weight, self.bn.bias, self.bn.running_mean, self.bn.running_var]
            wp, bp = self._cache_s2_x3.get(t, lambda: packing.pack_conv3d_s2_bf16x3(self.conv, self.bn))
            return ops.conv3d_k3s2_x3(srcs[0], wp, bp, self.out_channels, relu=self.relu)
        if (len(srcs) == 1 and skip is None and _triple(self.conv.stride) == (2, 2, 2) and self.out_channels in (16, 32)
                and self.conv.in_channels >= 8):
            # down-sampling U-Net levels: z-batched stride-2 2-D convolutions on the matrix cores
            t = [self.conv.weight, self.conv.bias]
            if self.bn is not None:
                t += [self.bn.weight, self.bn.bias, self.bn.running_mean, self.bn.running_var]
            wp, bp = self._cache_planes.get(t, lambda: packing.pack_conv3d_planes(self.conv, self.bn))
            return ops.conv3d_k3s2_mfma(srcs[0], wp, bp, self.out_channels, relu=self.relu)
        w, b = self._packed()
        return ops.conv3d_k3(srcs, w, b, self.out_channels, stride=_triple(self.conv.stride), relu=self.relu, skip=skip)

    @ops.on_tensor_device

    def forward(self, x):
        if self.training:                      # BatchNorm on batch statistics, differentiable: forward AND backward on HIP kernels
            from .. import train_path
            return train_path.conv3d_block(self, [x])
        return _stack([self.run([x[i].contiguous()]) for i in range(x.shape[0])])


class Deconv3d(nn.Module):
    """nn.ConvTranspose3d -> BatchNorm3d -> ReLU; kernel 3, padding 1, stride (s,2,2), output_padding (s-1,1,1)."""

    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, relu=True, bn=True, bn_momentum=0.1,
                 init_method="xavier", **kwargs):
        super().__init__()
        self.out_channels = out_channels
        self.stride = stride
        self.conv = nn.ConvTranspose3d(in_channels, out_channels, kernel_size, stride=stride, bias=(not bn), **kwargs)
        self.bn = nn.BatchNorm3d(out_channels, momentum=bn_momentum) if bn else None
        self.relu = relu
        self._cache = packing.PackCache()
        self._cache_x3 = packing.PackCache()

    def _packed(self):
        t = [self.conv.weight, self.conv.bias]
        if self.bn is not None:
            t += [self.bn.weight, self.bn.bias, self.bn.running_mean, self.bn.running_var]
        return self._cache.get(t, lambda: packing.pack_deconv3d(self.conv, self.bn))

    def run(self, x, skip=None):
        _require_eval(self)
        st, pad, op = _triple(self.conv.stride), _triple(self.conv.padding), _triple(self.conv.output_padding)
        if _triple(self.conv.kernel_size) != (3, 3, 3) or pad != (1, 1, 1) or st[1:] != (2, 2) or \
                op != (st[0] - 1, 1, 1) or st[0] not in (1, 2):
            raise NotImplementedError("Deconv3d: only k3 / p1 / stride (s,2,2) / output_padding (s-1,1,1) is instantiated")
        if st == (2, 2, 2) and x.shape[0] % 16 == 0 and self.out_channels <= 16 and ops.uses_split():
            t = [self.conv.weight, self.conv.bias]
            if self.bn is not None:
                t += [self.bn.weight, self.bn.bias, self.bn.running_mean, self.bn.running_var]
            wp, bp = self._cache_x3.get(t, lambda: packing.pack_deconv3d_s2_bf16x3(self.conv, self.bn))
            return ops.deconv3d_k3s2_x3(x, wp, bp, self.out_channels, relu=self.relu, skip=skip)
        w, b = self._packed()
        return ops.deconv3d_k3(x, w, b, self.out_channels, sz=st[0], relu=self.relu, skip=skip)

    @ops.on_tensor_device

    def forward(self, x):
        if self.training:
            from .. import train_path
            return train_path.deconv3d_block(self, x)
        return _stack([self.run(x[i].contiguous()) for i in range(x.shape[0])])


# =============================================================================================
# 2-D wrappers used by the view-weight net and by the (stock, out-of-scope) FPN
# =============================================================================================
class ConvBnReLU(nn.Module):
    """3x3 conv (no bias) + BN + ReLU (reference: models/module.py:213-220).  Holds parameters for the
    fused view-weight kernel; called on its own it runs the stock torch ops (it is not on the hot path
    individually)."""

    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, pad=1):
        super().__init__()
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size, stride=stride, padding=pad, bias=False)
        self.bn = nn.BatchNorm2d(out_channels)

    @ops.on_tensor_device

    def forward(self, x):
        return F.relu(self.bn(self.conv(x)), inplace=True)


class Conv2d(nn.Module):
    """2-D conv + BN + ReLU block of the feature pyramid (reference: models/module.py:32-75)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, relu=True, bn=True, bn_momentum=0.1,
                 norm_type="BN", init_method="xavier", **kwargs):
        super().__init__()
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size, stride=stride, bias=(not bn), **kwargs)
        self.kernel_size = kernel_size
        self.stride = stride
        if not bn:
            self.bn = None
        elif norm_type == "IN":
            self.bn = nn.InstanceNorm2d(out_channels, momentum=bn_momentum)
        else:
            self.bn = nn.BatchNorm2d(out_channels, momentum=bn_momentum)
        self.relu = relu

    @ops.on_tensor_device

    def forward(self, x):
        x = self.conv(x)
        if self.bn is not None:
            x = self.bn(x)
        return F.relu(x, inplace=True) if self.relu else x


class P_1to8_FeatureNet_Fast(nn.Module):
    """Feature / context pyramid (reference: models/module.py:346-412); outputs {stage1: 1/8, stage2: 1/4,
    stage3: 1/2 resolution}.  First "next" row of the scope table: in eval mode on CUDA
    tensors every layer runs on the fp32 matrix cores (BatchNorm folded; 3x3 / 5x5-stride-2 / 1x1 implicit-GEMM
    kernels of csrc/conv2d.hip; the top-down ``upsample + lateral 1x1`` is one fused kernel).  ``forward_torch`` is
    the stock PyTorch composite of the same layers (training, and the A/B baseline of tools/fpn_time.py)."""

    def __init__(self, base_channels=8, in_channel=[8, 16, 32, 64], out_channel=[32, 16, 8], stage_channel=True):
        super().__init__()
        self.base_channels = base_channels
        # eval/HIP path only: emit the stage maps channel-last in memory (same [B,C,h,w] shape, torch.channels_last
        # strides) so the warp kernels take them without a transpose; set by Effi_MVS_plus for its feature net
        self.channels_last_outputs = False
        c0, c1, c2, c3 = in_channel

        def level(cin, cout):
            return nn.Sequential(Conv2d(cin, cout, 5, stride=2, padding=2), Conv2d(cout, cout, 3, 1, padding=1),
                                 Conv2d(cout, cout, 3, 1, padding=1))

        self.conv0 = nn.Sequential(Conv2d(3, c0, 3, 1, padding=1), Conv2d(c0, c0, 3, 1, padding=1))
        self.conv1 = level(c0, c1)
        self.conv2 = level(c1, c2)
        self.conv3 = level(c2, c3)
        o1, o2, o3 = (out_channel[0], out_channel[1], out_channel[2]) if stage_channel else (out_channel[1],) * 3
        self.out1 = nn.Conv2d(c3, o1, 1, bias=False)
        self.inner1 = nn.Conv2d(c2, c3, 1, bias=True)
        self.inner2 = nn.Conv2d(c1, c3, 1, bias=True)
        self.out2 = nn.Conv2d(c3, o2, 3, padding=1, bias=False)
        self.out3 = nn.Conv2d(c3, o3, 3, padding=1, bias=False)
        self.out_channels = [c3, c1, c0]
        self._caches = {}
        self._ones = {}             # constant ones maps (the bias channel of the split last head), per (device, h, w)

    # ---- packed weights (BN folded), cached per layer ------------------------------------------------
    def _pk(self, name, conv, bn=None):
        cache = self._caches.setdefault(name, packing.PackCache())
        t = [conv.weight, conv.bias]
        if bn is not None:
            t += [bn.weight, bn.bias, bn.running_mean, bn.running_var]

        def build():
            w, b = conv.weight, conv.bias
            if bn is not None:
                scale, shift = packing.bn_scale_shift(bn)
                w = w * scale.view(-1, 1, 1, 1)
                b = shift if b is None else b * scale + shift
            return packing.pack_conv2d(w, b)

        return cache.get(t, build)

    @staticmethod
    def _is_k3s1(blk):
        return blk.conv.kernel_size == (3, 3) and blk.conv.stride == (1, 1) and blk.conv.padding == (1, 1)

    def _pk_oct(self, name, conv, bn):
        """BatchNorm-folded weights of a one-octet 3x3 layer for ``ops.conv2d_k3_twice``."""
        cache = self._caches.setdefault(name + ".oct", packing.PackCache())
        t = [conv.weight, conv.bias] + ([bn.weight, bn.bias, bn.running_mean, bn.running_var] if bn is not None else [])

        def build():
            w, b = conv.weight, conv.bias
            if bn is not None:
                scale, shift = packing.bn_scale_shift(bn)
                w = w * scale.view(-1, 1, 1, 1)
                b = shift if b is None else b * scale + shift
            return packing.pack_conv2d_bf16x3_oct(w, b)

        return cache.get(t, build)

    def _block(self, name, blk, x):
        """Conv2d wrapper (conv + BN + ReLU) on the HIP path; 3x3 stride 1 or 5x5 stride 2."""
        w, b = self._pk(name, blk.conv, blk.bn)
        act = ops.ACT_RELU if blk.relu else ops.ACT_NONE
        if blk.conv.kernel_size == (5, 5) and blk.conv.stride == (2, 2) and blk.conv.padding == (2, 2):
            return ops.conv2d_k5s2(x, w, b, blk.conv.out_channels, act=act)
        if blk.conv.kernel_size == (3, 3) and blk.conv.stride == (1, 1) and blk.conv.padding == (1, 1):
            return ops.conv2d([x], w, b, blk.conv.out_channels, 3, act=act)
        raise NotImplementedError("feature pyramid: only 3x3/s1/p1 and 5x5/s2/p2 blocks are instantiated on the HIP path")

    def run(self, img):
        """img planar [3,H,W] (H, W multiples of 8, W/2 multiple of 4) -> {stageK: [C,h,w]}."""
        _require_eval(self)
        x = img
        c0 = list(self.conv0)
        if (ops.uses_split() and len(c0) == 2 and all(self._is_k3s1(b) and b.relu for b in c0) and c0[0].conv.in_channels <= 8
                and c0[0].conv.out_channels <= 8 and c0[1].conv.out_channels <= 8 and img.shape[-1] % 4 == 0
                and ops.option("fpn_conv0_fused") != 0):
            # the two full-resolution layers in one kernel: their 8-channel intermediate (61 MB at 1184x1600) stays in LDS
            (w1, b1), (w2, b2) = (self._pk_oct(f"conv0.{i}", b.conv, b.bn) for i, b in enumerate(c0))
            x = ops.conv2d_k3_twice(x, w1, b1, w2, b2, c0[1].conv.out_channels)
        else:
            for i, blk in enumerate(c0):
                x = self._block(f"conv0.{i}", blk, x)
        levels = []
        for lname in ("conv1", "conv2", "conv3"):
            for i, blk in enumerate(getattr(self, lname)):
                x = self._block(f"{lname}.{i}", blk, x)
            levels.append(x)
        l1, l2, top = levels
        cl = self.channels_last_outputs and img.shape[-1] % 32 == 0

        def head(name, conv, x, ks):
            w, b = self._pk(name, conv)
            if cl:      # [h,w,C] in memory, presented as [C,h,w]
                return ops.conv2d([x], w, b, conv.out_channels, ks, epilogue=ops.EPI_NHWC).permute(2, 0, 1)
            return ops.conv2d([x], w, b, conv.out_channels, ks)

        out = {"stage1": head("out1", self.out1, top, 1)}
        w, b = self._pk("inner1", self.inner1)       # lateral 1x1 + nearest-upsampled coarser map, one kernel
        top = ops.conv2d([l2], w, b, self.inner1.out_channels, 1, epilogue=ops.EPI_ADD_UP2, aux0=top)
        out["stage2"] = head("out2", self.out2, top, 3)
        co3 = self.out3.out_channels
        if (ops.uses_split() and self.out3.bias is None and self.out3.kernel_size == (3, 3) and co3 <= 16 and l1.shape[0] % 8 == 0
                and top.shape[0] % 8 == 0 and l1.shape[-1] % 4 == 0 and top.shape[-1] % 4 == 0 and l1.shape[-2] % 2 == 0
                and ops.option("fpn_split_head") != 0):
            # the last head without its 64-channel full-resolution input: the upsampled branch is evaluated at half resolution
            # (4 parity groups of output channels), the lateral branch with out3 o inner2 composed (packing.pack_fpn_head_split)
            t = [self.out3.weight, self.inner2.weight, self.inner2.bias]
            (wu, bu), (wl, bl) = self._caches.setdefault("out3.split", packing.PackCache()).get(
                t, lambda: packing.pack_fpn_head_split(self.out3.weight, self.inner2.weight, self.inner2.bias))
            ones = self._ones.get((top.device, top.shape[-2], top.shape[-1]))
            if ones is None:
                ones = self._ones.setdefault((top.device, top.shape[-2], top.shape[-1]),
                                             torch.ones(1, top.shape[-2], top.shape[-1], device=top.device, dtype=torch.float32))
            u = ops.conv2d_k3_bf16x3([top, ones], wu, bu, 4 * co3)
            o3 = ops.conv2d_k3_bf16x3([l1], wl, bl, co3, epilogue=ops.EPI_NHWC_ADD_SHUF2 if cl else ops.EPI_ADD_SHUF2, aux0=u)
            out["stage3"] = o3.permute(2, 0, 1) if cl else o3
            return out
        w, b = self._pk("inner2", self.inner2)
        top = ops.conv2d([l1], w, b, self.inner2.out_channels, 1, epilogue=ops.EPI_ADD_UP2, aux0=top)
        out["stage3"] = head("out3", self.out3, top, 3)
        return out

    def forward_torch(self, x):
        l1 = self.conv1(self.conv0(x))
        l2 = self.conv2(l1)
        top = self.conv3(l2)
        outputs = {"stage1": self.out1(top)}
        top = F.interpolate(top, scale_factor=2, mode="nearest") + self.inner1(l2)
        outputs["stage2"] = self.out2(top)
        top = F.interpolate(top, scale_factor=2, mode="nearest") + self.inner2(l1)
        outputs["stage3"] = self.out3(top)
        return outputs

    @ops.on_tensor_device

    def forward(self, x):
        if self.training:
            # training: the differentiable HIP operators (forward AND backward); CPU tensors raise in there like everywhere else on
            # the path.  (The stock composite stays available as the explicit method ``forward_torch``: A/B baseline only; note that
            # MIOpen's weight gradient of the context pyramid's 32->32 3x3 layer on a 16x20 map is off by 7.7 % of its peak against
            # the CPU, tools/diag_fpn.py)
            from .. import train_path
            return train_path.feature_pyramid(self, x)
        outs = [self.run(x[i].contiguous()) for i in range(x.shape[0])]
        return {k: _stack([o[k] for o in outs]) for k in outs[0]}


# =============================================================================================
# a4: 3-D U-Net regulariser (reference: models/module.py:435-463)
# =============================================================================================
class CostRegNet_2_sample_FPN3D_Fast(nn.Module):
    def __init__(self, in_channels, base_channels):
        super().__init__()
        b = base_channels
        self.conv0 = Conv3d(in_channels, b, padding=1)
        self.conv1 = Conv3d(b, b, padding=1)
        self.conv2 = Conv3d(b, b * 2, stride=2, padding=1)
        self.conv3 = Conv3d(b * 2, b * 2, padding=1)
        self.conv4 = Conv3d(b * 2, b * 4, stride=2, padding=1)
        self.conv5 = Conv3d(b * 4, b * 4, padding=1)
        self.conv6 = Deconv3d(b * 4, b * 2, stride=2, padding=1, output_padding=1)
        self.conv7 = Deconv3d(b * 2, b, stride=2, padding=1, output_padding=1)
        self.prob = nn.Conv3d(b, 1, 3, stride=1, padding=1, bias=False)
        self._prob_cache = packing.PackCache()

    def run(self, vol):
        """vol planar [cin,D,h,w] -> (prob [1,D,h,w], pro [b,D,h,w]); skip adds are fused into the
        transposed-conv epilogues (added after their ReLU, models/module.py:460-461)."""
        _require_eval(self)
        _, D, h, w = vol.shape
        if D % 4 or h % 4 or w % 4:
            raise ValueError(f"cost volume dims {D}x{h}x{w} must be multiples of 4 (two stride-2 levels)")
        c1 = self.conv1.run([self.conv0.run([vol])])
        c3 = self.conv3.run([self.conv2.run([c1])])
        x = self.conv5.run([self.conv4.run([c3])])
        x = self.conv6.run(x, skip=c3)
        pro = self.conv7.run(x, skip=c1)
        wp, _ = self._prob_cache.get([self.prob.weight], lambda: packing.pack_conv3d(self.prob, None))
        prob = ops.conv3d_k3([pro], wp, None, 1, stride=(1, 1, 1), relu=False)
        return prob, pro

    @ops.on_tensor_device

    def forward(self, x):
        if self.training:
            from .. import train_path
            return train_path.cost_regnet(self, x)
        outs = [self.run(x[i].contiguous()) for i in range(x.shape[0])]
        return _stack([o[0] for o in outs]), _stack([o[1] for o in outs])


# =============================================================================================
# a9: cross-scale propagation block (reference: models/module.py:501-516)
# =============================================================================================
class cost_up_small(nn.Module):
    def __init__(self, in_channels, base_channels, IGEV_cost_channel=1):
        super().__init__()
        self.IGEV_cost_channel = IGEV_cost_channel
        self.conv0 = Conv3d(in_channels, base_channels, stride=(1, 2, 2), padding=1)
        self.conv_cost = Conv3d(self.IGEV_cost_channel, base_channels, padding=1)
        self.conv1 = Conv3d(base_channels * 2, base_channels, padding=1)
        self.conv2 = Deconv3d(base_channels, self.IGEV_cost_channel, stride=(1, 2, 2), padding=1,
                              output_padding=(0, 1, 1))

    def run(self, x, prior):
        """x [cin,D,h,w], prior [1,D,h/2,w/2] -> (conv2 [1,D,h,w], conv1 [b,D,h/2,w/2]); the channel
        concatenation of models/module.py:513 is read in place from the two tensors."""
        a = self.conv0.run([x])
        b = self.conv_cost.run([prior])
        c1 = self.conv1.run([a, b])
        return self.conv2.run(c1), c1

    def _roll_packed(self):
        c = self.conv1
        t = [c.conv.weight, c.conv.bias] + ([c.bn.weight, c.bn.bias, c.bn.running_mean, c.bn.running_var] if c.bn is not None else [])
        return c._cache_roll.get(t, lambda: packing.pack_conv3d_roll_bf16x3(c.conv, c.bn))

    @staticmethod
    def pairable(a, b, x, prior_w):
        """Two blocks can share their launches when they have the stock shape (1 -> 8 -> 8 -> 1 channels, stride (1,2,2)), the
        same input volume shape and the split-precision rolling conv applies to conv1."""
        def stock(m):
            return (m.conv0.conv.in_channels == 1 and m.conv0.out_channels == 8 and _triple(m.conv0.conv.stride) == (1, 2, 2)
                    and m.conv_cost.conv.in_channels == 1 and m.conv_cost.out_channels == 8
                    and _triple(m.conv_cost.conv.stride) == (1, 1, 1) and m.conv1.out_channels == 8
                    and _triple(m.conv1.conv.stride) == (1, 1, 1) and m.conv2.out_channels == 1
                    and _triple(m.conv2.conv.stride) == (1, 2, 2) and m.conv0.relu and m.conv_cost.relu and m.conv1.relu
                    and m.conv2.relu and not m.training)
        if ops.option("csp_pair") == 0:        # A/B switch: every layer of the two blocks as its own launch
            return False
        return stock(a) and stock(b) and ops.uses_split() and x.shape[0] == 1 and prior_w % 4 == 0

    @staticmethod
    def run_pair(a, b, x, prior_a, prior_b):
        """``a.run(x, prior_a)`` and ``b.run(x, prior_b)`` with every layer of the two blocks in ONE launch (the blocks are
        independent: CSP_R[s] and CSP_C[s] of a stage, reference models/Effi_MVS_plus.py:520-531) -> ((out_a, c1_a), (out_b, c1_b))."""
        (w0a, b0a), (w0b, b0b) = a.conv0._packed(), b.conv0._packed()
        (wca, bca), (wcb, bcb) = a.conv_cost._packed(), b.conv_cost._packed()
        (w1a, b1a), (w1b, b1b) = a._roll_packed(), b._roll_packed()
        if ops.option("csp_gen") and x.shape[1] <= 14 and ops.get_option("roll_rp") != 0:
            # conv0 | conv_cost generated inside conv1's rolling window (csrc/conv2d.hip: csp_gen_roll_rp_kernel): one launch, bitwise
            c1a, c1b = ops.csp_gen_roll_pair(x, prior_a, w0a, b0a, wca, bca, w1a, b1a, prior_b, w0b, b0b, wcb, bcb, w1b, b1b)
        else:
            fa, fb = ops.conv3d_k3_pair(x, w0a, b0a, x, w0b, b0b, 8, sxy=2, relu=True)
            ga, gb = ops.conv3d_k3_pair(prior_a, wca, bca, prior_b, wcb, bcb, 8, sxy=1, relu=True)
            c1a, c1b = ops.conv3d_k3s1_roll_pair([fa, ga], w1a, b1a, [fb, gb], w1b, b1b, 8, relu=True)
        (w2a, b2a), (w2b, b2b) = a.conv2._packed(), b.conv2._packed()
        oa, ob = ops.deconv3d_k3_pair(c1a, w2a, b2a, c1b, w2b, b2b, 1, sz=1, relu=True)
        return (oa, c1a), (ob, c1b)

    @ops.on_tensor_device

    def forward(self, x, IGEV_cost):
        if self.training:
            from .. import train_path
            return train_path.cost_up_small(self, x, IGEV_cost)
        outs = [self.run(x[i].contiguous(), IGEV_cost[i].contiguous()) for i in range(x.shape[0])]
        return _stack([o[0] for o in outs]), _stack([o[1] for o in outs])


# =============================================================================================
# a1: homography warp (reference: models/module.py:303-344)
# =============================================================================================
@ops.on_tensor_device
def homo_warping_new(src_fea, src_proj, ref_proj, depth_values):
    """src_fea [B,C,H,W]; src_proj/ref_proj [B,4,4]; depth_values [B,D] or [B,D,H,W] -> [B,C,D*H,W].

    Materialises the warped volume for API parity; the fused cost-volume kernels never do.
    """
    B, C_, H, W = src_fea.shape
    D = depth_values.shape[1]
    outs = []
    if src_fea.requires_grad and torch.is_grad_enabled():
        # differentiable form (scope row n2): forward and backward on the HIP kernels, gradient to src_fea only (the reference
        # builds its grid under no_grad, module.py:313)
        from .. import autograd as _ag
        for b in range(B):
            outs.append(_ag.homo_warp(src_fea[b], src_proj[b], ref_proj[b], depth_values[b]).reshape(C_, D * H, W))
        return torch.stack(outs)
    for b in range(B):
        nhwc = ops.to_nhwc([src_fea[b]])[0]
        rt = ops.rel_proj(src_proj[b].contiguous(), ref_proj[b].contiguous())
        dv = depth_values[b]
        outs.append(ops.homo_warp(nhwc, rt, dv if dv.dim() == 1 else dv, D).view(C_, D * H, W))
    return _stack(outs)


# =============================================================================================
# a6/a7: small tensor helpers kept for API parity (plain tensor algebra, device agnostic)
# =============================================================================================
def depth_regression(p, depth_values):
    """sum_D p * depth_values (reference: models/module.py:518-524)."""
    if depth_values.dim() <= 2:
        depth_values = depth_values.view(*depth_values.shape, 1, 1)
    return torch.sum(p * depth_values, 1)


def get_cur_depth_range_samples(cur_depth, ndepth, depth_inteval_pixel, shape, max_depth=192.0, min_depth=0.0):
    """Per-pixel hypotheses around cur_depth (reference: models/module.py:554-570).  The fused stage-2/3
    kernel generates these in registers; this tensor form exists for callers that want the samples."""
    half = ndepth // 2 * depth_inteval_pixel
    lo = (cur_depth - half).clamp(min=1e-4)
    hi = (cur_depth + half).clamp(min=1e-4, max=1e4)
    assert cur_depth.shape == torch.Size(shape), "cur_depth:{}, input shape:{}".format(cur_depth.shape, shape)
    step = (hi - lo) / (ndepth - 1)
    idx = torch.arange(0, ndepth, device=cur_depth.device, dtype=cur_depth.dtype).reshape(1, -1, 1, 1)
    return (lo.unsqueeze(1) + idx * step.unsqueeze(1)).clamp(min=1e-5)


def get_depth_range_samples(cur_depth, ndepth, depth_inteval_pixel, device, dtype, shape, max_depth=192.0,
                            min_depth=0.0):
    """(B,D) input: uniform samples between first and last entry, repeated over the map;
    (B,H,W) input: per-pixel samples (reference: models/module.py:572-591)."""
    if cur_depth.dim() == 2:
        lo, hi = cur_depth[:, 0], cur_depth[:, -1]
        step = (hi - lo) / (ndepth - 1)
        idx = torch.arange(0, ndepth, device=device, dtype=dtype).reshape(1, -1)
        s = lo.unsqueeze(1) + idx * step.unsqueeze(1)
        return s.unsqueeze(-1).unsqueeze(-1).repeat(1, 1, shape[1], shape[2])
    return get_cur_depth_range_samples(cur_depth, ndepth, depth_inteval_pixel, shape, max_depth, min_depth)


def mvs_loss_static(inputs, depth_gt_ms, mask_ms, dloss, depth_values=[425, 935], loss_rate=0.9):
    """``mvs_loss`` without data-dependent shapes: the mean over the valid pixels as sum(where(valid, loss, 0)) / count instead of
    boolean indexing (which synchronises with the host and cannot be captured into a graph).  Same value up to the order of the
    additions; an Inf / NaN estimate at a MASKED pixel is dropped as the reference's indexing drops it (``where`` selects, a
    product with a 0 / 1 mask would turn it into NaN), forward and backward."""
    total = torch.zeros((), dtype=torch.float32, device=mask_ms["stage1"].device)
    per_output = {}
    n = len(inputs)
    for i, est in enumerate(inputs):
        key = "stage{}".format(dloss[i])
        valid = mask_ms[key] > 0.5
        diff = torch.where(valid, est - depth_gt_ms[key], torch.zeros((), dtype=est.dtype, device=est.device))   # masked pixels: zero loss
        li = F.smooth_l1_loss(diff, torch.zeros_like(diff), reduction="none").sum() / valid.sum().to(est.dtype)     # AND zero gradient
        per_output["l{}".format(i)] = li
        total = total + (1.0 if i == 0 else loss_rate ** (n - i - 1)) * li
    return total, per_output


def mvs_loss(inputs, depth_gt_ms, mask_ms, dloss, depth_values=[425, 935], loss_rate=0.9):
    """Smooth-L1 over the cascade's outputs with geometric weights (reference: models/module.py:526-552).
    Training-side consumer of the path's outputs; plain tensor algebra."""
    total = torch.tensor(0.0, dtype=torch.float32, device=mask_ms["stage1"].device, requires_grad=False)
    per_output = {}
    n = len(inputs)
    for i, est in enumerate(inputs):
        key = "stage{}".format(dloss[i])
        valid = mask_ms[key] > 0.5
        li = F.smooth_l1_loss(est[valid], depth_gt_ms[key][valid], reduction="mean")
        per_output["l{}".format(i)] = li
        total = total + (1.0 if i == 0 else loss_rate ** (n - i - 1)) * li
    return total, per_output
