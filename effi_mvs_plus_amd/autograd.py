"""Scope row n2: differentiable forms of the path's operators -- forward AND backward on the HIP kernels.

Every class below is a ``torch.autograd.Function`` whose ``forward`` calls the same C-ABI entry the inference path uses and
whose ``backward`` calls the training kernels of ``csrc/train_ops.hip`` / ``csrc/warpcorr.hip`` (weight gradients, BatchNorm on
batch statistics, activation / gating derivatives, lookup / soft-argmin / upsampling / warp backward).  Input gradients of the
convolutions reuse the forward convolution kernels with re-arranged weights:

    conv, stride 1          dx = conv(g, W flipped, in/out swapped)
    conv, stride s          dx = transposed_conv(g, W)                       (the reference's own Deconv3d kernel shape)
    transposed conv         dx = conv(g, W with in/out swapped, stride s)

PyTorch contributes the autograd graph, gradient accumulation and tensor views; no arithmetic of the path runs in stock
operators.  All products are exact fp32 here (the fp32 matrix-core / vector kernels), independent of the inference-time
``ops.set_precision``: several gradients of this network are ill-conditioned (tests/test_gpu_train.py), so the training path does
not add the 2^-17 product error of the split-bf16 convolutions on top.  What the reference does at each place is cited per function (paths relative to the reference repository).
Batched tensors [B, ...] in and out; kernels are launched per sample (BatchNorm statistics span the batch).
"""
import torch

from . import ops, packing

def _stack(ts):
    """Per-sample results -> [B, ...]: a view for B = 1 (DTU training runs one sample per GPU, train.py:37), a copy otherwise."""
    return ts[0].unsqueeze(0) if len(ts) == 1 else torch.stack(ts)


def _c(t_):
    return t_.contiguous()


class _GradArena:
    """Zero-initialised storage for the weight gradients of one backward pass: the weight-gradient kernels ACCUMULATE (one atomic
    per weight and workgroup), so every gradient starts from zeros -- ~300 ``zeros_like`` launches per step as separate tensors.
    Here they are slices of large blocks that one fill each clears: a slice is handed out once, the rest of a block is still zero
    for the next step's gradients, and an exhausted block lives as long as a gradient references it."""
    BLOCK = 1 << 21          # floats per block (8 MB)

    def __init__(self):
        self.blocks = {}     # device -> [tensor, used]

    def zeros_like(self, w):
        n = w.numel()
        if w.dtype != torch.float32 or n > self.BLOCK // 4:
            return torch.zeros_like(w)
        blk = self.blocks.get(w.device)
        cap = torch.cuda.is_current_stream_capturing()
        # inside a graph capture the block's fill must be part of the graph (a replay accumulates into the same addresses again)
        if blk is None or blk[1] + n > blk[0].numel() or blk[2] != cap:
            blk = [torch.zeros(self.BLOCK, device=w.device, dtype=torch.float32), 0, cap]
            self.blocks[w.device] = blk
        out = blk[0][blk[1]:blk[1] + n].view(w.shape)
        blk[1] += (n + 63) & ~63
        return out


grad_arena = _GradArena()


# =============================================================================================
# convolutions
# =============================================================================================
class _Conv2d(torch.autograd.Function):
    """nn.Conv2d (k 1 / 3 / 7, stride 1, padding k//2, optional bias) + activation over the channel concatenation of ``xs``
    (models/update.py:14-15,36-38,73-81,109-112; models/module.py:213-220)."""

    @staticmethod
    def forward(ctx, weight, bias, act, *xs):
        cout, cin, ks = weight.shape[0], weight.shape[1], weight.shape[-1]
        xs = [_c(x) for x in xs]
        B = xs[0].shape[0]
        if sum(x.shape[1] for x in xs) != cin:
            raise ValueError("conv2d: input channels do not match the weight")
        with torch.no_grad():
            if ks == 7:
                if cin != 1 or act != ops.ACT_RELU or len(xs) != 1:
                    raise NotImplementedError("7x7 convolution: single-channel input followed by ReLU (ProjectionInput.convd1)")
                w7, b7 = packing.pack_conv2d_c1k7(weight, bias)
                y = _stack([ops.conv2d_c1k7_relu(xs[0][b], w7, b7, cout, exact=True) for b in range(B)])
            else:
                wp, bp = ops.pack_conv2d_mfma_dev(weight, bias)       # exact fp32 products in training, whatever ops.get_precision() says
                y = _stack([ops.conv2d([x[b] for x in xs], wp, bp, cout, ks, act=act) for b in range(B)])
        ctx.save_for_backward(weight, y, *xs)
        ctx.act, ctx.has_bias = act, bias is not None
        return y

    @staticmethod
    def backward(ctx, gy):
        weight, y, *xs = ctx.saved_tensors
        cout, cin, ks = weight.shape[0], weight.shape[1], weight.shape[-1]
        B = y.shape[0]
        g = _c(gy)
        if ctx.act != ops.ACT_NONE:
            g = ops.pointwise(ops.PW_ACT_BWD[ctx.act], g, y)
        gb = ops.channel_sum(g) if ctx.has_bias else None
        gw = grad_arena.zeros_like(weight) if ctx.needs_input_grad[0] else None
        if gw is not None:
            dw = gw.view(cout, cin, ks * ks)
            for b in range(B):
                off = 0
                for x in xs:
                    ops.conv_wgrad(g[b], x[b], dw, off, 1, ks)
                    off += x.shape[1]
        gxs = [None] * len(xs)
        if any(ctx.needs_input_grad[3:]):
            if ks == 7:
                raise NotImplementedError("7x7 convolution: no input gradient (its input is the detached inverse depth, update.py:121)")
            wp, bp = ops.pack_conv2d_mfma_dev(weight, None, dgrad=True)
            gcat = _stack([ops.conv2d([g[b]], wp, bp, cin, ks) for b in range(B)])
            off = 0
            for i, x in enumerate(xs):
                if ctx.needs_input_grad[3 + i]:
                    gxs[i] = gcat[:, off:off + x.shape[1]]
                off += x.shape[1]
        return (gw, gb, None, *gxs)


def conv2d(xs, weight, bias=None, act=ops.ACT_NONE):
    return _Conv2d.apply(weight, bias, act, *xs)


class _Conv2dK5S2(torch.autograd.Function):
    """nn.Conv2d 5x5 / stride 2 / padding 2 / no bias of the feature pyramid's down-sampling layers (models/module.py:376-388;
    BatchNorm + ReLU follow separately)."""

    @staticmethod
    def forward(ctx, weight, x):
        x = _c(x)
        cout = weight.shape[0]
        with torch.no_grad():
            wp, bp = ops.pack_conv2d_mfma_dev(weight, None)
            y = _stack([ops.conv2d_k5s2(x[b], wp, bp, cout, act=ops.ACT_NONE) for b in range(x.shape[0])])
        ctx.save_for_backward(weight, x)
        return y

    @staticmethod
    def backward(ctx, gy):
        weight, x = ctx.saved_tensors
        g = _c(gy)
        B, cin, hin, win = x.shape
        gw = gx = None
        if ctx.needs_input_grad[0]:
            gw = grad_arena.zeros_like(weight)
            dw = gw.view(weight.shape[0], cin, 25)
            for b in range(B):
                ops.conv_wgrad(g[b], x[b], dw, 0, 1, 5, stride=(1, 2))
        if ctx.needs_input_grad[1]:
            wc = _c(weight)
            gx = _stack([ops.conv2d_k5s2_dgrad_mfma(g[b], wc, hin, win) for b in range(B)])
        return gw, gx


def conv2d_k5s2(x, weight):
    return _Conv2dK5S2.apply(weight, x)


def _stride3(stride):
    s = tuple(stride) if isinstance(stride, (tuple, list)) else (stride,) * 3
    if s not in ((1, 1, 1), (2, 2, 2), (1, 2, 2)):
        raise NotImplementedError(f"3-D convolution stride {s}: the path uses (1,1,1), (2,2,2), (1,2,2)")
    return s


class _Conv3d(torch.autograd.Function):
    """nn.Conv3d k3 / p1 / no bias over the channel concatenation of ``xs`` (models/module.py:124-166; BatchNorm is separate)."""

    @staticmethod
    def forward(ctx, weight, stride, *xs):
        cout, cin = weight.shape[0], weight.shape[1]
        s = _stride3(stride)
        xs = [_c(x) for x in xs]
        B = xs[0].shape[0]
        with torch.no_grad():
            wp = _c(weight.permute(1, 2, 3, 4, 0).reshape(cin, 27, cout))
            y = _stack([ops.conv3d_k3([x[b] for x in xs], wp, None, cout, stride=s, relu=False) for b in range(B)])
        ctx.save_for_backward(weight, *xs)
        ctx.stride = s
        return y

    @staticmethod
    def backward(ctx, gy):
        weight, *xs = ctx.saved_tensors
        cout, cin = weight.shape[0], weight.shape[1]
        s = ctx.stride
        g = _c(gy)
        B = g.shape[0]
        gw = None
        if ctx.needs_input_grad[0]:
            gw = grad_arena.zeros_like(weight)
            dw = gw.view(cout, cin, 27)
            for b in range(B):
                off = 0
                for x in xs:
                    ops.conv_wgrad(g[b], x[b], dw, off, 3, 3, stride=(s[0], s[1]))
                    off += x.shape[1]
        gxs = [None] * len(xs)
        if any(ctx.needs_input_grad[2:]):
            D, h, w = xs[0].shape[2:]
            if s == (1, 1, 1):
                wd = _c(weight.flip(2, 3, 4).permute(0, 2, 3, 4, 1).reshape(cout, 27, cin))
                gcat = _stack([ops.conv3d_k3([g[b]], wd, None, cin, stride=s, relu=False) for b in range(B)])
            else:
                if (D % s[0]) or (h % 2) or (w % 2):
                    raise NotImplementedError("strided 3-D convolution backward needs even input sizes along the strided axes")
                wt = _c(weight.permute(0, 2, 3, 4, 1).reshape(cout, 27, cin))
                gcat = _stack([ops.deconv3d_k3(g[b], wt, None, cin, sz=s[0], relu=False) for b in range(B)])
            off = 0
            for i, x in enumerate(xs):
                if ctx.needs_input_grad[2 + i]:
                    gxs[i] = gcat[:, off:off + x.shape[1]]
                off += x.shape[1]
        return (gw, None, *gxs)


def conv3d(xs, weight, stride=1):
    return _Conv3d.apply(weight, stride, *xs)


class _Deconv3d(torch.autograd.Function):
    """nn.ConvTranspose3d k3 / p1 / stride (s,2,2) / output_padding (s-1,1,1) / no bias (models/module.py:168-209)."""

    @staticmethod
    def forward(ctx, weight, stride, x):
        cin, cout = weight.shape[0], weight.shape[1]
        s = _stride3(stride)
        if s == (1, 1, 1):
            raise NotImplementedError("transposed 3-D convolution: stride (2,2,2) or (1,2,2)")
        x = _c(x)
        with torch.no_grad():
            wp = _c(weight.permute(0, 2, 3, 4, 1).reshape(cin, 27, cout))
            y = _stack([ops.deconv3d_k3(x[b], wp, None, cout, sz=s[0], relu=False) for b in range(x.shape[0])])
        ctx.save_for_backward(weight, x)
        ctx.stride = s
        return y

    @staticmethod
    def backward(ctx, gy):
        weight, x = ctx.saved_tensors
        cin, cout = weight.shape[0], weight.shape[1]
        s = ctx.stride
        g = _c(gy)
        B = g.shape[0]
        gw = gx = None
        if ctx.needs_input_grad[0]:
            gw = grad_arena.zeros_like(weight)
            dw = gw.view(cin, cout, 27)
            for b in range(B):
                ops.conv_wgrad(x[b], g[b], dw, 0, 3, 3, stride=(s[0], s[1]))
        if ctx.needs_input_grad[2]:
            wc = _c(weight.permute(1, 2, 3, 4, 0).reshape(cout, 27, cin))
            gx = _stack([ops.conv3d_k3([g[b]], wc, None, cin, stride=s, relu=False) for b in range(B)])
        return gw, None, gx


def deconv3d(x, weight, stride):
    return _Deconv3d.apply(weight, stride, x)


# =============================================================================================
# BatchNorm on batch statistics (+ the ReLU that follows it everywhere on the path)
# =============================================================================================
class _BatchNormTrain(torch.autograd.Function):
    """nn.BatchNorm2d / 3d in training mode (models/module.py:148-157,191-200,217-220): batch statistics, running statistics
    updated in place with ``momentum`` (unbiased variance), optional fused ReLU."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, momentum, eps, relu, n_tracked=None):
        x = _c(x)
        with torch.no_grad():
            y, mean, invstd = ops.bn_train_fwd(x, _c(gamma), _c(beta), eps, momentum, running_mean, running_var, n_tracked, relu)
        ctx.save_for_backward(x, y, mean, invstd, gamma)
        ctx.relu = relu
        return y

    @staticmethod
    def backward(ctx, gy):
        x, y, mean, invstd, gamma = ctx.saved_tensors
        gx, s1, s2 = ops.bn_bwd(_c(gy), y, x, mean, invstd, _c(gamma), ctx.relu)
        return gx, s2, s1, None, None, None, None, None, None


def batch_norm_train(x, bn, relu):
    """``bn``: nn.BatchNorm2d / 3d module (its running statistics and counter are updated like nn.BatchNorm does)."""
    n_tracked = bn.num_batches_tracked            # incremented by the forward kernel (nn.BatchNorm counts before it uses the value)
    if bn.momentum is None:                 # nn.BatchNorm: cumulative moving average, factor 1 / num_batches_tracked
        if n_tracked is None:
            raise NotImplementedError("batch_norm_train: momentum=None needs track_running_stats (the counter)")
        n_tracked.add_(1)
        momentum, n_tracked = 1.0 / float(n_tracked.item()), None
    else:
        momentum = bn.momentum
    if n_tracked is not None and (n_tracked.device != x.device or bn.running_mean is None):
        n_tracked.add_(1)
        n_tracked = None
    return _BatchNormTrain.apply(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, momentum, bn.eps, relu, n_tracked)


# =============================================================================================
# element-wise pieces
# =============================================================================================
class _Act(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, act):
        y = ops.pointwise({ops.ACT_TANH: ops.PW_TANH, ops.ACT_RELU: ops.PW_RELU, ops.ACT_SIGMOID: ops.PW_SIGMOID}[act], _c(x))
        ctx.save_for_backward(y)
        ctx.act = act
        return y

    @staticmethod
    def backward(ctx, g):
        (y,) = ctx.saved_tensors
        return ops.pointwise(ops.PW_ACT_BWD[ctx.act], _c(g), y), None


def activation(x, act):
    return _Act.apply(x, act)


class _Mul(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, y):
        x, y = _c(x), _c(y)
        ctx.save_for_backward(x, y)
        return ops.pointwise(ops.PW_MUL, x, y)

    @staticmethod
    def backward(ctx, g):
        x, y = ctx.saved_tensors
        gx, gy = ops.pointwise(ops.PW_MUL_BWD, _c(g), x, y, n_out=2)
        return gx, gy


class _SplitChannels(torch.autograd.Function):
    """[B,C,...] -> ([B,:k,...], [B,k:,...]) as views; the backward writes the two gradients side by side (one launch; autograd's own
    slice backward would zero-fill and copy a full-size tensor per half and add the two)."""

    @staticmethod
    def forward(ctx, x, k):
        ctx.k, ctx.C = k, x.shape[1]
        return x[:, :k], x[:, k:]

    @staticmethod
    def backward(ctx, ga, gb):
        if ga is None or gb is None:
            ref = ga if ga is not None else gb
            shape = list(ref.shape)
            shape[1] = ctx.k if ga is None else ctx.C - ctx.k
            z = torch.zeros(shape, device=ref.device, dtype=ref.dtype)
            ga, gb = (z, gb) if ga is None else (ga, z)
        return torch.cat([ga, gb], dim=1), None


def split_channels(x, k):
    return _SplitChannels.apply(x, k)


class _GruCombine(torch.autograd.Function):
    """h' = (1 - z) h + z q  (models/update.py:48)."""

    @staticmethod
    def forward(ctx, z, h, q):
        z, h, q = _c(z), _c(h), _c(q)
        ctx.save_for_backward(z, h, q)
        return ops.pointwise(ops.PW_GRU, z, h, q)

    @staticmethod
    def backward(ctx, g):
        z, h, q = ctx.saved_tensors
        gz, gh, gq = ops.pointwise(ops.PW_GRU_BWD, _c(g), z, h, q, n_out=3)
        return gz, gh, gq


class _InvToDepth(torch.autograd.Function):
    """scale_inv_depth(inv)[1] with the global range (models/Effi_MVS_plus.py:138-148,423): lo / hi = first / last inverse depth."""

    @staticmethod
    def forward(ctx, inv, lo, hi):
        inv = _c(inv)
        ctx.save_for_backward(inv)
        ctx.range = (lo, hi)
        return ops.pointwise(ops.PW_INV_TO_DEPTH, inv, s0=lo, s1=hi)

    @staticmethod
    def backward(ctx, g):
        (inv,) = ctx.saved_tensors
        return ops.pointwise(ops.PW_INV_TO_DEPTH_BWD, _c(g), inv, s0=ctx.range[0], s1=ctx.range[1]), None, None


def inv_to_depth(inv, lo, hi):
    return _InvToDepth.apply(inv, float(lo), float(hi))


class _ScaleChannels(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, factors):
        x = _c(x)
        ctx.save_for_backward(factors)
        inner = x.numel() // factors.numel()
        ctx.inner = inner
        return ops.pointwise(ops.PW_SCALE_CH, x, factors, inner=inner, C=factors.numel())

    @staticmethod
    def backward(ctx, g):
        (factors,) = ctx.saved_tensors
        return ops.pointwise(ops.PW_SCALE_CH, _c(g), factors, inner=ctx.inner, C=factors.numel()), None


def dropout2d(x, p, factors=None):
    """nn.Dropout2d(p) in training mode (models/update.py:22-23,97-98): one Bernoulli(1 - p) draw per (sample, channel), scaled by
    1 / (1 - p).  The draws come from torch's generator of ``x``'s device (or ``factors`` [B*C] when given: tests)."""
    if p <= 0.0 and factors is None:
        return x
    if factors is None:
        factors = torch.empty(x.shape[0] * x.shape[1], device=x.device, dtype=torch.float32).bernoulli_(1 - p).div_(1 - p)
    return _ScaleChannels.apply(x, factors)


# =============================================================================================
# volume operators
# =============================================================================================
class _VolLookup(torch.autograd.Function):
    """pro_bilinear_sampler (models/Effi_MVS_plus.py:118-134) on planar volumes [B,Dp,h,w]; queries [B,nq,qh,qw] (qh, qw = h, w or
    2h, 2w read nearest-downsampled) come from detached depths: gradient to the volume only."""

    @staticmethod
    def forward(ctx, vol, query, dmin, dmax):
        vol, query = _c(vol), _c(query)
        B, Dp, h, w = vol.shape
        pick = lambda r, b: r[b] if r.shape[0] == B else r[0]          # noqa: E731
        y = _stack([ops.vol_lookup1d(vol[b], query[b], pick(dmin, b), pick(dmax, b), h, w) for b in range(B)])
        ctx.save_for_backward(query, dmin, dmax)
        ctx.dims = (B, Dp, h, w)
        return y

    @staticmethod
    def backward(ctx, g):
        query, dmin, dmax = ctx.saved_tensors
        B, Dp, h, w = ctx.dims
        g = _c(g)
        pick = lambda r, b: r[b] if r.shape[0] == B else r[0]          # noqa: E731
        gv = _stack([ops.vol_lookup1d_bwd(g[b], Dp, query[b], pick(dmin, b), pick(dmax, b), h, w) for b in range(B)])
        return gv, None, None, None


def vol_lookup(vol, query, dmin, dmax):
    return _VolLookup.apply(vol, query.detach(), dmin.detach(), dmax.detach())


class _GetCost(torch.autograd.Function):
    """GetCost.forward (models/Effi_MVS_plus.py:257-303) behind scale_inv_depth: cur / reg volumes [B,D,h,w], normalised inverse depth
    [B,1,h,w] (detached, update.py:121) -> cost [B,2*nq,h,w]."""

    @staticmethod
    def forward(ctx, cur, reg, inv_depth, disp_range, interval, dmin, dmax, nq, is_depth):
        cur, reg, inv_depth = _c(cur), _c(reg), _c(inv_depth)
        B, Dc, h, w = cur.shape
        pick = lambda r, b: r[b] if r.shape[0] == B else r[0]          # noqa: E731
        y = _stack([ops.getcost(inv_depth[b], None if is_depth else disp_range[b], interval[b].reshape(1), cur[b], reg[b], pick(dmin, b),
                                pick(dmax, b), nq, h, w, input_is_depth=is_depth) for b in range(B)])
        ctx.save_for_backward(inv_depth, disp_range, interval, dmin, dmax)
        ctx.dims = (B, Dc, reg.shape[1], h, w, nq, is_depth)
        return y

    @staticmethod
    def backward(ctx, g):
        inv_depth, disp_range, interval, dmin, dmax = ctx.saved_tensors
        B, Dc, Dr, h, w, nq, is_depth = ctx.dims
        g = _c(g)
        pick = lambda r, b: r[b] if r.shape[0] == B else r[0]          # noqa: E731
        res = [ops.getcost_bwd(g[b], inv_depth[b], None if is_depth else disp_range[b], interval[b].reshape(1), Dc, Dr, pick(dmin, b),
                               pick(dmax, b), nq, h, w, input_is_depth=is_depth) for b in range(B)]
        return _stack([r[0] for r in res]), _stack([r[1] for r in res]), None, None, None, None, None, None, None


def getcost(cur, reg, inv_depth, disp_range, interval, dmin, dmax, nq, input_is_depth=False):
    """``inv_depth``: normalised inverse depth (with ``disp_range`` [B,n]) or, with ``input_is_depth``, the depth itself."""
    if disp_range is None:
        disp_range = torch.zeros(cur.shape[0], 2, device=cur.device)
    return _GetCost.apply(cur, reg, inv_depth.detach(), _c(disp_range), _c(interval.reshape(-1, 1)), _c(dmin.detach()), _c(dmax.detach()), nq,
                          bool(input_is_depth))


class _SoftArgmin(torch.autograd.Function):
    """softmax over D + depth regression + 4-window confidence (models/Effi_MVS_plus.py:79-88): logits [B,D,h,w], hypotheses [B,D] or
    [B,D,h,w] -> (depth [B,h,w], confidence [B,h,w]); the confidence is computed under no_grad in the reference as well."""

    @staticmethod
    def forward(ctx, logits, hyp):
        logits = _c(logits)
        res = [ops.softmax_regress_conf(logits[b], hyp[b]) for b in range(logits.shape[0])]
        ctx.save_for_backward(logits, hyp)
        depth, conf = _stack([r[0] for r in res]), _stack([r[1] for r in res])
        ctx.mark_non_differentiable(conf)
        return depth, conf

    @staticmethod
    def backward(ctx, gdepth, _gconf):
        logits, hyp = ctx.saved_tensors
        g = _c(gdepth)
        return _stack([ops.softargmin_bwd(logits[b], hyp[b], g[b]) for b in range(logits.shape[0])]), None


def soft_argmin(logits, hyp):
    return _SoftArgmin.apply(logits, hyp.detach())


class _ViewAggregate(torch.autograd.Function):
    """sum_v sim_v w_v / (sum_v w_v + 1e-6)  (models/Effi_MVS_plus.py:48-53,67): sim_views [B,S,D,h,w], weights [B,S,h,w]."""

    @staticmethod
    def forward(ctx, sim_views, weights):
        sim_views, weights = _c(sim_views), _c(weights)
        ctx.save_for_backward(sim_views, weights)
        return _stack([ops.view_aggregate(sim_views[b], weights[b]) for b in range(sim_views.shape[0])])

    @staticmethod
    def backward(ctx, g):
        sim_views, weights = ctx.saved_tensors
        g = _c(g)
        res = [ops.view_aggregate_bwd(sim_views[b], weights[b], g[b]) for b in range(g.shape[0])]
        return _stack([r[0] for r in res]), _stack([r[1] for r in res])


def view_aggregate(sim_views, weights):
    return _ViewAggregate.apply(sim_views, weights)


class _ConvexUpsample(torch.autograd.Function):
    """upsample_depth, ratio 2 (models/Effi_MVS_plus.py:167-178): inv [B,1,h,w], mask [B,36,h,w] -> [B,2h,2w]."""

    @staticmethod
    def forward(ctx, inv, mask):
        inv, mask = _c(inv), _c(mask)
        ctx.save_for_backward(inv, mask)
        return _stack([ops.convex_upsample2x(inv[b], mask[b])[0] for b in range(inv.shape[0])])

    @staticmethod
    def backward(ctx, g):
        inv, mask = ctx.saved_tensors
        g = _c(g)
        res = [ops.convex_upsample2x_bwd(inv[b], mask[b], g[b]) for b in range(g.shape[0])]
        return _stack([r[1] for r in res]), _stack([r[0] for r in res])


def convex_upsample(inv, mask):
    return _ConvexUpsample.apply(inv, mask)


# =============================================================================================
# warps
# =============================================================================================
class _WarpCorrelate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ref_fea, pairs, depth_values, *src_feas):
        # ref_fea / src_feas planar [C,h,w]; pairs [N,2,4,4]; depth_values [D] or [D,h,w]
        D = depth_values.shape[0]
        nhwc = ops.to_nhwc([ref_fea.contiguous()] + [s.contiguous() for s in src_feas])
        rt = ops.compose_rel_proj(pairs.contiguous())
        sim, ent = ops.warpcorr_views(nhwc[0], nhwc[1:], rt, depth_values, D)
        ctx.save_for_backward(*nhwc, rt, depth_values)
        ctx.D = D
        ctx.mark_non_differentiable(ent)
        return sim, ent

    @staticmethod
    def backward(ctx, grad_sim, _grad_ent):
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


class _WarpCorrDyn(torch.autograd.Function):
    """GetCost_initvolume.forward for one sample (models/Effi_MVS_plus.py:184-251): ref / sources planar [C,h,w], view weights
    [S,h>>k,w>>k], current depth [h,w] (detached, :495) -> (similarity [D,h,w], hypotheses [D,h,w]).  Gradients reach the feature
    maps and the view weights."""

    @staticmethod
    def forward(ctx, ref_fea, view_w, pairs, cur_depth, interval, D, *src_feas):
        nhwc = ops.to_nhwc([ref_fea.contiguous()] + [s.contiguous() for s in src_feas])
        rt = ops.compose_rel_proj(pairs.contiguous())
        view_w, cur_depth, interval = _c(view_w), _c(cur_depth), _c(interval.reshape(1))
        sim, samples = ops.warpcorr_dyn(nhwc[0], nhwc[1:], rt, cur_depth, interval, view_w, D)
        ctx.save_for_backward(*nhwc, rt, cur_depth, interval, view_w, sim)
        ctx.D = D
        ctx.mark_non_differentiable(samples)
        return sim, samples

    @staticmethod
    def backward(ctx, grad_sim, _grad_samples):
        *nhwc, rt, cur_depth, interval, view_w, sim = ctx.saved_tensors
        g_ref, g_src, g_vw = ops.warpcorr_dyn_bwd(nhwc[0], nhwc[1:], rt, cur_depth, interval, view_w, ctx.D, sim, _c(grad_sim))
        planar = lambda t: t.permute(2, 0, 1).contiguous()  # noqa: E731
        return (planar(g_ref), g_vw, None, None, None, None) + tuple(planar(g) for g in g_src)


def homo_warp(src_fea, src_proj, ref_proj, depth_values):
    """Differentiable ``homo_warping_new`` for one sample (models/module.py:303-344): src_fea [C,h,w] -> warped [C,D,h,w]; the
    gradient flows to ``src_fea`` (the grid is constant, module.py:313)."""
    return _HomoWarp.apply(src_fea, src_proj, ref_proj, depth_values)


def warp_correlate(ref_fea, src_feas, pairs, depth_values, with_entropy=False):
    """ref_fea [C,h,w], src_feas list of [C,h,w] (C in 8/16/32), pairs [N,2,4,4] (view 0 = reference), depth_values [D] or
    [D,h,w] -> similarity [S,D,h,w] (and the softmax entropy [S,h,w] of it, which the reference computes on the DETACHED
    similarity, models/Effi_MVS_plus.py:43); differentiable w.r.t. the feature maps."""
    sim, ent = _WarpCorrelate.apply(ref_fea, pairs, depth_values, *src_feas)
    return (sim, ent) if with_entropy else sim


def warp_correlate_dyn(ref_fea, src_feas, view_w, pairs, cur_depth, interval, D):
    return _WarpCorrDyn.apply(ref_fea, view_w, pairs, cur_depth.detach(), interval, D, *src_feas)
