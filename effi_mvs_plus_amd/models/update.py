"""Drop-in mirror of the reference's ``models/update.py`` (GRU update block) on the HIP path.

Class names, constructor / forward signatures, return structures and state-dict keys follow
``models/update.py`` of bdwsq1996/Effi-MVS-plus.  All 2-D convolutions run on the fp32 matrix cores
(``effi_conv2d_f32``), with bias / activation / GRU gating / depth update fused into their epilogues and
channel concatenations (``torch.cat`` at update.py:41-42,46,92,94) read in place from their parts.
Inference only.
"""
from __future__ import annotations

import functools

import torch
import torch.nn as nn

from .. import ops, packing
from .module import _require_eval


def _pack(cache, conv, scale=1.0):
    return cache.get([conv.weight, conv.bias], lambda: packing.pack_conv2d(conv.weight, conv.bias, scale))


def _unbatched(t):
    """Iterate a [B,...] tensor as contiguous unbatched slices."""
    return [t[i].contiguous() for i in range(t.shape[0])]


def _stack(ts):
    """Re-attach the batch dimension (a view when B == 1)."""
    return ts[0].unsqueeze(0) if len(ts) == 1 else torch.stack(ts)


class DepthHead(nn.Module):
    """3x3 conv -> ReLU -> 3x3 conv (1 channel) -> act (reference: models/update.py:10-27)."""

    def __init__(self, input_dim=256, hidden_dim=128, scale=False):
        super().__init__()
        self.scale = scale
        self.conv1 = nn.Conv2d(input_dim, hidden_dim, 3, padding=1)
        self.conv2 = nn.Conv2d(hidden_dim, 1, 3, padding=1)
        self.relu = nn.ReLU(inplace=True)
        self.dropout = nn.Dropout2d(p=0.1)
        self._c1, self._c2, self._c2t = packing.PackCache(), packing.PackCache(), packing.PackCache()

    def run_hidden(self, net, out=None):
        w, b = _pack(self._c1, self.conv1)
        return ops.conv2d([net], w, b, self.conv1.out_channels, 3, act=ops.ACT_RELU, out0=out)

    def run_update(self, hidden, inv_depth, disp_range):
        """Fused tail: inv_new = inv_depth + tanh(conv2(hidden)); also the depth it scales to."""
        w, b = _pack(self._c2, self.conv2)
        return ops.conv2d([hidden], w, b, 1, 3, epilogue=ops.EPI_HEAD, aux0=inv_depth, disp_range=disp_range)

    def taps_fusable(self, net):
        """conv1 + ReLU + the nine tap projections of conv2 in one kernel (``run_taps``): split / bf16 precision, hidden sizes the
        fused 3x3 -> 1x1 kernel is built for."""
        c1 = self.conv1.out_channels
        if ops.option("head_taps") == 0:      # A/B switch: the two-kernel form
            return False
        return (ops.uses_split() and net.shape[-1] % 4 == 0 and c1 % 16 == 0 and c1 // 16 in (1, 2, 3, 4, 6)
                and self.conv1.in_channels % 8 == 0 and self.conv2.out_channels == 1)

    def run_taps(self, net, inv_depth, disp_range, out=None):
        """inv_new = inv_depth + tanh(conv2(relu(conv1(net)))) and its depth without the hidden map in HBM: conv2 has one output
        channel, so it is nine 1x1 projections of the hidden map (applied in conv1's epilogue) summed over the 3x3 neighbourhood."""
        w, b = _pack(self._c1, self.conv1)
        c1 = self.conv1.out_channels
        w2, b2 = self._c2t.get([self.conv2.weight], lambda: packing.pack_head_taps(self.conv2.weight, c1))
        part = ops.conv2d_k3_k1_x3([net], w.wx, b, c1, None, w2, b2, 9, relu=False, relu1=True,
                                   out=None if out is None else out[:9])
        return ops.head_update(part, self.conv2.bias, inv_depth, disp_range)

    @ops.on_tensor_device

    def forward(self, x_d, act_fn=torch.tanh):
        if self.training:
            if act_fn is not torch.tanh:
                raise NotImplementedError("DepthHead (training): act_fn=torch.tanh, as the update block calls it")
            from .. import train_path
            return train_path.depth_head(self, x_d)
        w, b = _pack(self._c2, self.conv2)
        outs = []
        for x in _unbatched(x_d):
            hid = self.run_hidden(x)
            if act_fn is torch.tanh:
                outs.append(ops.conv2d([hid], w, b, 1, 3, act=ops.ACT_TANH))
            else:
                outs.append(act_fn(ops.conv2d([hid], w, b, 1, 3, act=ops.ACT_NONE)))
        return _stack(outs)


class ConvGRU(nn.Module):
    """z, r = sigmoid(conv([h,x])); q = tanh(conv([r*h, x])); h' = (1-z)h + zq  (reference: update.py:33-49).
    convz and convr share their input and run as ONE convolution with 2*hidden output channels whose
    epilogue writes z and r*h; convq's epilogue applies the gate."""

    def __init__(self, hidden_dim=128, input_dim=192 + 128):
        super().__init__()
        self.convz = nn.Conv2d(hidden_dim + input_dim, hidden_dim, 3, padding=1)
        self.convr = nn.Conv2d(hidden_dim + input_dim, hidden_dim, 3, padding=1)
        self.convq = nn.Conv2d(hidden_dim + input_dim, hidden_dim, 3, padding=1)
        self._czr, self._cq = packing.PackCache(), packing.PackCache()

    def _packed_zr(self):
        t = [self.convz.weight, self.convz.bias, self.convr.weight, self.convr.bias]
        return self._czr.get(t, lambda: packing.pack_conv2d(
            torch.cat([self.convz.weight, self.convr.weight], 0), torch.cat([self.convz.bias, self.convr.bias], 0)))

    def run(self, h, xs, z_buf=None, rh_buf=None, out=None):
        hd = self.convz.out_channels
        if hd % 16:
            raise NotImplementedError("ConvGRU: hidden_dim must be a multiple of 16 on the HIP path")
        wzr, bzr = self._packed_zr()
        z, rh = ops.conv2d([h] + xs, wzr, bzr, 2 * hd, 3, epilogue=ops.EPI_GRU_ZR, aux0=h, out0=z_buf, out1=rh_buf)
        wq, bq = _pack(self._cq, self.convq)
        return ops.conv2d([rh] + xs, wq, bq, hd, 3, epilogue=ops.EPI_GRU_Q, aux0=h, aux1=z, out0=out)

    @ops.on_tensor_device

    def forward(self, h, *x_list):
        if len(x_list) > 2:
            raise NotImplementedError("ConvGRU: at most two input tensors besides h")
        if self.training:
            from .. import train_path
            return train_path.conv_gru(self, h, *x_list)
        xs = [_unbatched(x) for x in x_list]
        return _stack([self.run(hb, [x[i] for x in xs]) for i, hb in enumerate(_unbatched(h))])


class ProjectionInput(nn.Module):
    """Encoder of (inverse depth, cost lookup, context) (reference: models/update.py:69-99)."""

    def __init__(self, cost_dim, hidden_dim, context_dim, out_chs, depth_num=1, G=8):
        super().__init__()
        self.convc1 = nn.Conv2d(cost_dim, hidden_dim, 1, padding=0)
        self.convc2 = nn.Conv2d(hidden_dim, hidden_dim, 3, padding=1)
        self.convd1 = nn.Conv2d(depth_num, hidden_dim, 7, padding=3)
        self.convd2 = nn.Conv2d(hidden_dim, hidden_dim, 3, padding=1)
        self.convd = nn.Conv2d(hidden_dim + hidden_dim, hidden_dim - context_dim, 3, padding=1)
        self.convc = nn.Conv2d(hidden_dim, hidden_dim, 1, padding=0)
        self.out_chs = hidden_dim
        self.dropout = nn.Dropout2d(p=0.1)
        self._caches = {k: packing.PackCache() for k in ("c1", "c1raw", "c2", "d1", "d2", "d", "c", "c_after")}

    def convc1_raw(self):
        """convc1 as ([cost_dim, hidden] weight, [hidden] bias) for the lookup kernel that applies it in place."""
        return self._caches["c1raw"].get(
            [self.convc1.weight, self.convc1.bias],
            lambda: (self.convc1.weight.reshape(self.convc1.out_channels, -1).t().contiguous().float(),
                     self.convc1.bias.contiguous().float()))

    def conv7_packed(self):
        return self._caches["d1"].get([self.convd1.weight, self.convd1.bias],
                                      lambda: packing.pack_conv2d_c1k7(self.convd1.weight, self.convd1.bias))

    def run(self, disp, cost, context, bufs=None, cor1=None, inputs=None):
        """disp [1,h,w], cost [2*nq,h,w], context [cd,h,w] -> [hidden,h,w].  ``bufs``: optional dict of
        scratch tensors reused across GRU iterations; ``cor1``: relu(convc1(cost)), or a callable returning it, when the
        lookup kernel produces it (then ``cost`` is not needed); ``inputs``: callable returning (relu(convc1(cost)),
        relu(convd1(disp))) when one launch produces both (``ops.encoder_inputs``)."""
        hd = self.convc1.out_channels
        if self.convd1.in_channels != 1:
            raise NotImplementedError("ProjectionInput: depth_num must be 1 on the HIP path")
        g = (lambda k: bufs.get(k)) if bufs is not None else (lambda k: None)
        wd2, bd2 = _pack(self._caches["d2"], self.convd2)
        wc2, bc2 = _pack(self._caches["c2"], self.convc2)
        if (inputs is not None and ops.uses_split() and wd2.wx is not None and wc2.wx is not None
                and disp.shape[-1] % 4 == 0 and hd <= 64):
            # the cost chain (lookup + 1x1 -> 3x3) and the depth chain (7x7 -> 3x3) are independent: each level of the two
            # chains is ONE launch whose workgroups are shared between them (no second stream, no fork / join bubbles)
            cor1, dfm = inputs()
            w, b = _pack(self._caches["d"], self.convd)
            cmix, cd = self.convd.out_channels, context.shape[0]
            if (hd == 16 and cmix <= 16 and cd <= 16 and w.wx is not None and self.convc.in_channels == cmix + cd
                    and ops.option("enc_tail") == 1):
                # the rest of the encoder in one kernel: the two 3x3 maps of this level never reach HBM.  Opt-in (EFFI_ENC_TAIL=1):
                # measured at 592x800 the kernel takes 69.1 us, exactly the 31.5 + 37.3 us of the two launches it replaces (the
                # 120 MB it saves are paid back by 1.31x first-layer work at two workgroups per CU), and a view gets 0.8 % slower
                w2, b2 = self._caches["c_after"].get([self.convc.weight, self.convc.bias],
                                                     lambda: packing.pack_conv1x1_after(self.convc.weight, self.convc.bias, cmix, cd))
                # (its output must not alias its inputs: other workgroups still read their halos -> "enc_tail", not "enc")
                return ops.encoder_tail(cor1, dfm, wc2.wx, bc2, wd2.wx, bd2, w.wx, b, cmix, context, w2, b2, hd, out=g("enc_tail"))
            cor, dfm = ops.conv2d_k3_bf16x3_pair([cor1], wc2.wx, bc2, [dfm], wd2.wx, bd2, hd, act=ops.ACT_RELU,
                                                 out_a=g("cor2"), out_b=g("dfm2"))
        else:
            # the depth branch (7x7 -> 3x3) does not depend on the cost branch: with branches on it goes to the side stream
            with ops.Branch() as br:
                w7, b7 = self.conv7_packed()
                dfm = ops.conv2d_c1k7_relu(disp, w7, b7, hd, out=g("dfm1"))
                dfm = ops.conv2d([dfm], wd2, bd2, hd, 3, act=ops.ACT_RELU, out0=g("dfm2"))
            if callable(cor1):                     # produced by the lookup kernel, enqueued after the fork so both chains overlap
                cor1 = cor1()
            elif cor1 is None:
                w, b = _pack(self._caches["c1"], self.convc1)
                cor1 = ops.conv2d([cost], w, b, hd, 1, act=ops.ACT_RELU, out0=g("cor1"))
            cor = ops.conv2d([cor1], wc2, bc2, hd, 3, act=ops.ACT_RELU, out0=g("cor2"))
            br.join(dfm)
        w, b = _pack(self._caches["d"], self.convd)
        cmix, cd = self.convd.out_channels, context.shape[0]
        if (ops.uses_split() and w.wx is not None and cor.shape[-1] % 4 == 0 and hd % 8 == 0 and hd % 16 == 0
                and cmix <= 48 and cd <= 16 and self.convc.in_channels == cmix + cd):
            # convd (3x3) and convc (1x1 over [convd, context], ReLU) in one kernel: the intermediate stays in registers
            w2, b2 = self._caches["c_after"].get([self.convc.weight, self.convc.bias],
                                                 lambda: packing.pack_conv1x1_after(self.convc.weight, self.convc.bias, cmix, cd))
            return ops.conv2d_k3_k1_x3([cor, dfm], w.wx, b, cmix, context, w2, b2, hd, relu=True, out=g("enc"))
        mix = ops.conv2d([cor, dfm], w, b, cmix, 3, act=ops.ACT_NONE, out0=g("mix"))
        w, b = _pack(self._caches["c"], self.convc)
        return ops.conv2d([mix, context], w, b, hd, 1, act=ops.ACT_RELU, out0=g("enc"))

    @ops.on_tensor_device

    def forward(self, disp, cost, context):
        if self.training:
            from .. import train_path
            return train_path.projection_input(self, disp, cost, context)
        d, c, x = _unbatched(disp), _unbatched(cost), _unbatched(context)
        return _stack([self.run(d[i], c[i], x[i]) for i in range(len(d))])


class BasicUpdateBlock(nn.Module):
    """seq_len x {cost lookup -> encoder -> ConvGRU -> depth head}; the last iteration also emits the
    36-channel convex-upsampling mask (reference: models/update.py:101-141)."""

    def __init__(self, hidden_dim=128, cost_dim=256, ratio=8, context_dim=64, UpMask=False, Inverse=False,
                 cost_num=1, G=8):
        super().__init__()
        self.encoder = ProjectionInput(cost_dim=cost_dim * cost_num, hidden_dim=hidden_dim, context_dim=context_dim,
                                       out_chs=hidden_dim, G=G)
        self.depth_gru = ConvGRU(hidden_dim=hidden_dim, input_dim=self.encoder.out_chs)
        self.depth_head = DepthHead(hidden_dim, hidden_dim=hidden_dim, scale=False)
        self.UpMask = UpMask
        self.Inverse = Inverse
        self.mask = nn.Sequential(nn.Conv2d(hidden_dim, hidden_dim * 2, 3, padding=1), nn.ReLU(inplace=True),
                                  nn.Conv2d(hidden_dim * 2, ratio * ratio * 9, 1, padding=0))
        self._m0, self._m2, self._m2x, self._m2u = (packing.PackCache() for _ in range(4))
        self.last_depths = None     # depths of the last forward's iterations (filled on the fused path)

    def run_mask(self, net):
        """0.25 * mask(net); the factor is folded into the 1x1 conv's weights and bias (exact: power of two)."""
        w, b = _pack(self._m0, self.mask[0])
        c1, c2 = self.mask[0].out_channels, self.mask[2].out_channels
        if (ops.uses_split() and w.wx is not None and net.shape[-1] % 4 == 0 and c1 <= 96 and c1 // 16 in (1, 2, 3, 4, 6)
                and c1 % 16 == 0 and c2 <= 96):
            # 3x3 + ReLU + 1x1 (x0.25) in one kernel: the 2*hidden-channel intermediate stays in registers
            w2, b2 = self._m2x.get([self.mask[2].weight, self.mask[2].bias],
                                   lambda: packing.pack_conv1x1_after(self.mask[2].weight, self.mask[2].bias, c1, 0, scale=0.25))
            return ops.conv2d_k3_k1_x3([net], w.wx, b, c1, None, w2, b2, c2, relu=False, relu1=True)
        hid = ops.conv2d([net], w, b, c1, 3, act=ops.ACT_RELU)
        w, b = _pack(self._m2, self.mask[2], scale=0.25)
        return ops.conv2d([hid], w, b, self.mask[2].out_channels, 1, act=ops.ACT_NONE)

    def mask_upsample_fusable(self, net):
        c1, c2 = self.mask[0].out_channels, self.mask[2].out_channels
        return (ops.uses_split() and net.shape[-1] % 4 == 0 and c1 in (32, 64, 96) and c2 == 36
                and self.mask[0].in_channels % 8 == 0)

    def run_mask_upsample(self, net, inv_depth, disp_range):
        """0.25 * mask(net) and the convex x2 upsampling of ``inv_depth`` with it in one kernel (the mask stays in registers)
        -> (depth [2h,2w], depth_to_inv(depth) [2h,2w])."""
        w, b = _pack(self._m0, self.mask[0])
        c1 = self.mask[0].out_channels
        w2, b2 = self._m2u.get([self.mask[2].weight, self.mask[2].bias],
                               lambda: packing.pack_mask_taps_per_lane(self.mask[2].weight, self.mask[2].bias, c1, scale=0.25))
        return ops.conv2d_k3_k1_up2x([net], w.wx, b, c1, w2, b2, inv_depth, disp_range)

    # -- fused path on split-resident maps (ops.SRMap): every 3x3 convolution of an iteration reads ready-made (hi, lo) bf16 octets
    # and writes its result the same way; bitwise the results of ``run_fused``'s fp32-map chain --------------------------------
    N_SR_MAPS = 5          # A: cor1 -> x, B: dfm1 -> r*h, C: cor2, D: dfm2, H: hidden state

    def _sr_structure_ok(self, hd):
        e = self.encoder
        cmix, cd = e.convd.out_channels, e.convc.in_channels - e.convd.out_channels
        c1m = self.mask[0].out_channels
        return (hd in (16, 32, 48) and e.convd1.in_channels == 1 and cmix <= 48 and 0 <= cd <= 16 and e.convc.out_channels == hd
                and self.depth_head.conv1.out_channels == hd and self.depth_head.conv1.in_channels == hd
                and self.depth_head.conv2.out_channels == 1 and (not self.UpMask or (c1m in (32, 64, 96) and self.mask[2].out_channels == 36)))

    def sr_fusable(self, net, lookup):
        return (ops.uses_sr(net.shape[-1] * net.shape[-2]) and net.shape[-1] % 4 == 0 and getattr(lookup, "encoder_inputs_sr", None) is not None
                and self._sr_structure_ok(net.shape[0]))

    def state_q4_ok(self, fuse_upsample):
        """The fp32 copy of the hidden state (and z) may live in the Q4 layout [hd/4][h][w][4] (ops.EPI_Q4) when nothing outside the
        two GRU epilogues reads it: the mask head must be the fused one (it reads the SR state), the one-launch ConvGRU off."""
        return (bool(ops.option("state_q4")) and fuse_upsample and self.UpMask and not ops.option("gru_fused")
                and self._sr_structure_ok(self.depth_head.conv1.in_channels))

    def run_fused_sr(self, net, lookup, inv_depth, context, seq_len, disp_range, fuse_upsample=False, maps=None, net_sr_ready=False,
                     net_owned=False, net_q4=False):
        """``run_fused`` with the iteration's maps split-resident.  ``maps``: the N_SR_MAPS SRMaps [A, B, C, D, H] of this block
        (borders already zero; allocated here otherwise); ``net_sr_ready``: H already holds ``net`` (written by
        ``ops.split_tanh_relu_stages_sr``), otherwise it is converted here; ``net_owned``: ``net`` is a temporary of the caller
        that may be overwritten (the fp32 copy of the state is then updated in place from the first iteration on).

        Working set of an iteration at 592x800 (hd 16): five SR maps of 31.7 MB, the fp32 state and ONE more fp32 block that is z
        between the z / r convolution and the update and the depth head's nine tap planes after it -- 220 MB, inside the 256-MB
        MALL; with separate z / head / ping-pong state buffers (280 MB) stage 3 ran 7 % SLOWER than on fp32 maps."""
        hd = net.shape[0]
        h, w = net.shape[-2:]
        dev = net.device
        if maps is None:
            maps = ops.sr_alloc(self.N_SR_MAPS, hd, h, w, dev)
        A, B, Cm, Dm, Hm = maps
        if net_q4 and not (net_sr_ready and net_owned and self.state_q4_ok(fuse_upsample)):
            raise ValueError("run_fused_sr: a Q4 state needs net_sr_ready, net_owned and the fused mask head")
        q4 = net_q4                              # the returned ``net`` is then Q4 too (the cascade does not read it)
        if not net_sr_ready:
            ops.sr_from_planar(net, out=Hm)
        gru_fused = hd <= 16 * ops.option("gru_fused")          # option gru_fused: 0 off, 1 = hd 16, 2 = hd 16 and 32
        zh_buf = torch.empty(9 if gru_fused else max(hd, 9), h, w, device=dev, dtype=torch.float32)   # z, then (z is dead) the head's 9 tap planes
        z_buf, head_buf = (None if gru_fused else zh_buf[:hd]), zh_buf[:9]
        h_own = net if net_owned else None       # fp32 state we may write: each lane reads h[p] and writes h'[p] of its own pixel
        e = self.encoder
        wc1, bc1 = e.convc1_raw()
        w7, b7 = e.conv7_packed()
        wd2, bd2 = _pack(e._caches["d2"], e.convd2)
        wc2, bc2 = _pack(e._caches["c2"], e.convc2)
        wd, bd = _pack(e._caches["d"], e.convd)
        cmix, cd = e.convd.out_channels, context.shape[0]
        wca, bca = e._caches["c_after"].get([e.convc.weight, e.convc.bias],
                                            lambda: packing.pack_conv1x1_after(e.convc.weight, e.convc.bias, cmix, cd))
        wzr, bzr = self.depth_gru._packed_zr()
        wq, bq = _pack(self.depth_gru._cq, self.depth_gru.convq)
        dh = self.depth_head
        wh1, bh1 = _pack(dh._c1, dh.conv1)
        wh2, bh2 = dh._c2t.get([dh.conv2.weight], lambda: packing.pack_head_taps(dh.conv2.weight, hd))
        inv_list, mask_list, depth_list = [], [], []
        gen_pair = ops.option("enc_gen") != 0 and getattr(lookup, "encoder_pair_sr", None) is not None
        H_first = Hm
        h_pp = [None, net if net_owned else None]               # fp32 state buffers of the fused form: iteration i writes h_pp[i % 2]
                                                                # (iteration 0 reads the caller's ``net``; it is reused from iteration 1 on if ours)
        for i in range(seq_len):
            if gen_pair:                       # cor1 / dfm1 generated inside the pair kernel: cor -> C, dfm -> D in one launch
                lookup.encoder_pair_sr(inv_depth, wc1, bc1, w7, b7, hd, wc2.wx, bc2, Cm, wd2.wx, bd2, Dm)
            else:
                lookup.encoder_inputs_sr(inv_depth, wc1, bc1, w7, b7, hd, A, B)                        # cor1 -> A, dfm1 -> B
                ops.conv2d_k3_pair_sr([A], wc2.wx, bc2, Cm, [B], wd2.wx, bd2, Dm, hd, act=ops.ACT_RELU)  # cor -> C, dfm -> D
            ops.conv2d_k3_k1_sr([Cm, Dm], wd.wx, bd, cmix, context, wca, bca, hd, relu=True, out_sr=A)   # x -> A (cor1 is dead)
            if gru_fused:
                # ConvGRU as one launch (csrc/gru_fused.hpp): r * h and z stay on chip.  A workgroup reads its neighbours' pixels of
                # the state, so the state ping-pongs between two buffers in both forms (H <-> B as SR maps; two fp32 maps)
                h_next = h_pp[i % 2]
                if h_next is None or h_next.data_ptr() == net.data_ptr():
                    h_next = h_pp[i % 2] = torch.empty(hd, h, w, device=dev, dtype=torch.float32)
                H_next = B if Hm is not B else H_first
                net, _ = ops.gru_zr_q_fused_sr(Hm, A, net, wzr.wx, bzr, wq.wx, bq, h_next, H_next)
                Hm = H_next
            else:
                z, _ = ops.conv2d_k3_sr([Hm, A], wzr.wx, bzr, 2 * hd, epilogue=ops.EPI_GRU_ZR, aux0=net, out0=z_buf, out_sr=B, q4=q4)   # r*h -> B
                # the new state overwrites H in place (no launch reads H between the z / r convolution and here), and so does its fp32
                # copy once it lives in a buffer of ours
                if h_own is None:
                    h_own = torch.empty(hd, h, w, device=dev, dtype=torch.float32)
                net, _ = ops.conv2d_k3_sr([B, A], wq.wx, bq, hd, epilogue=ops.EPI_GRU_Q, aux0=net, aux1=z, out0=h_own, out_sr=Hm, q4=q4)
            want_mask = self.UpMask and i == seq_len - 1
            fused_up = want_mask and fuse_upsample
            if want_mask and not fused_up:
                with ops.Branch() as br:
                    mask = self.run_mask(net)
            part = ops.conv2d_k3_k1_sr([Hm], wh1.wx, bh1, hd, None, wh2, bh2, 9, relu=False, relu1=True, out=head_buf)
            inv_depth, depth = ops.head_update(part, dh.conv2.bias, inv_depth, disp_range)
            if fused_up:                       # needs the NEW inverse depth: after the head
                wm, bm = _pack(self._m0, self.mask[0])
                c1 = self.mask[0].out_channels
                w2, b2 = self._m2u.get([self.mask[2].weight, self.mask[2].bias],
                                       lambda: packing.pack_mask_taps_per_lane(self.mask[2].weight, self.mask[2].bias, c1, scale=0.25))
                mask = ops.conv2d_k3_k1_up2x_sr([Hm], wm.wx, bm, c1, w2, b2, inv_depth, disp_range)
            elif want_mask:
                br.join(mask)
            inv_list.append(inv_depth)
            depth_list.append(depth)
            mask_list.append(mask if want_mask else inv_depth)
        return net, mask_list, inv_list, depth_list

    # -- fused path: the cost lookup is our GetCost and scale_inv_depth is the global-range rescale ------
    def run_fused(self, net, lookup, inv_depth, context, seq_len, disp_range, fuse_upsample=False, sr_maps=None, net_sr_ready=False,
                  net_owned=False, net_q4=False):
        """Unbatched tensors; ``lookup(inv_depth, out)`` fills the [2*nq,h,w] cost for a normalised
        inverse-depth map.  Returns (net, mask_list, inv_list, depth_list).  ``fuse_upsample``: the last iteration's mask
        head and the convex upsampling it feeds run as one kernel; mask_list[-1] is then the pair (upsampled depth,
        depth_to_inv of it) instead of the mask."""
        if self.sr_fusable(net, lookup):
            return self.run_fused_sr(net, lookup, inv_depth, context, seq_len, disp_range, fuse_upsample, sr_maps, net_sr_ready, net_owned, net_q4)
        if net_q4:
            raise ValueError("run_fused: a Q4 state was passed but the block does not take the split-resident path")
        hd = net.shape[0]
        h, w = net.shape[-2:]
        dev = net.device
        mk = lambda c: torch.empty(c, h, w, device=dev, dtype=torch.float32)
        bufs = {"cor1": mk(hd), "cor2": mk(hd), "dfm1": mk(hd), "dfm2": mk(hd)}
        # Buffers are recycled inside an iteration so that its working set stays small (at 592x800 one iteration touched ~390 MB
        # of 16-channel maps, more than the 256 MB MALL; measured 2.85 -> 2.80 ms per view): four map buffers serve the eight
        # intermediates (cor1 -> enc, dfm1 -> z, cor2 -> r*h, dfm2 -> head hidden: each successor is written only after its
        # predecessor's last reader, on both streams), and the hidden state ping-pongs between two buffers (only the last state
        # is returned).
        bufs["enc"] = bufs["cor1"]                # the encoder's output overwrites its first intermediate (dead after convc2)
        bufs["enc_tail"] = bufs["dfm2"]           # one-kernel encoder tail: reads cor1 / dfm1 with halos, so it writes elsewhere
                                                  # (dfm2 is free until the depth head, which runs after the last reader of x)
        z_buf, rh_buf, head_buf, cost_buf = bufs["dfm1"], bufs["cor2"], bufs["dfm2"], None
        h_bufs = [mk(hd), mk(hd)]
        inv_list, mask_list, depth_list = [], [], []
        fuse_c1 = getattr(lookup, "conv1x1", None) is not None and hd % 8 == 0
        fuse_in = fuse_c1 and getattr(lookup, "encoder_inputs", None) is not None and hd in (16, 32, 48)
        for i in range(seq_len):
            if fuse_c1:      # lookup + convc1 + ReLU in one kernel: the cost map never reaches HBM
                wc1, bc1 = self.encoder.convc1_raw()
                both = None
                if fuse_in:
                    w7, b7 = self.encoder.conv7_packed()
                    both = lambda d=inv_depth: lookup.encoder_inputs(d, wc1, bc1, w7, b7, hd, bufs["cor1"], bufs["dfm1"])
                x = self.encoder.run(inv_depth, None, context, bufs,
                                     cor1=lambda d=inv_depth: lookup.conv1x1(d, wc1, bc1, hd, bufs["cor1"]), inputs=both)
            else:
                cost_buf = lookup(inv_depth, cost_buf)
                x = self.encoder.run(inv_depth, cost_buf, context, bufs)
            net = self.depth_gru.run(net, [x], z_buf, rh_buf, out=h_bufs[i % 2])
            want_mask = self.UpMask and i == seq_len - 1
            fused_up = want_mask and fuse_upsample and self.mask_upsample_fusable(net)
            if want_mask and not fused_up:     # the mask head only needs the new hidden state (side stream when branches are on)
                with ops.Branch() as br:
                    mask = self.run_mask(net)
            if self.depth_head.taps_fusable(net):
                inv_depth, depth = self.depth_head.run_taps(net, inv_depth, disp_range, head_buf)
            else:
                hid = self.depth_head.run_hidden(net, head_buf)
                inv_depth, depth = self.depth_head.run_update(hid, inv_depth, disp_range)
            if fused_up:                       # needs the NEW inverse depth: after the head
                mask = self.run_mask_upsample(net, inv_depth, disp_range)
            elif want_mask:
                br.join(mask)
            inv_list.append(inv_depth)
            depth_list.append(depth)
            mask_list.append(mask if want_mask else inv_depth)
        return net, mask_list, inv_list, depth_list

    @ops.on_tensor_device

    def forward(self, net, depth_cost_func, inv_depth, context, seq_len=4, scale_inv_depth=None):
        if self.training:
            from .. import train_path
            return train_path.update_block(self, net, lambda inv, i: depth_cost_func(scale_inv_depth(inv)[1], iter=i), inv_depth,
                                           context, seq_len)
        from .Effi_MVS_plus import GetCost, disp_to_depth     # late import (module cycle)
        B = net.shape[0]
        fused = (isinstance(depth_cost_func, functools.partial) and isinstance(depth_cost_func.func, GetCost)
                 and isinstance(scale_inv_depth, functools.partial) and scale_inv_depth.func is disp_to_depth
                 and hasattr(scale_inv_depth, "effi_disp_range"))
        nets, masks, invs, depths = [], [], [], []
        for b in range(B):
            nb, ib, cb = net[b].contiguous(), inv_depth[b].contiguous(), context[b].contiguous()
            if fused:
                lookup = depth_cost_func.func.make_lookup(b, disp_range=scale_inv_depth.effi_disp_range,
                                                          **depth_cost_func.keywords)
                n, m, iv, dp = self.run_fused(nb, lookup, ib, cb, seq_len, scale_inv_depth.effi_disp_range[b])
            else:
                # generic callables: same kernels, but the lookup goes through the public GetCost signature
                n, m, iv, dp = nb, [], [], None
                cur = ib
                for i in range(seq_len):
                    depth = scale_inv_depth(cur.unsqueeze(0))[1]
                    cost = depth_cost_func(depth, iter=i)[0].contiguous()
                    x = self.encoder.run(cur, cost, cb)
                    n = self.depth_gru.run(n, [x])
                    cur = cur + self.depth_head(n.unsqueeze(0))[0]
                    iv.append(cur)
                    m.append(self.run_mask(n) if (self.UpMask and i == seq_len - 1) else cur)
            nets.append(n), masks.append(m), invs.append(iv), depths.append(dp)
        self.last_depths = None if depths[0] is None else [_stack([d[i] for d in depths]) for i in range(seq_len)]
        net_out = _stack(nets)
        mask_list = [_stack([m[i] for m in masks]) for i in range(seq_len)]
        inv_list = [_stack([v[i] for v in invs]) for i in range(seq_len)]
        return net_out, mask_list, inv_list
