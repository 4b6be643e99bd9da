"""Host-side weight preparation: BatchNorm folding and re-ordering into the layouts the kernels read.

Runs once per weight version (cached per module, invalidated when a parameter is replaced or
modified in place, e.g. by ``load_state_dict`` / ``.to()``); it is set-up work, not part of the
timed path.
"""
from __future__ import annotations

from . import _lib

import torch


def params_key(*tensors):
    """Cheap fingerprint of a set of parameters/buffers: storage address + in-place version."""
    return tuple((t.data_ptr(), t._version, t.device) for t in tensors if t is not None)


class PackCache:
    """Per-module cache of packed weights keyed by ``params_key``."""

    def __init__(self):
        self._key = None
        self._val = None

    def get(self, tensors, builder):
        key = params_key(*tensors)
        if key != self._key:
            with torch.no_grad():
                self._val = builder()
            self._key = key
        return self._val


def bn_scale_shift(bn):
    """Eval-mode BatchNorm as y = x*scale + shift (models/module.py:148,191,217; eps = bn.eps)."""
    scale = bn.weight / torch.sqrt(bn.running_var + bn.eps)
    shift = bn.bias - bn.running_mean * scale
    return scale, shift


def pack_conv3d(conv, bn):
    """nn.Conv3d [cout,cin,3,3,3] (+BN) -> (weight [cin,27,cout], bias [cout] or None)."""
    w = conv.weight
    bias = conv.bias
    if bn is not None:
        scale, shift = bn_scale_shift(bn)
        w = w * scale.view(-1, 1, 1, 1, 1)
        bias = shift if bias is None else bias * scale + shift
    cout, cin = w.shape[0], w.shape[1]
    wp = w.permute(1, 2, 3, 4, 0).reshape(cin, 27, cout).contiguous().float()
    return wp, (None if bias is None else bias.contiguous().float())


def pack_conv3d_planes(conv, bn):
    """nn.Conv3d [cout,cin,3,3,3] (+BN) for the z-batched matrix-core path: the 3-D conv of output plane z is a 2-D
    conv over cat(plane z-1, z, z+1), i.e. weight [cout][kd*cin+ci][ky][kx] in MFMA packing; bias [16*NT]."""
    w = conv.weight
    bias = conv.bias
    if bn is not None:
        scale, shift = bn_scale_shift(bn)
        w = w * scale.view(-1, 1, 1, 1, 1)
        bias = shift if bias is None else bias * scale + shift
    cout, cin = w.shape[0], w.shape[1]
    w2 = w.permute(0, 2, 1, 3, 4).reshape(cout, 3 * cin, 3, 3)
    return pack_conv2d_mfma(w2, bias)


def pack_conv3d_planes_bf16x3(conv, bn):
    """Same view of the weight as ``pack_conv3d_planes`` in the split-bf16 operand order (``pack_conv2d_bf16x3``)."""
    w = conv.weight
    bias = conv.bias
    if bn is not None:
        scale, shift = bn_scale_shift(bn)
        w = w * scale.view(-1, 1, 1, 1, 1)
        bias = shift if bias is None else bias * scale + shift
    cout, cin = w.shape[0], w.shape[1]
    w2 = w.permute(0, 2, 1, 3, 4).reshape(cout, 3 * cin, 3, 3)
    return pack_conv2d_bf16x3(w2, bias)


def pack_conv3d_s2_bf16x3(conv, bn):
    """nn.Conv3d [cout,cin,3,3,3] stride 2 (+BN) for ``effi_conv3d_k3s2_bf16x3_f32``: chunk = (kd, octet of input channels), K item =
    tap ky*3 + kx (9 items, 3 K-steps), output tiles in groups of NT (1 if cout <= 16 else 2).
    -> (bf16 [3 * ceil(cin/8), G, 3, NT, 2(hi|lo), 64, 8], bias fp32 [16 * NT * G])."""
    w = conv.weight
    bias = conv.bias
    if bn is not None:
        scale, shift = bn_scale_shift(bn)
        w = w * scale.view(-1, 1, 1, 1, 1)
        bias = shift if bias is None else bias * scale + shift
    cout, cin = w.shape[0], w.shape[1]
    nt = 1 if cout <= 16 else 2
    g = (cout + 16 * nt - 1) // (16 * nt)
    noct = (cin + 7) // 8
    wz = torch.zeros(g * nt * 16, 3, noct * 8, 12, device=w.device, dtype=torch.float32)       # [co, kd, ci, tap]
    wz[:cout, :, :cin, :9] = w.float().permute(0, 2, 1, 3, 4).reshape(cout, 3, cin, 9)
    # [g, n, j, kd, oct, e, s, q] -> [kd, oct, g, s, n, q, j, e]
    wz = wz.view(g, nt, 16, 3, noct, 8, 3, 4).permute(3, 4, 0, 6, 1, 7, 2, 5).contiguous()
    hi = wz.to(torch.bfloat16)
    lo = (wz - hi.float()).to(torch.bfloat16)
    wp = torch.stack([hi, lo], dim=5).contiguous().view(3 * noct, g, 3, nt, 2, 64, 8)
    b = torch.zeros(g * nt * 16, device=w.device, dtype=torch.float32)
    if bias is not None:
        b[:cout] = bias.float()
    return wp, b


def pack_conv3d_roll_bf16x3(conv, bn):
    """nn.Conv3d [cout,cin,3,3,3] (+BN), cin in {8,16}, for the rolling-window split-bf16 kernel:
    K index = (kd, ky, kx, octet, e); K-step s takes items 4s..4s+3 of (kd, ky, kx, octet); lane = q*16 + j holds
    W[16n+j][oct*8+e][kd][ky][kx] for item 4s+q.  -> (bf16 [NKS, NT, 2, 64, 8], bias fp32 [16*NT])."""
    w = conv.weight
    bias = conv.bias
    if bn is not None:
        scale, shift = bn_scale_shift(bn)
        w = w * scale.view(-1, 1, 1, 1, 1)
        bias = shift if bias is None else bias * scale + shift
    cout, cin = w.shape[0], w.shape[1]
    assert cin in (8, 16)
    if cout <= 8 and _lib.lib().effi_get_option(b"roll_rp") != 0:      # (A/B switch shared with the kernel's launch rule)
        return _pack_conv3d_roll_rowpair(w, bias)
    noct, nt = cin // 8, (cout + 15) // 16
    nit = 27 * noct
    nks = (nit + 3) // 4
    wz = torch.zeros(nt * 16, 4 * nks, 8, device=w.device, dtype=torch.float32)          # [cout, item, e]
    # [cout, cin, 27] -> [cout, 27, oct, e] -> items (tap3d, oct)
    wz[:cout, :nit] = w.reshape(cout, noct, 8, 27).permute(0, 3, 1, 2).reshape(cout, nit, 8).float()
    wz = wz.view(nt, 16, nks, 4, 8).permute(2, 0, 3, 1, 4).contiguous()                  # [s, n, q, j, e]
    hi = wz.to(torch.bfloat16)
    lo = (wz - hi.float()).to(torch.bfloat16)
    wp = torch.stack([hi, lo], dim=2).contiguous().view(nks, nt, 2, 64, 8)
    b = torch.zeros(nt * 16, device=w.device, dtype=torch.float32)
    if bias is not None:
        b[:cout] = bias.float()
    return wp, b


def _pack_conv3d_roll_rowpair(w, bias):
    """cout <= 8: the ROW-PAIR operand of the rolling-window kernel (csrc/conv2d.hip, conv3d_roll_rp_bf16x3_body): MFMA rows 0-7 are
    the output channels of image row y, rows 8-15 the same channels of row y + 1; K index = (kd, dy in 0..3, kx, octet, e) over the
    4 x 3 window both rows see: W[j][c][kd][dy][kx] for rows j < 8 (zero at dy = 3), W[j - 8][c][kd][dy - 1][kx] for rows j >= 8
    (zero at dy = 0).  -> (bf16 [9 * cin / 8, 1, 2, 64, 8], bias fp32 [16])."""
    cout, cin = w.shape[0], w.shape[1]
    noct = cin // 8
    nks = 9 * noct
    wf = w.float().reshape(cout, noct, 8, 3, 3, 3)                                       # [co, oct, e, kd, ky, kx]
    wz = torch.zeros(16, 3, 4, 3, noct, 8, device=w.device, dtype=torch.float32)         # [row, kd, dy, kx, oct, e]
    src = wf.permute(0, 3, 4, 5, 1, 2)                                                   # [co, kd, ky, kx, oct, e]
    wz[:cout, :, 0:3] = src
    wz[8:8 + cout, :, 1:4] = src
    wz = wz.reshape(16, nks, 4, 8).permute(1, 2, 0, 3).contiguous()                      # [s, q, j, e]
    hi = wz.to(torch.bfloat16)
    lo = (wz - hi.float()).to(torch.bfloat16)
    wp = torch.stack([hi, lo], dim=1).contiguous().view(nks, 1, 2, 64, 8)
    b = torch.zeros(16, device=w.device, dtype=torch.float32)
    if bias is not None:
        b[:cout] = bias.float()
    return wp, b


def pack_deconv3d(conv, bn):
    """nn.ConvTranspose3d [cin,cout,3,3,3] (+BN) -> (weight [cin,27,cout], bias [cout] or None)."""
    w = conv.weight
    bias = conv.bias
    if bn is not None:
        scale, shift = bn_scale_shift(bn)
        w = w * scale.view(1, -1, 1, 1, 1)
        bias = shift if bias is None else bias * scale + shift
    cin, cout = w.shape[0], w.shape[1]
    wp = w.permute(0, 2, 3, 4, 1).reshape(cin, 27, cout).contiguous().float()
    return wp, (None if bias is None else bias.contiguous().float())


def deconv_s2_slots():
    """(parity p, K-step s) pairs of the stride-2 transposed conv GEMM in kernel order: p = pz*4+py*2+px,
    s = nz*2+ny; K-step s reaches parity p iff nz <= pz and ny <= py."""
    return [(p, s) for p in range(8) for s in range(4) if (s & ~(p >> 1)) == 0]


def pack_deconv3d_s2_bf16x3(conv, bn):
    """nn.ConvTranspose3d [cin,cout,3,3,3] (+BN), stride 2 / padding 1 / output_padding 1, cin % 16 == 0, cout <= 16 ->
    (bf16 [cin/16, 18 (9 when cout <= 8), 2(hi|lo), 64, 8], bias fp32 [16]).  Per dimension parity 0 takes tap 1 of neighbour 0; parity 1 takes
    tap 2 of neighbour 0 and tap 0 of neighbour 1.  Lane = q*16 + j of slot (p, s): neighbour (nz, ny, nx = q >> 1),
    octet q & 1, output channel j, element e = input channel chunk*16 + octet*8 + e."""
    w = conv.weight
    bias = conv.bias
    if bn is not None:
        scale, shift = bn_scale_shift(bn)
        w = w * scale.view(1, -1, 1, 1, 1)
        bias = shift if bias is None else bias * scale + shift
    cin, cout = w.shape[0], w.shape[1]
    assert cin % 16 == 0 and cout <= 16
    nch = cin // 16
    tap = {(0, 0): 1, (1, 0): 2, (1, 1): 0}                 # (parity, neighbour) -> kernel index; (0, 1) contributes nothing
    half = cout <= 8            # the two x-parities share one 16-row tile: row j = px*8 + co (9 fragments instead of 18)
    slots = [(p, s_) for (p, s_) in deconv_s2_slots() if not half or (p & 1) == 0]
    wp = torch.zeros(nch, len(slots), 4, 16, 8, device=w.device, dtype=torch.float32)      # [chunk, slot, q, j, e]
    wf = w.float()
    for si, (p, s_) in enumerate(slots):
        pz, py = p >> 2, (p >> 1) & 1
        nz, ny = s_ >> 1, s_ & 1
        for q in range(4):
            nx, octet = q >> 1, q & 1
            for px in ((0, 1) if half else (p & 1,)):
                if (pz, nz) not in tap or (py, ny) not in tap or (px, nx) not in tap:
                    continue
                kz, ky, kx = tap[(pz, nz)], tap[(py, ny)], tap[(px, nx)]
                blk = wf[:, :, kz, ky, kx].reshape(nch, 2, 8, cout)[:, octet]             # [chunk, e, cout]
                j0 = px * 8 if half else 0
                wp[:, si, q, j0:j0 + cout, :] = blk.permute(0, 2, 1)
    wp = wp.view(nch, len(slots), 64, 8)
    hi = wp.to(torch.bfloat16)
    lo = (wp - hi.float()).to(torch.bfloat16)
    out = torch.stack([hi, lo], dim=2).contiguous()                                      # [chunk, slot, hl, lane, e]
    b = torch.zeros(16, device=w.device, dtype=torch.float32)
    if bias is not None:
        b[:cout] = bias.float()
    return out, b


def pack_conv2d_mfma(weight, bias, scale=1.0):
    """[cout,cin,ks,ks] (+bias [cout]) -> (wpack [ceil(cin/4), ks*ks, ceil(cout/16), 64], bias [16*NT]).

    Lane order of the v_mfma_f32_16x16x4_f32 B operand: lane = k*16 + j holds W[cout=16n+j][cin=4g+k].
    """
    cout, cin, ks, _ = weight.shape
    nt, kg = (cout + 15) // 16, (cin + 3) // 4
    w = torch.zeros(nt * 16, kg * 4, ks * ks, device=weight.device, dtype=torch.float32)
    w[:cout, :cin] = weight.reshape(cout, cin, ks * ks).float() * scale
    w = w.view(nt, 16, kg, 4, ks * ks).permute(2, 4, 0, 3, 1).contiguous()     # [kg, tap, nt, k, j]
    b = torch.zeros(nt * 16, device=weight.device, dtype=torch.float32)
    if bias is not None:
        b[:cout] = bias.float() * scale
    w = w.view(kg, ks * ks, nt, 64)
    if cout == 1 and ks == 3:
        # single-output-channel convs run on the vector ALUs and read plain [cin][9] weights, which ride
        # behind the MFMA block in the same buffer (effi_conv2d_f32 finds them at offset kg*9*64)
        raw = (weight.reshape(cin, 9).float() * scale).reshape(-1)
        flat = torch.cat([w.reshape(-1), raw]).contiguous()
        return flat, b
    return w, b


def pack_conv2d_bf16x3(weight, bias, scale=1.0):
    """[cout,cin,3,3] (+bias) -> (wpack bf16 [ceil(cin/16), 5, NT, 2(hi|lo), 64, 8], bias fp32 [16*NT]).

    Operand order of effi_conv2d_k3_bf16x3_f32: inside a 16-channel chunk the K index is (tap, octet); K-step s of
    v_mfma_f32_16x16x32_bf16 takes items 4s..4s+3, lane = q*16 + j holds W[cout=16n+j][chunk*16 + oct*8 + e][tap] for
    item 4s+q = 2*tap + oct (items 18, 19 are zero).  hi = bf16(W), lo = bf16(W - hi).
    """
    cout, cin, ks, _ = weight.shape
    assert ks == 3
    nt, nch = (cout + 15) // 16, (cin + 15) // 16
    w = torch.zeros(nt * 16, nch * 16, 10, device=weight.device, dtype=torch.float32)
    w[:cout, :cin, :9] = weight.reshape(cout, cin, 9).float() * scale
    # [n, j, chunk, oct, e, tap] -> [chunk, tap, oct, n, j, e]; (tap, oct) flattens to item = 2*tap + oct = 4*s + q
    w = w.view(nt, 16, nch, 2, 8, 10).permute(2, 5, 3, 0, 1, 4).contiguous()
    w = w.view(nch, 5, 4, nt, 16, 8).permute(0, 1, 3, 2, 4, 5).contiguous()      # [chunk, s, n, q, j, e]
    hi = w.to(torch.bfloat16)
    lo = (w - hi.float()).to(torch.bfloat16)
    wp = torch.stack([hi, lo], dim=3).contiguous().view(nch, 5, nt, 2, 64, 8)    # [chunk, s, n, hl, lane, e]
    b = torch.zeros(nt * 16, device=weight.device, dtype=torch.float32)
    if bias is not None:
        b[:cout] = bias.float() * scale
    return wp, b


def pack_conv2d_bf16x3_oct(weight, bias, scale=1.0):
    """[cout <= 8, cin <= 8, 3, 3] (+bias) for the one-octet layers of ``effi_conv2d_k3_twice_bf16x3_f32``, ROW-PAIR operand: MFMA rows
    0-7 are the output channels of an image row, rows 8-15 the same channels of the row below; K index = tap of the 4 x 3 window both
    rows see, tap = dy*3 + dx = 4 s + q: lane = q*16 + j holds W[j][e][dy][dx] for j < 8 (zero at dy = 3) and W[j-8][e][dy-1][dx] for
    j >= 8 (zero at dy = 0); zero for e >= cin, channels >= cout.  -> (bf16 [3, 2(hi|lo), 64, 8], bias fp32 [16], entries 0..7 used)."""
    cout, cin, ks, _ = weight.shape
    assert ks == 3 and cout <= 8 and cin <= 8
    wf = weight.float() * scale
    w = torch.zeros(16, 8, 4, 3, device=weight.device, dtype=torch.float32)     # [row, e, dy, dx]
    w[:cout, :cin, 0:3] = wf
    w[8:8 + cout, :cin, 1:4] = wf
    w = w.reshape(16, 8, 3, 4).permute(2, 3, 0, 1).contiguous()                 # taps (dy, dx) flattened = 4 s + q -> [s, q, j, e]
    hi = w.to(torch.bfloat16)
    lo = (w - hi.float()).to(torch.bfloat16)
    wp = torch.stack([hi, lo], dim=1).contiguous().view(3, 2, 64, 8)
    b = torch.zeros(16, device=weight.device, dtype=torch.float32)
    if bias is not None:
        b[:cout] = bias.float() * scale
    return wp, b


def pack_conv1x1_after(weight, bias, cout1, c_extra, scale=1.0):
    """1x1 conv [cout2, cout1 + c_extra, 1, 1] applied inside the epilogue of a 3x3 split-precision conv with cout1 outputs
    (``effi_conv2d_k3_k1_bf16x3_f32``): A fragments of v_mfma_f32_16x16x16_bf16 per (output tile t, input tile n):
    lane = q*16 + j holds W[16t + j][16n + 4q + e] (n < ceil(cout1/16); zero beyond cout1 / cout2) and, for the last input
    tile, the extra channels W[16t + j][cout1 + 4q + e] (zero beyond c_extra).
    -> (bf16 [NT2, NT1+1, 2, 64, 4], bias fp32 [16*NT2]); ``scale`` multiplies weights and bias (exact for powers of two)."""
    cout2 = weight.shape[0]
    assert weight.shape[1] == cout1 + c_extra and c_extra <= 16
    nt1, nt2 = (cout1 + 15) // 16, (cout2 + 15) // 16
    w = weight.reshape(cout2, cout1 + c_extra).float() * scale
    full = torch.zeros(nt2 * 16, (nt1 + 1) * 16, device=w.device, dtype=torch.float32)
    full[:cout2, :cout1] = w[:, :cout1]
    full[:cout2, nt1 * 16:nt1 * 16 + c_extra] = w[:, cout1:]
    # [t, j, n, q, e] -> [t, n, q, j, e]
    fr = full.view(nt2, 16, nt1 + 1, 4, 4).permute(0, 2, 3, 1, 4).contiguous().view(nt2, nt1 + 1, 64, 4)
    hi = fr.to(torch.bfloat16)
    lo = (fr - hi.float()).to(torch.bfloat16)
    wp = torch.stack([hi, lo], dim=2).contiguous()
    b = torch.zeros(nt2 * 16, device=w.device, dtype=torch.float32)
    if bias is not None:
        b[:cout2] = bias.float() * scale
    return wp, b


def pack_fpn_head_split(w_out, w_inner, b_inner):
    """Last head of the feature pyramid without its widest map (models/module.py:407-408):
        out3 = conv3x3_W(up2(top) + inner(l1)),   inner = 1x1 (W_in, b_in),   up2 = nearest x2,   W = ``w_out`` [Co, F, 3, 3], no bias.
    By linearity out3 = L + shuffle(U) with
      * L = conv3x3 over l1 with the composed weights  Wc[co, c, t] = sum_f W[co, f, t] W_in[f, c]   (full resolution, Ci -> Co);
      * U = conv3x3 over cat(top, ones) at HALF resolution with 4 Co outputs, channel (2 py + px) Co + co = output parity (py, px):
        a 3x3 window on the nearest-upsampled map touches 2x2 coarse pixels per parity, so the taps of W that land on the same
        coarse pixel are summed (py = 0: rows {-1} <- t0, {0} <- t1 + t2;  py = 1: {0} <- t0 + t1, {+1} <- t2; columns alike).
        The ones channel carries inner's bias: sum_f W[co, f, t] b_in[f] per tap, merged the same way -- zero padding of the
        coarse conv then reproduces exactly the taps that fall outside the map at full resolution.
    The F-channel full-resolution map (121 MB at 592x800) is neither written nor read.  Composition in fp64, result fp32.
    -> ((wU bf16x3 pack, bias), (wL bf16x3 pack, bias))"""
    co, f = w_out.shape[0], w_out.shape[1]
    ci = w_inner.shape[1]
    W = w_out.double()
    Win = w_inner.reshape(f, ci).double()
    wl = torch.einsum("oftu,fc->octu", W, Win)                                  # [Co, Ci, 3, 3]
    wb = torch.einsum("oftu,f->otu", W, b_inner.double()).unsqueeze(1)           # [Co, 1, 3, 3]: the bias as a ones channel
    wcat = torch.cat([W, wb], dim=1)                                            # [Co, F + 1, 3, 3] acting on up2(cat(top, ones))
    rows = {0: ((0,), (1, 2), ()), 1: ((), (0, 1), (2,))}                       # parity -> fine taps merged into coarse tap -1, 0, +1
    wu = torch.zeros(4 * co, f + 1, 3, 3, dtype=torch.float64, device=w_out.device)
    for py in (0, 1):
        for px in (0, 1):
            blk = wu[(2 * py + px) * co:(2 * py + px + 1) * co]
            for dy in range(3):
                for dx in range(3):
                    for ty in rows[py][dy]:
                        for tx in rows[px][dx]:
                            blk[:, :, dy, dx] += wcat[:, :, ty, tx]
    return pack_conv2d_bf16x3(wu.float(), None), pack_conv2d_bf16x3(wl.float(), None)


def pack_mask_taps_per_lane(weight, bias, cout1, scale=1.0):
    """The 1x1 convolution of the x2 mask head, [36, cout1, 1, 1] with channel 4*k + s = (tap k, sub-pixel s)
    (models/Effi_MVS_plus.py:170), for ``effi_conv2d_k3_k1_up2x_bf16x3_f32``: rows reordered so that MFMA row 16 t + 4 q + r is
    (tap 4 t + r, sub-pixel q) -- the lane quarter q then owns all nine taps of one sub-pixel; rows of taps 9..11 are zero."""
    assert weight.shape[0] == 36 and weight.shape[1] == cout1
    w = weight.reshape(36, cout1).float()
    wp = torch.zeros(48, cout1, device=w.device, dtype=torch.float32)
    bp = torch.zeros(48, device=w.device, dtype=torch.float32)
    for t in range(3):
        for q in range(4):
            for r in range(4):
                k = 4 * t + r
                if k < 9:
                    wp[16 * t + 4 * q + r] = w[4 * k + q]
                    if bias is not None:
                        bp[16 * t + 4 * q + r] = bias[4 * k + q].float()
    return pack_conv1x1_after(wp.view(48, cout1, 1, 1), bp, cout1, 0, scale)


def pack_head_taps(weight, cin):
    """conv2 of the depth head, [1, cin, 3, 3], as the 1x1 convolution [9, cin] of its nine taps (plane ky*3+kx), in the form
    ``pack_conv1x1_after`` gives the fused 3x3 -> 1x1 kernel; its bias is applied by ``ops.head_update``."""
    assert tuple(weight.shape) == (1, cin, 3, 3)
    taps = weight[0].reshape(cin, 9).t().contiguous().view(9, cin, 1, 1)
    return pack_conv1x1_after(taps, None, cin, 0)


class Conv2dWeights:
    """Both operand orders of one 2-D convolution: ``w32`` for the exact-fp32 MFMA kernel and, for 3x3 kernels with more
    than one output channel, ``wx`` for the split-bf16 kernel.  ``ops.conv2d`` picks per call (precision mode, shape)."""
    __slots__ = ("w32", "wx")

    def __init__(self, w32, wx):
        self.w32, self.wx = w32, wx


def pack_conv2d_k5s2_bf16x3(weight, bias, scale=1.0):
    """[cout, cin, 5, 5] (+bias) for ``effi_conv2d_k5s2_bf16x3_f32``: chunks of 8 input channels, K item = tap ky*5 + kx (25 items,
    7 K-steps), output tiles in groups of NT (1 if cout <= 16 else 2).  -> (bf16 [ceil(cin/8), G, 7, NT, 2(hi|lo), 64, 8],
    bias fp32 [16 * NT * G])."""
    cout, cin, ks, _ = weight.shape
    assert ks == 5
    nt = 1 if cout <= 16 else 2
    g = (cout + 16 * nt - 1) // (16 * nt)
    nch = (cin + 7) // 8
    w = torch.zeros(g * nt * 16, nch * 8, 28, device=weight.device, dtype=torch.float32)      # [co, ci, tap]
    w[:cout, :cin, :25] = weight.reshape(cout, cin, 25).float() * scale
    # [g, n, j, chunk, e, s, q] -> [chunk, g, s, n, q, j, e]
    w = w.view(g, nt, 16, nch, 8, 7, 4).permute(3, 0, 5, 1, 6, 2, 4).contiguous()
    hi = w.to(torch.bfloat16)
    lo = (w - hi.float()).to(torch.bfloat16)
    wp = torch.stack([hi, lo], dim=4).contiguous().view(nch, g, 7, nt, 2, 64, 8)
    b = torch.zeros(g * nt * 16, device=weight.device, dtype=torch.float32)
    if bias is not None:
        b[:cout] = bias.float() * scale
    return wp, b


def pack_conv2d(weight, bias, scale=1.0):
    """-> (Conv2dWeights, bias [16*NT]) for ``ops.conv2d``."""
    w32, b = pack_conv2d_mfma(weight, bias, scale)
    wx = None
    if weight.shape[2] == 3 and weight.shape[0] > 1:
        wx, _ = pack_conv2d_bf16x3(weight, bias, scale)
    elif weight.shape[2] == 5:
        wx, bx = pack_conv2d_k5s2_bf16x3(weight, bias, scale)      # the stride-2 pyramid layers in split precision
        if bx.numel() > b.numel():                                 # groups of two output tiles: the longer zero-padded bias serves both
            b = bx
    return Conv2dWeights(w32, wx), b


def pack_conv2d_c1k7(weight, bias):
    """[cout,1,7,7] -> ([49,cout], [cout])."""
    cout = weight.shape[0]
    return weight.reshape(cout, 49).t().contiguous().float(), bias.contiguous().float()


def pack_pixelwise_net(seq):
    """ConvBnReLU x3 + Conv2d(8,1,1) -> one fp32 block:
    w0[9][16] b0[16] | w1[16][9][16] b1[16] | w2[16][9][8] b2[8] | w3[8] b3[1]  (3x3 layers [cin][tap][cout])."""
    parts = []
    for i in range(3):
        conv, bn = seq[i].conv, seq[i].bn
        scale, shift = bn_scale_shift(bn)
        w = conv.weight * scale.view(-1, 1, 1, 1)                          # [cout,cin,3,3]
        cout, cin = w.shape[0], w.shape[1]
        parts.append(w.permute(1, 2, 3, 0).reshape(cin * 9 * cout))         # [cin][tap][cout]
        parts.append(shift.reshape(-1))
    parts.append(seq[3].weight.reshape(-1))
    parts.append(seq[3].bias.reshape(-1))
    blk = torch.cat([p.float() for p in parts]).contiguous()
    assert blk.numel() == 144 + 16 + 2304 + 16 + 1152 + 8 + 8 + 1, blk.numel()
    return blk
