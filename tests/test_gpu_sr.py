"""GPU: the split-resident ("SR") form of the update block (include/effi_mvs_hip.h, section "Split-resident activation maps";
reference: models/update.py:33-49,69-99,109-141) against the fp32-map form of the same kernels.

The bar is BITWISE: an SR map holds hi = bf16(x), lo = bf16(x - hi) of exactly the fp32 value the planar kernel writes, and an SR
consumer multiplies exactly the operands the planar consumer derives from that fp32 value -- so every layer, the whole GRU block
and the whole cascade must not move by one bit when the maps go split-resident.  (Parity of the block against the oracle and the
reference's golden vectors is what tests/test_gpu_kernels.py and tests/test_gpu_model.py check, with SR on by default.)
"""
import pytest
import torch

from common import build_model
from effi_mvs_plus_amd import synth

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

SIZES = [(20, 28), (37, 52), (64, 96), (148, 200)]        # odd heights, widths that are not tile multiples; 148x200 = cfg3 stage 1


def split_parts(x):
    hi = x.to(torch.bfloat16)
    lo = (x - hi.float()).to(torch.bfloat16)
    return hi.float(), lo.float()


def assert_sr_equals(m, x, what):
    """SR map ``m`` holds exactly the split of fp32 ``x`` and its border is zero."""
    hi, lo = m.parts()
    whi, wlo = split_parts(x)
    assert torch.equal(hi, whi), f"{what}: hi planes differ"
    assert torch.equal(lo, wlo), f"{what}: lo planes differ"
    full = m.t.float()
    full[:, :, 1:m.h + 1, 1:m.w + 1, :] = 0
    assert float(full.abs().max()) == 0.0, f"{what}: border written"


@pytest.fixture(scope="module")
def model():
    return build_model("8,8,8", seed=11, device=DEV)


@pytest.fixture
def split_precision():
    from effi_mvs_plus_amd import ops
    before = ops.get_precision()
    ops.set_precision("split")
    yield
    ops.set_precision(before)


@pytest.mark.parametrize("h,w", SIZES)
@pytest.mark.parametrize("stage", [0, 1, 2])
def test_every_layer_of_a_gru_iteration_is_bitwise_the_planar_layer(model, split_precision, stage, h, w):
    from effi_mvs_plus_amd import ops, packing
    from effi_mvs_plus_amd.models.update import _pack
    net, _ = model
    blk = net.update_block[stage]
    hd, cd = net.hdim_stage[stage], net.cdim_stage[stage]
    g = torch.Generator().manual_seed(100 * stage + h)
    rnd = lambda c: torch.randn(c, h, w, generator=g).to(DEV)
    e = blk.encoder
    # geometry / conversion / border
    assert ops.sr_geometry(h, w) == (((h + 15) // 16) * 16 + 2, ((w + 15) // 16) * 16 + 2)       # (w < 512: 16-column tiles only)
    assert ops.sr_geometry(592, 800) == (594, 834)
    maps = ops.sr_alloc(6, hd, h, w, DEV, clear=False)
    for m in maps:
        m.t.fill_(7.0)                           # poison: the border clear and the producers must overwrite what they own
    ops.sr_clear_border([maps])
    A, B, Cm, Dm, Hm, Xm = maps
    cor1, dfm1, hcur, ctx = rnd(hd), rnd(hd), torch.tanh(rnd(hd)), torch.relu(rnd(cd))
    for m, x in ((A, cor1), (B, dfm1), (Hm, hcur)):
        ops.sr_from_planar(x, out=m)
        assert_sr_equals(m, x, "sr_from_planar")
    # convc2 / convd2 (pair)
    wd2, bd2 = _pack(e._caches["d2"], e.convd2)
    wc2, bc2 = _pack(e._caches["c2"], e.convc2)
    cor, dfm = ops.conv2d_k3_bf16x3_pair([cor1], wc2.wx, bc2, [dfm1], wd2.wx, bd2, hd, act=ops.ACT_RELU)
    ops.conv2d_k3_pair_sr([A], wc2.wx, bc2, Cm, [B], wd2.wx, bd2, Dm, hd, act=ops.ACT_RELU)
    assert_sr_equals(Cm, cor, "convc2")
    assert_sr_equals(Dm, dfm, "convd2")
    # convd + convc (3x3 -> 1x1 with the context channels)
    wd, bd = _pack(e._caches["d"], e.convd)
    cmix = e.convd.out_channels
    wca, bca = packing.pack_conv1x1_after(e.convc.weight, e.convc.bias, cmix, cd)
    x = ops.conv2d_k3_k1_x3([cor, dfm], wd.wx, bd, cmix, ctx, wca, bca, hd, relu=True)
    ops.conv2d_k3_k1_sr([Cm, Dm], wd.wx, bd, cmix, ctx, wca, bca, hd, relu=True, out_sr=Xm)
    assert_sr_equals(Xm, x, "convd+convc")
    # ConvGRU: z / r*h, then the update
    wzr, bzr = blk.depth_gru._packed_zr()
    z, rh = ops.conv2d_k3_bf16x3([hcur, x], wzr.wx, bzr, 2 * hd, epilogue=ops.EPI_GRU_ZR, aux0=hcur)
    z2, _ = ops.conv2d_k3_sr([Hm, Xm], wzr.wx, bzr, 2 * hd, epilogue=ops.EPI_GRU_ZR, aux0=hcur, out_sr=B)
    assert torch.equal(z, z2), "z"
    assert_sr_equals(B, rh, "r*h")
    wq, bq = _pack(blk.depth_gru._cq, blk.depth_gru.convq)
    hn = ops.conv2d_k3_bf16x3([rh, x], wq.wx, bq, hd, epilogue=ops.EPI_GRU_Q, aux0=hcur, aux1=z)
    hn2, _ = ops.conv2d_k3_sr([B, Xm], wq.wx, bq, hd, epilogue=ops.EPI_GRU_Q, aux0=hcur, aux1=z, out_sr=Hm)   # in place over h
    assert torch.equal(hn, hn2), "new hidden state (fp32)"
    assert_sr_equals(Hm, hn, "new hidden state (SR)")
    # depth head: conv1 + tap projections (fp32 output from SR input)
    dh = blk.depth_head
    wh1, bh1 = _pack(dh._c1, dh.conv1)
    wh2, bh2 = packing.pack_head_taps(dh.conv2.weight, hd)
    p1 = ops.conv2d_k3_k1_x3([hn], wh1.wx, bh1, hd, None, wh2, bh2, 9, relu=False, relu1=True)
    p2 = ops.conv2d_k3_k1_sr([Hm], wh1.wx, bh1, hd, None, wh2, bh2, 9, relu=False, relu1=True)
    assert torch.equal(p1, p2), "depth-head tap projections"
    # mask head + convex upsampling
    inv = torch.rand(1, h, w, generator=g).to(DEV)
    dr = torch.linspace(1 / 935.0, 1 / 425.0, 384).to(DEV)
    wm, bm = _pack(blk._m0, blk.mask[0])
    c1 = blk.mask[0].out_channels
    w2, b2 = packing.pack_mask_taps_per_lane(blk.mask[2].weight, blk.mask[2].bias, c1, scale=0.25)
    u1 = ops.conv2d_k3_k1_up2x([hn], wm.wx, bm, c1, w2, b2, inv, dr)
    u2 = ops.conv2d_k3_k1_up2x_sr([Hm], wm.wx, bm, c1, w2, b2, inv, dr)
    assert torch.equal(u1[0], u2[0]) and torch.equal(u1[1], u2[1]), "mask head + upsampling"
    # PLAIN epilogue with both outputs and a non-ReLU activation
    y = ops.conv2d_k3_bf16x3([cor1], wc2.wx, bc2, hd, act=ops.ACT_TANH)
    y2, ysr = ops.conv2d_k3_sr([A], wc2.wx, bc2, hd, act=ops.ACT_TANH, out0=torch.empty_like(y), out_sr=Cm)
    assert torch.equal(y, y2)
    assert_sr_equals(ysr, y, "plain conv, tanh")


@pytest.mark.parametrize("h,w", [(20, 28), (74, 100)])
def test_initial_states_and_encoder_inputs_in_sr_form(model, split_precision, h, w):
    from effi_mvs_plus_amd import ops
    net, _ = model
    g = torch.Generator().manual_seed(5)
    hds, cds = net.hdim_stage, net.cdim_stage
    ctxs = [torch.randn(hd + cd, h * 2 ** s, w * 2 ** s, generator=g).to(DEV) for s, (hd, cd) in enumerate(zip(hds, cds))]
    want = ops.split_tanh_relu_stages(ctxs, hds, cds)
    blocks = [ops.sr_alloc(2, hd, c.shape[1], c.shape[2], DEV, clear=False) for hd, c in zip(hds, ctxs)]
    for b in blocks:
        for m in b:
            m.t.fill_(3.0)
    ops.sr_clear_border(blocks)                                   # three geometries, one launch
    got = ops.split_tanh_relu_stages_sr(ctxs, hds, cds, [b[1] for b in blocks])
    for (wh, wi), (gh, gi), b in zip(want, got, blocks):
        assert torch.equal(wh, gh) and torch.equal(wi, gi)
        assert_sr_equals(b[1], wh, "initial hidden state")
    # encoder inputs (lookup + 1x1, 7x7) for every stage's channel count
    D = 8
    for s, hd in enumerate(hds):
        hh, ww = ctxs[s].shape[1:]
        blk = net.update_block[s]
        wc1, bc1 = blk.encoder.convc1_raw()
        w7, b7 = blk.encoder.conv7_packed()
        inv = torch.rand(1, hh, ww, generator=g).to(DEV)
        cur, reg = torch.randn(D, hh, ww, generator=g).to(DEV), torch.randn(D, hh, ww, generator=g).to(DEV)
        dr = torch.linspace(1 / 935.0, 1 / 425.0, 384).to(DEV)
        itv = torch.tensor([(dr[-1] - dr[0]).item() / 384], device=DEV)
        lo, hi = torch.full((1,), 425.0, device=DEV), torch.full((1,), 935.0, device=DEV)
        c1, d1 = ops.encoder_inputs(inv, dr, itv, cur, reg, lo, hi, 3, hh, ww, wc1, bc1, w7, b7, hd)
        a, b_ = blocks[s]
        ops.encoder_inputs_sr(inv, dr, itv, cur, reg, lo, hi, 3, hh, ww, wc1, bc1, w7, b7, hd, a, b_)
        assert_sr_equals(a, c1, f"relu(convc1(cost)) stage {s + 1}")
        assert_sr_equals(b_, d1, f"relu(convd1(inv)) stage {s + 1}")


@pytest.mark.parametrize("precision", ["split", "bf16"])
@pytest.mark.parametrize("h,w,force_mr", [(20, 28, None), (37, 52, 2), (37, 52, 4), (74, 100, 1), (148, 200, None), (100, 528, None)])
def test_generated_encoder_pair_is_bitwise_the_two_launches(model, precision, h, w, force_mr):
    """``encoder_pair_gen_sr`` (cor1 / dfm1 generated inside the convc2 | convd2 kernel) against ``encoder_inputs_sr`` +
    ``conv2d_k3_pair_sr``: every bit of both output maps, for the three stages' channel counts, every tile shape (100x528: the
    4 x 64 tiles of the 592x800 stage), map edges that cut tiles, queries that leave the volume."""
    from effi_mvs_plus_amd import ops
    from effi_mvs_plus_amd.models.update import _pack
    net, _ = model
    g = torch.Generator().manual_seed(7 * h + w)
    D = 8
    before = ops.get_precision()
    try:
        ops.set_precision(precision)
        for s, hd in enumerate(net.hdim_stage):
            e = net.update_block[s].encoder
            wc1, bc1 = e.convc1_raw()
            w7, b7 = e.conv7_packed()
            wd2, bd2 = _pack(e._caches["d2"], e.convd2)
            wc2, bc2 = _pack(e._caches["c2"], e.convc2)
            inv = (torch.rand(1, h, w, generator=g) * 1.2 - 0.1).to(DEV)          # some estimates outside [0, 1]: taps off the volume
            cur, reg = torch.randn(D, h, w, generator=g).to(DEV), torch.randn(D, h, w, generator=g).to(DEV)
            dr = torch.linspace(1 / 935.0, 1 / 425.0, 384).to(DEV)
            itv = torch.tensor([(dr[-1] - dr[0]).item() / 384], device=DEV)
            lo, hi = torch.full((1,), 425.0, device=DEV), torch.full((1,), 935.0, device=DEV)
            maps = ops.sr_alloc(6, hd, h, w, DEV, clear=False)
            for m in maps:
                m.t.fill_(5.0)
            ops.sr_clear_border([maps])
            A, B, C1, D1, C2, D2 = maps
            # (the kernel takes 3 rows per wave where the rule says 4; enc_gen_mr3 = 0 keeps the 4-row instantiation reachable: tested at 37x52)
            with ops.options(force_mr=force_mr, enc_gen_mr3=0 if (h, force_mr) == (37, 4) and s == 2 else None):
                ops.encoder_inputs_sr(inv, dr, itv, cur, reg, lo, hi, 3, h, w, wc1, bc1, w7, b7, hd, A, B)
                ops.conv2d_k3_pair_sr([A], wc2.wx, bc2, C1, [B], wd2.wx, bd2, D1, hd, act=ops.ACT_RELU)
                ops.encoder_pair_gen_sr(inv, dr, itv, cur, reg, lo, hi, 3, h, w, wc1, bc1, w7, b7, hd, wc2.wx, bc2, C2, wd2.wx, bd2, D2, hd,
                                        act=ops.ACT_RELU)
            torch.cuda.synchronize()
            assert float(C1.t.float().abs().max()) > 0 and float(D1.t.float().abs().max()) > 0
            assert torch.equal(C1.t, C2.t), f"stage {s + 1} (hd {hd}): relu(convc2(cor1)) differs"
            assert torch.equal(D1.t, D2.t), f"stage {s + 1} (hd {hd}): relu(convd2(dfm1)) differs"
    finally:
        ops.set_precision(before)


@pytest.mark.parametrize("precision", ["split", "bf16"])
@pytest.mark.parametrize("h,w", [(20, 28), (37, 52), (74, 100), (148, 200), (29, 15)])
def test_fused_conv_gru_is_bitwise_the_two_launches(model, precision, h, w):
    """``gru_zr_q_fused_sr`` (z | r convolution, r * h, q convolution and the state update in one kernel, r * h and z on chip) against
    ``conv2d_k3_sr(GRU_ZR)`` + ``conv2d_k3_sr(GRU_Q)``: every bit of the new state in both forms (fp32 map, split-resident map incl.
    its untouched zero border), for hd = 32 and 16, map sizes that cut the 14 x 14 output tiles everywhere."""
    from effi_mvs_plus_amd import ops
    from effi_mvs_plus_amd.models.update import _pack
    net, _ = model
    g = torch.Generator().manual_seed(3 * h + w)
    before = ops.get_precision()
    try:
        ops.set_precision(precision)
        for s in (1, 2):
            hd = net.hdim_stage[s]
            gru = net.update_block[s].depth_gru
            wzr, bzr = gru._packed_zr()
            wq, bq = _pack(gru._cq, gru.convq)
            hcur = torch.tanh(torch.randn(hd, h, w, generator=g)).to(DEV)
            x = torch.relu(torch.randn(hd, h, w, generator=g)).to(DEV)
            maps = ops.sr_alloc(5, hd, h, w, DEV, clear=False)
            for m in maps:
                m.t.fill_(9.0)
            ops.sr_clear_border([maps])
            Hm, Xm, RH, H1, H2 = maps
            ops.sr_from_planar(hcur, out=Hm)
            ops.sr_from_planar(x, out=Xm)
            z, _ = ops.conv2d_k3_sr([Hm, Xm], wzr.wx, bzr, 2 * hd, epilogue=ops.EPI_GRU_ZR, aux0=hcur, out_sr=RH)
            want, _ = ops.conv2d_k3_sr([RH, Xm], wq.wx, bq, hd, epilogue=ops.EPI_GRU_Q, aux0=hcur, aux1=z, out_sr=H1)
            got = torch.full_like(hcur, 7.0)
            ops.gru_zr_q_fused_sr(Hm, Xm, hcur, wzr.wx, bzr, wq.wx, bq, got, H2)
            torch.cuda.synchronize()
            assert torch.equal(got, want), f"hd {hd}: fp32 state differs ({int((got != want).sum())} values)"
            assert torch.equal(H1.t, H2.t), f"hd {hd}: split-resident state differs"
            with pytest.raises(ValueError):
                ops.gru_zr_q_fused_sr(Hm, Xm, hcur, wzr.wx, bzr, wq.wx, bq, hcur, H2)          # in place: refused
    finally:
        ops.set_precision(before)


@pytest.mark.parametrize("H,W,N,nd", [(128, 160, 4, "8,8,8"), (192, 256, 5, "48,8,8"), (320, 416, 3, "48,32,8")])
@pytest.mark.parametrize("precision", ["split", "bf16"])
def test_cascade_is_bitwise_unchanged_by_split_resident_maps(H, W, N, nd, precision):
    """The whole hot path (13 depth maps + confidence) with the update blocks on SR maps vs on fp32 maps."""
    from effi_mvs_plus_amd import ops
    net, _ = build_model(nd, seed=3, device=DEV)
    imgs, pm, dv = synth.synth_sample(H, W, N, seed=2)
    imgs = imgs.to(DEV)
    pm = {k: v.to(DEV) for k, v in pm.items()}
    dv = dv.to(DEV)
    before = ops.get_precision()
    try:
        ops.set_precision(precision)
        with torch.no_grad():
            feats = [net.feature(imgs[:, v]) for v in range(N)]
            ctx = net.cnet_depth(imgs[:, 0])
            ops.set_sr(False)
            want = net.forward_hot(feats, ctx, pm, dv)
            want_inter = net.forward_hot(feats, ctx, pm, dv, want_intermediates=True)
            ops.set_sr(True)
            assert ops.uses_sr()
            got = net.forward_hot(feats, ctx, pm, dv)
            inter = net.forward_hot(feats, ctx, pm, dv, want_intermediates=True)       # unfused mask head on the fp32 state
            with ops.options(gru_fused=2):                                              # ConvGRU as one launch at hd 16 and 32 (option; default off)
                got_f = net.forward_hot(feats, ctx, pm, dv)
            with ops.options(enc_gen=0):                                                # encoder inputs as maps instead of generated in the pair kernel
                got_e = net.forward_hot(feats, ctx, pm, dv)
    finally:
        ops.set_precision(before)
        ops.set_sr(True)
    for i, (a, b) in enumerate(zip(got["depth"], want["depth"])):
        assert torch.equal(a, b), f"depth map {i} differs"
        assert torch.equal(got_f["depth"][i], b), f"depth map {i} differs with the one-launch ConvGRU"
        assert torch.equal(got_e["depth"][i], b), f"depth map {i} differs with encoder-input maps"
    assert torch.equal(got["photometric_confidence"], want["photometric_confidence"])
    for a, b in zip(inter["depth"], want_inter["depth"]):     # (the unfused mask head evaluates its softmax with expf: its own pair)
        assert torch.equal(a, b)


def test_update_block_module_forward_takes_the_sr_path_and_matches(model, split_precision):
    """``BasicUpdateBlock.forward`` with the reference's calling convention (fp32 hidden state in, partial(GetCost), partial(disp_to_depth)):
    the block converts the caller's state at its boundary; result bitwise equal to the fp32-map form."""
    import functools
    from effi_mvs_plus_amd import ops
    from effi_mvs_plus_amd.models.Effi_MVS_plus import disp_to_depth
    net, _ = model
    s, h, w, D = 1, 40, 56, 8
    hd, cd = net.hdim_stage[s], net.cdim_stage[s]
    g = torch.Generator().manual_seed(9)
    hid = torch.tanh(torch.randn(1, hd, h, w, generator=g)).to(DEV)
    ctx = torch.relu(torch.randn(1, cd, h, w, generator=g)).to(DEV)
    inv = torch.rand(1, 1, h, w, generator=g).to(DEV)
    dv = torch.linspace(1 / 935.0, 1 / 425.0, 384).unsqueeze(0).to(DEV)
    vols = [torch.randn(1, D, h, w, generator=g).to(DEV) for _ in range(2)]
    pro = [v.permute(0, 2, 3, 1).reshape(h * w, 1, 1, D) for v in vols]
    dmin, dmax = torch.full((1, 1, 1, 1), 425.0, device=DEV), torch.full((1, 1, 1, 1), 935.0, device=DEV)
    itv = ((dv[:, -1] - dv[:, 0]) / 384).view(1, 1, 1, 1)
    cost = functools.partial(net.GetCost, pro=pro, features=None, proj_matrices=None, depth_interval=itv, depth_max=None, depth_min=None,
                             view_weights=None, CostNum=3, Inverse=True, G=1, depth_max_cur_volume=dmax, depth_min_cur_volume=dmin)
    scale = functools.partial(disp_to_depth, min_depth=1. / dv[:, -1, None, None, None], max_depth=1. / dv[:, 0, None, None, None])
    scale.effi_disp_range = dv
    outs = {}
    try:
        for sr in (False, True):
            ops.set_sr(sr)
            with torch.no_grad():
                outs[sr] = net.update_block[s](hid, cost, inv, ctx, seq_len=3, scale_inv_depth=scale)
    finally:
        ops.set_sr(True)
    n0, m0, i0 = outs[False]
    n1, m1, i1 = outs[True]
    assert torch.equal(n0, n1)
    for a, b in zip(m0 + i0, m1 + i1):
        assert torch.equal(a, b)


def test_sr_entries_reject_bad_geometry(split_precision):
    from effi_mvs_plus_amd import ops
    from effi_mvs_plus_amd._lib import EffiLibraryError
    m = ops.sr_alloc(1, 16, 20, 28, DEV)[0]
    bad = ops.SRMap(torch.zeros(2, 2, 22, 30, 8, device=DEV, dtype=torch.bfloat16), 16, 20, 28)    # too small for the tile overhang
    wgt = torch.zeros(5 * 1 * 2 * 64 * 8, device=DEV, dtype=torch.bfloat16)
    with pytest.raises(EffiLibraryError):
        ops.conv2d_k3_sr([bad], wgt, torch.zeros(16, device=DEV), 16, out_sr=bad)
    with pytest.raises(ValueError):
        ops.conv2d_k3_sr([ops.SRMap(m.t[:1], 8, 20, 28)], wgt, torch.zeros(16, device=DEV), 16, out_sr=m)   # 8 channels: not a whole chunk



@pytest.mark.parametrize("hd,h,w", [(16, 40, 72), (32, 24, 36), (48, 20, 28), (16, 70, 528)])
def test_q4_state_layout_is_bitwise_the_planar_one(hd, h, w):
    """EFFI_EPI_Q4: the GRU epilogues with the fp32 state h, the gate z and the new state laid out [hd/4][h][w][4] (one 16-byte
    access per lane and map) against the planar maps -- same values bit for bit, both in the fp32 and the split-resident outputs;
    and split_tanh_relu_stages_sr writing the initial state in that layout (models/update.py:40-49, models/Effi_MVS_plus.py:442-452)."""
    from effi_mvs_plus_amd import ops, packing
    before = ops.get_precision()
    ops.set_precision("split")
    g = torch.Generator().manual_seed(hd + h)
    rnd = lambda *sh: torch.randn(*sh, generator=g).to(DEV)
    hcur, x, rh, z = rnd(hd, h, w), rnd(hd, h, w), rnd(hd, h, w), torch.rand(hd, h, w, generator=g).to(DEV)
    H, X, RH = (ops.sr_alloc(1, hd, h, w, DEV)[0] for _ in range(3))
    for m, t_ in ((H, hcur), (X, x), (RH, rh)):
        ops.sr_from_planar(t_, out=m)
    wzr_x, bzr = packing.pack_conv2d_bf16x3(rnd(2 * hd, 2 * hd, 3, 3) * 0.05, rnd(2 * hd) * 0.1)
    wq_x, bq = packing.pack_conv2d_bf16x3(rnd(hd, 2 * hd, 3, 3) * 0.05, rnd(hd) * 0.1)
    # z | r
    z_p, rh_p = ops.conv2d_k3_sr([H, X], wzr_x, bzr, 2 * hd, epilogue=ops.EPI_GRU_ZR, aux0=hcur)
    z_q, rh_q = ops.conv2d_k3_sr([H, X], wzr_x, bzr, 2 * hd, epilogue=ops.EPI_GRU_ZR, aux0=ops.q4_from_planar(hcur).view(hd, h, w), q4=True)
    assert torch.equal(ops.q4_to_planar(z_q.view(hd // 4, h, w, 4), hd), z_p) and torch.equal(rh_q.t, rh_p.t)
    # q + update
    h_p, H_p = ops.conv2d_k3_sr([RH, X], wq_x, bq, hd, epilogue=ops.EPI_GRU_Q, aux0=hcur, aux1=z)
    h_q, H_q = ops.conv2d_k3_sr([RH, X], wq_x, bq, hd, epilogue=ops.EPI_GRU_Q, aux0=ops.q4_from_planar(hcur).view(hd, h, w),
                                aux1=ops.q4_from_planar(z).view(hd, h, w), q4=True)
    assert torch.equal(ops.q4_to_planar(h_q.view(hd // 4, h, w, 4), hd), h_p) and torch.equal(H_q.t, H_p.t)
    assert torch.isfinite(h_p).all() and float(h_p.abs().max()) > 0.1
    # initial state
    cd = 4
    ctx = rnd(hd + cd, h, w)
    m1, m2 = ops.sr_alloc(1, hd, h, w, DEV)[0], ops.sr_alloc(1, hd, h, w, DEV)[0]
    (hid_p, inp_p), = ops.split_tanh_relu_stages_sr([ctx], [hd], [cd], [m1])
    (hid_q, inp_q), = ops.split_tanh_relu_stages_sr([ctx], [hd], [cd], [m2], q4=[True])
    assert torch.equal(ops.q4_to_planar(hid_q.view(hd // 4, h, w, 4), hd), hid_p) and torch.equal(inp_q, inp_p) and torch.equal(m1.t, m2.t)
    ops.set_precision(before)
