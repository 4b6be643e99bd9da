"""CPU: host-side logic of the product -- interface parity (state-dict keys, signatures), the C-ABI
library loads and exports every declared symbol, weight packing layouts, synthetic rig, loud failure
without a GPU.  No compute calls into the library here."""
import ctypes
import inspect
import json
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from common import GOLDEN, build_model
from effi_mvs_plus_amd import _lib, packing, synth


def test_state_dict_keys_match_the_reference():
    with open(os.path.join(GOLDEN, "state_dict_keys.json")) as f:
        want = json.load(f)
    net, _ = build_model("48,8,8")
    got = {k: [list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in net.state_dict().items()}
    assert got == want
    assert len(got) == 563
    # the aliases the reference creates by registering modules twice
    assert net.update_block[1] is net.update_block_depth2 and net.CSP_R[0] is net.CSP_R1 and net.CSP_C[1] is net.CSP_C2


def test_public_signatures_match_the_reference():
    import sys
    import effi_mvs_plus_amd.models as M
    main = sys.modules["effi_mvs_plus_amd.models.Effi_MVS_plus"]
    from effi_mvs_plus_amd.models import module, update
    assert {"Effi_MVS_plus", "mvs_loss"} <= set(dir(M))

    def params(f):
        return [p for p in inspect.signature(f).parameters if p != "self"]

    assert params(module.homo_warping_new) == ["src_fea", "src_proj", "ref_proj", "depth_values"]
    assert params(main.DepthNet.forward) == ["features", "proj_matrices", "depth_values", "num_depth", "cost_regularization",
                                             "pixel_wise_net", "G"]
    assert params(main.GetCost_initvolume.forward) == ["depth_values", "features", "proj_matrices", "depth_interval", "depth_max",
                                                       "depth_min", "view_weights", "CostNum", "Inverse", "G", "iter", "inter_iter"]
    assert params(main.GetCost.forward)[:15] == ["depth_values", "pro", "features", "proj_matrices", "depth_interval", "depth_max",
                                                 "depth_min", "view_weights", "CostNum", "Inverse", "G", "depth_max_cur_volume",
                                                 "depth_min_cur_volume", "iter", "inter_iter"]
    assert params(update.BasicUpdateBlock.__init__) == ["hidden_dim", "cost_dim", "ratio", "context_dim", "UpMask", "Inverse",
                                                        "cost_num", "G"]
    assert params(update.BasicUpdateBlock.forward) == ["net", "depth_cost_func", "inv_depth", "context", "seq_len", "scale_inv_depth"]
    assert params(update.ProjectionInput.forward) == ["disp", "cost", "context"]
    assert params(update.DepthHead.forward) == ["x_d", "act_fn"]
    assert params(module.cost_up_small.forward) == ["x", "IGEV_cost"]
    assert params(main.Effi_MVS_plus.__init__) == ["args", "refine", "ndepths", "depth_interals_ratio", "share_cr", "CostNum",
                                                   "inverse", "stage_channel"]
    assert params(main.Effi_MVS_plus.forward) == ["imgs", "proj_matrices", "depth_values"]
    assert params(main.pro_bilinear_sampler) == ["pro", "depth_sample", "depth_min", "depth_max"]
    assert params(main.upsample_depth) == ["depth", "mask", "ratio"]
    assert params(main.bilinear_sampler) == ["img", "coords", "mode", "mask"]


def test_library_exports_every_declared_symbol():
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    handle = ctypes.CDLL(_lib.LIB_PATH)                 # loads without a GPU
    declared = _lib.declared_symbols()
    assert len(declared) >= 22
    for name in declared:
        assert hasattr(handle, name), f"{name} declared in include/effi_mvs_hip.h but not exported"
    assert set(_lib.SIGNATURES) | set(_lib.NON_STATUS_SYMBOLS) == set(declared)
    handle.effi_version.restype = ctypes.c_int
    assert handle.effi_version() >= 100
    handle.effi_error_string.restype = ctypes.c_char_p
    assert b"bad argument" in handle.effi_error_string(-1)
    # caller-owned per-device workspace registry (no GPU needed: it only stores the pointer)
    handle.effi_workspace_bytes.restype = ctypes.c_long
    handle.effi_get_workspace.restype = ctypes.c_void_p
    handle.effi_set_workspace.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_long]
    need = handle.effi_workspace_bytes()
    assert 64 <= need <= 1 << 20
    assert handle.effi_get_workspace(3) is None
    assert handle.effi_set_workspace(3, ctypes.c_void_p(0x1000), need - 1) == -1          # too small
    assert handle.effi_set_workspace(99, ctypes.c_void_p(0x1000), need) == -1             # no such ordinal
    assert handle.effi_set_workspace(3, ctypes.c_void_p(0x1000), need) == 0 and handle.effi_get_workspace(3) == 0x1000
    assert handle.effi_get_workspace(2) is None                                           # slots are per device
    assert handle.effi_set_workspace(3, None, 0) == 0 and handle.effi_get_workspace(3) is None
    assert b"workspace" in handle.effi_error_string(-4)


def test_product_fails_loudly_without_gpu():
    from effi_mvs_plus_amd._lib import EffiLibraryError
    net, _ = build_model("8,8,8")
    imgs, pm, dv = synth.synth_sample(64, 96, 3, seed=0)
    with pytest.raises(EffiLibraryError):                # CPU tensors: no fallback
        net(imgs, pm, dv)
    net.train()
    with pytest.raises(EffiLibraryError):                # training mode takes the differentiable HIP path: no CPU fallback either
        net(imgs, pm, dv)
    with pytest.raises(NotImplementedError):             # the fused single-sample launches are inference only
        net.cost_regularization.run(torch.zeros(1, 8, 8, 8))


def test_product_never_imports_the_oracle():
    import re
    root = os.path.join(os.path.dirname(GOLDEN), "..", "effi_mvs_plus_amd")
    for dirpath, _, files in os.walk(root):
        for fn in files:
            if fn.endswith(".py"):
                text = open(os.path.join(dirpath, fn)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", text, flags=re.M), fn


def _emulate_mfma_conv(x, wpack, bias, cout, ks):
    """Reference of what conv2d_mfma_*_kernel computes from the PACKED weights (numpy, tiny sizes):
    B fragment lane = k*16 + j holds W[cout = 16n + j][cin = 4g + k] for tap t."""
    cin, h, w = x.shape
    kg, taps, nt, _ = wpack.shape
    r = ks // 2
    xp = np.zeros((kg * 4, h + 2 * r, w + 2 * r), np.float64)
    xp[:cin, r:r + h, r:r + w] = x
    out = np.zeros((nt * 16, h, w), np.float64)
    wp = wpack.reshape(kg, taps, nt, 4, 16).astype(np.float64)
    for g in range(kg):
        for t in range(taps):
            ky, kx = divmod(t, ks)
            patch = xp[4 * g:4 * g + 4, ky:ky + h, kx:kx + w]                # [k, h, w]
            out += np.einsum("khw,nkj->njhw", patch, wp[g, t]).reshape(nt * 16, h, w)
    return (out + bias.astype(np.float64)[:, None, None])[:cout]


@pytest.mark.parametrize("ks,cin,cout", [(3, 5, 7), (1, 6, 48), (3, 32, 36), (3, 16, 1)])
def test_conv2d_weight_packing_layout(ks, cin, cout):
    g = torch.Generator().manual_seed(ks * 100 + cin)
    wt, b = torch.randn(cout, cin, ks, ks, generator=g), torch.randn(cout, generator=g)
    x = torch.randn(cin, 6, 9, generator=g)
    wp, bp = packing.pack_conv2d_mfma(wt, b)
    shape4 = ((cin + 3) // 4, ks * ks, (cout + 15) // 16, 64)
    if cout == 1 and ks == 3:          # depth head: plain [cin][9] weights ride behind the MFMA block
        n = shape4[0] * shape4[1] * shape4[2] * 64
        assert wp.dim() == 1 and wp.numel() == n + cin * 9
        assert torch.equal(wp[n:].view(cin, 9), wt.reshape(cin, 9))
        wp = wp[:n].view(shape4)
    assert wp.shape == shape4 and bp.shape == (16 * ((cout + 15) // 16),)
    got = _emulate_mfma_conv(x.numpy(), wp.numpy(), bp.numpy(), cout, ks)
    want = F.conv2d(x.unsqueeze(0).double(), wt.double(), b.double(), padding=ks // 2)[0].numpy()
    assert np.allclose(got, want, rtol=1e-6, atol=1e-6)
    wq, bq = packing.pack_conv2d_mfma(wt, b, scale=0.25)              # mask head: exact power-of-two folding
    assert torch.equal(wq.reshape(-1)[:wp.numel()], wp.reshape(-1) * 0.25) and torch.equal(bq, bp * 0.25)


def test_bn_folding_and_3d_packing():
    from effi_mvs_plus_amd.models.module import Conv3d, Deconv3d
    g = torch.Generator().manual_seed(0)
    for mod, transposed in ((Conv3d(4, 8, stride=2, padding=1), False),
                            (Deconv3d(8, 4, stride=2, padding=1, output_padding=1), True)):
        mod.eval()
        mod.bn.running_mean.data = torch.randn(mod.bn.running_mean.shape, generator=g)
        mod.bn.running_var.data = torch.rand(mod.bn.running_var.shape, generator=g) + 0.5
        mod.bn.weight.data = torch.rand(mod.bn.weight.shape, generator=g) + 0.5
        mod.bn.bias.data = torch.randn(mod.bn.bias.shape, generator=g)
        wp, bp = (packing.pack_deconv3d if transposed else packing.pack_conv3d)(mod.conv, mod.bn)
        cin = mod.conv.in_channels
        cout = mod.conv.out_channels
        assert wp.shape == (cin, 27, cout)
        # unpack and run through torch: must equal conv -> BN (eval)
        w5 = wp.view(cin, 3, 3, 3, cout)
        x = torch.randn(1, cin, 4, 6, 6, generator=g)
        if transposed:
            y = F.conv_transpose3d(x, w5.permute(0, 4, 1, 2, 3), bp, stride=2, padding=1, output_padding=1)
        else:
            y = F.conv3d(x, w5.permute(4, 0, 1, 2, 3), bp, stride=2, padding=1)
        assert torch.allclose(y, mod.bn(mod.conv(x)), rtol=1e-4, atol=1e-5)
    # cache invalidation: in-place edits and load_state_dict give a new packing
    c = Conv3d(1, 8, padding=1).eval()
    a = c._packed()[0]
    assert c._packed()[0] is a
    c.conv.weight.data.mul_(2.0)
    c.conv.weight._version  # noqa: B018
    with torch.no_grad():
        c.conv.weight.mul_(1.0)
    assert c._packed()[0] is not a


def test_synthetic_rig_is_deterministic_and_well_posed():
    a = synth.synth_sample(64, 96, 5, seed=3)
    b = synth.synth_sample(64, 96, 5, seed=3)
    assert torch.equal(a[0], b[0]) and torch.equal(a[2], b[2])
    assert a[2].shape == (1, 384) and (a[2][0, 1:] > a[2][0, :-1]).all()           # ascending inverse depths
    pm = a[1]
    assert set(pm) == {"stage0", "stage1", "stage2", "stage3", "stage4"} and pm["stage1"].shape == (1, 5, 2, 4, 4)
    assert torch.allclose(pm["stage2"][0, 0, 1, :2, :3], 2 * pm["stage1"][0, 0, 1, :2, :3])
    sd1 = synth.randomize_state_dict({"a.conv.weight": torch.empty(4, 3, 3, 3), "update_block.0.x.bias": torch.empty(5),
                                      "update_block_depth1.x.bias": torch.empty(5)}, seed=1)
    assert torch.equal(sd1["update_block.0.x.bias"], sd1["update_block_depth1.x.bias"])   # aliases get equal values


def test_static_loss_drops_non_finite_estimates_at_masked_pixels_like_the_reference():
    """``mvs_loss_static`` (the capturable form of models/module.py:526-552) against ``mvs_loss``: depth is 1 / inv_depth and can be
    Inf where inv_depth is 0; the reference's boolean indexing drops masked pixels, a product with a 0 / 1 mask would turn them into
    NaN.  Value and gradient must agree, with Inf / NaN sitting at masked pixels."""
    import torch
    from effi_mvs_plus_amd.models.module import mvs_loss, mvs_loss_static
    torch.manual_seed(0)
    ins = [torch.rand(2, 12, 10, requires_grad=True) for _ in range(4)]
    gt = {"stage1": torch.rand(2, 12, 10), "stage2": torch.rand(2, 12, 10)}
    mask = {k: (torch.rand(2, 12, 10) > 0.4).float() for k in gt}
    with torch.no_grad():
        ins[0][0, 0, 0], ins[1][1, 3, 4], ins[3][0, 5, 5] = float("inf"), float("nan"), -float("inf")
    mask["stage1"][0, 0, 0] = mask["stage1"][1, 3, 4] = mask["stage2"][0, 5, 5] = 0
    dl = (1, 1, 2, 2)
    a, pa = mvs_loss_static(ins, gt, mask, dl)
    b, pb = mvs_loss(ins, gt, mask, dl)
    assert torch.isfinite(a) and abs(float(a) - float(b)) <= 1e-6 * abs(float(b))
    assert all(abs(float(pa[k]) - float(pb[k])) <= 1e-6 for k in pb)
    a.backward()
    ga = [t.grad.clone() for t in ins]
    for t in ins:
        t.grad = None
    b.backward()
    for x, t in zip(ga, ins):
        assert torch.isfinite(x).all() and torch.allclose(x, t.grad, rtol=1e-6, atol=1e-9)

