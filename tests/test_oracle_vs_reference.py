"""CPU, build container only: the oracle against the REFERENCE ITSELF (imported from /root/reference).
Skipped where the reference tree is absent (the GPU box).  This is what pins the oracle: the reference
ships no tests or golden vectors of its own."""
import os

import pytest
import torch

from refimport import REF_ROOT, reference_model
from effi_mvs_plus_amd import synth
from oracle import effi_oracle as O

pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF_ROOT, "models")), reason="reference tree not present")


def _compare(net, sd, H, W, N, nd):
    imgs, pm, dv = synth.synth_sample(H, W, N, seed=5)
    with torch.no_grad():
        want = net(imgs, pm, dv)
        got = O.full_forward(sd, imgs, pm, dv, ndepths=nd)
    assert len(want["depth"]) == len(got["depth"]) == 13
    for a, b in zip(got["depth"], want["depth"]):
        assert torch.equal(a, b)                      # same op sequence on the same CPU: bitwise
    assert torch.equal(got["photometric_confidence"], want["photometric_confidence"])


def test_random_weights_bitwise():
    _, net = reference_model("48,8,8")
    sd = synth.randomize_state_dict(net.state_dict(), seed=3)
    net.load_state_dict(sd, strict=True)
    _compare(net, sd, 128, 160, 4, (48, 8, 8))


def test_other_hypothesis_counts_bitwise():
    _, net = reference_model("48,32,8")               # BASELINE.json's wording of the cascade
    sd = synth.randomize_state_dict(net.state_dict(), seed=4)
    net.load_state_dict(sd, strict=True)
    _compare(net, sd, 96, 128, 3, (48, 32, 8))


def test_shipped_checkpoint_bitwise():
    ckpt = os.path.join(REF_ROOT, "checkpoints", "Effi_MVS_plus", "model_dtu.ckpt")
    if not os.path.exists(ckpt):
        pytest.skip("shipped checkpoint not present")
    _, net = reference_model("48,8,8")
    sd = torch.load(ckpt, map_location="cpu", weights_only=True)["model"]
    net.load_state_dict(sd, strict=True)
    _compare(net, sd, 192, 256, 5, (48, 8, 8))


def test_depthnet_without_view_weight_net_bitwise():
    """The reference's ``pixel_wise_net=None`` branch of DepthNet (models/Effi_MVS_plus.py:55-58,70), which the shipped model never
    takes: the oracle's restatement of it against the reference's DepthNet called that way."""
    _, net = reference_model("8,8,8")
    sd = synth.randomize_state_dict(net.state_dict(), seed=6)
    net.load_state_dict(sd, strict=True)
    net.eval()
    feats = [f for f in synth.smooth_features(4, 32, 16, 20, seed=2)]
    _, pm, dv = synth.synth_sample(128, 160, 4, seed=7)
    proj = pm["stage1"]
    samples = (1.0 / torch.linspace(1 / 935.0, 1 / 425.0, 8)).view(1, 8, 1, 1).expand(1, 8, 16, 20).contiguous()
    with torch.no_grad():
        want = net.depthnet(feats, proj, depth_values=samples, num_depth=8, cost_regularization=net.cost_regularization,
                            pixel_wise_net=None, G=1)
        got = O.depthnet(sd, feats, proj, samples, 8, pixelwise_prefix=None)
    assert want["view_weights"] == [] and got["view_weights"] == []
    for k in ("volume", "reg_volume", "depth", "photometric_confidence"):
        assert torch.equal(got[k], want[k]), k


def test_our_modules_load_the_shipped_checkpoint_strictly():
    ckpt = os.path.join(REF_ROOT, "checkpoints", "Effi_MVS_plus", "model_tank.ckpt")
    if not os.path.exists(ckpt):
        pytest.skip("shipped checkpoint not present")
    from common import build_model
    net, _ = build_model("96,8,8")
    sd = torch.load(ckpt, map_location="cpu", weights_only=True)["model"]
    res = net.load_state_dict(sd, strict=True)
    assert not res.missing_keys and not res.unexpected_keys


def _loss_inputs(H, W, seed):
    """Synthetic ground truth for mvs_loss: per-stage depth maps + masks (stage k at 1/8, 1/4, 1/2, 1 of the image)."""
    g = torch.Generator().manual_seed(seed)
    gt, mask = {}, {}
    for k, f in (("stage1", 8), ("stage2", 4), ("stage3", 2), ("stage4", 1)):
        gt[k] = synth.DEPTH_MIN_MM + (synth.DEPTH_MAX_MM - synth.DEPTH_MIN_MM) * torch.rand(1, H // f, W // f, generator=g)
        mask[k] = (torch.rand(1, H // f, W // f, generator=g) > 0.3).float()
    return gt, mask


DLOSS = [1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4]       # train.py:246: which ground-truth scale each of the 13 outputs is compared with


@pytest.mark.parametrize("dropout_p", [0.0, 0.1])
def test_training_mode_loss_and_gradients_match_the_reference(dropout_p):
    """Scope row n2's checker: torch autograd through the oracle in training mode (batch-statistics BatchNorm, Dropout2d, the
    detach points of update.py:121 and Effi_MVS_plus.py:495) reproduces the reference's model.train() forward, loss and
    parameter gradients on CPU.  The FPN is included (its BatchNorms also run on batch statistics)."""
    ref, net = reference_model("8,8,8")
    sd = synth.randomize_state_dict(net.state_dict(), seed=11)
    net.load_state_dict(sd, strict=True)
    net.train()
    for m in net.modules():
        if isinstance(m, torch.nn.Dropout2d):
            m.p = dropout_p
    H, W, N = 64, 96, 3
    imgs, pm, dv = synth.synth_sample(H, W, N, seed=6)
    gt, mask = _loss_inputs(H, W, 1)
    torch.manual_seed(123)
    out = net(imgs, pm, dv)
    loss, _ = ref.module.mvs_loss(out["depth"], gt, mask, DLOSS)
    loss.backward()
    want = {k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None}
    want_stats = {k: v.clone() for k, v in net.state_dict().items() if "running_" in k or "num_batches" in k}

    # the reference registers some modules under two names (update_block.N / update_block_depthN+1, CSP_R.N / CSP_RN+1): tie the
    # aliases to one leaf tensor, as the shared nn.Parameter is
    sd2 = {k: v.clone() for k, v in sd.items()}
    groups = {}
    for k, p_ in net.named_parameters(remove_duplicate=False):
        groups.setdefault(id(p_), []).append(k)
    leaves = {}
    for ks in groups.values():
        leaf = sd2[ks[0]].requires_grad_(True)
        for k in ks:
            sd2[k] = leaf
            leaves[k] = leaf
    bgroups = {}
    for k, b_ in net.named_buffers(remove_duplicate=False):
        bgroups.setdefault(id(b_), []).append(k)
    for ks in bgroups.values():
        for k in ks[1:]:
            sd2[k] = sd2[ks[0]]
    torch.manual_seed(123)
    with O.training(dropout_p):
        got = O.full_forward(sd2, imgs, pm, dv, ndepths=(8, 8, 8))
        loss2, _ = O.mvs_loss(got["depth"], gt, mask, DLOSS)
    loss2.backward()
    assert torch.equal(loss2.detach(), loss.detach()), (float(loss2), float(loss))
    for a, b in zip(got["depth"], out["depth"]):
        assert torch.equal(a.detach(), b.detach())
    n_checked = 0
    for k, g in want.items():
        assert leaves[k].grad is not None, k
        assert torch.allclose(leaves[k].grad, g, rtol=1e-5, atol=1e-7 * float(g.abs().max()) + 1e-12), k
        n_checked += 1
    assert n_checked > 150
    for k, v in want_stats.items():                     # running statistics were updated the same way
        assert torch.allclose(sd2[k].detach().float(), v.float(), rtol=1e-6, atol=1e-7), k


def test_public_callables_of_the_three_mirrored_files_have_the_references_signatures():
    """Every live public class / function the reference's models/{module,update,Effi_MVS_plus}.py define (dead code of SURVEY
    section 2 excluded) exists in the mirror with the same parameter names, order and defaults; a mirror may only APPEND
    keyword arguments with defaults."""
    import inspect
    import sys

    from refimport import load_reference
    import effi_mvs_plus_amd.models  # noqa: F401
    ref = load_reference()
    ours = {"main": sys.modules["effi_mvs_plus_amd.models.Effi_MVS_plus"], "module": sys.modules["effi_mvs_plus_amd.models.module"],
            "update": sys.modules["effi_mvs_plus_amd.models.update"]}
    live = {
        "main": ["DepthNet", "bilinear_sampler", "pro_bilinear_sampler", "disp_to_depth", "depth_to_disp", "upsample_depth",
                 "GetCost_initvolume", "GetCost", "Effi_MVS_plus"],
        "module": ["homo_warping_new", "Conv3d", "Deconv3d", "Conv2d", "ConvBnReLU", "CostRegNet_2_sample_FPN3D_Fast", "cost_up_small",
                   "depth_regression", "get_cur_depth_range_samples", "get_depth_range_samples", "P_1to8_FeatureNet_Fast", "mvs_loss"],
        "update": ["DepthHead", "ConvGRU", "ProjectionInput", "BasicUpdateBlock"],
    }

    def sigs(obj):
        fs = [obj.__init__, obj.forward] if inspect.isclass(obj) else [obj]
        return [[(p.name, p.default if p.default is inspect.Parameter.empty or isinstance(p.default, (int, float, bool, str, type(None), list))
                  else repr(p.default)) for p in inspect.signature(f).parameters.values()] for f in fs]

    for mod, names in live.items():
        for name in names:
            want, got = sigs(getattr(getattr(ref, mod), name)), sigs(getattr(ours[mod], name))
            for w_, g_ in zip(want, got):
                assert g_[:len(w_)] == w_, f"{mod}.{name}: {g_} vs reference {w_}"
                for extra in g_[len(w_):]:
                    assert extra[1] is not inspect.Parameter.empty, f"{mod}.{name}: appended parameter {extra[0]} needs a default"


def test_fusion_and_dtu_filter_functions_have_the_references_signatures():
    """Row n3's surface: the functions of misc/fusion.py that test_tank.py:486-509 calls and the two of test_dtu_dypcd.py:164-233,
    by name, parameter names, order and defaults (the reference's files are read as SOURCE here: misc/fusion.py calls ``.cuda()`` at
    import-free definition time only, but test_dtu_dypcd.py imports cv2 / plyfile, absent from this image -- its two ``def`` lines are
    parsed with ``ast`` instead of imported)."""
    import ast
    import inspect
    import os
    import sys

    from refimport import REF_ROOT, load_reference
    if load_reference() is None:
        pytest.skip("reference tree not present")
    from effi_mvs_plus_amd import dtu_fusion, fusion

    def ast_sigs(path, names):
        tree = ast.parse(open(path).read())
        out = {}
        for node in tree.body:
            if isinstance(node, ast.FunctionDef) and node.name in names:
                a = node.args
                pos = [x.arg for x in a.args]
                defaults = [None] * (len(pos) - len(a.defaults)) + [ast.literal_eval(d) for d in a.defaults]
                out[node.name] = list(zip(pos, defaults, [i >= len(pos) - len(a.defaults) for i in range(len(pos))]))
        return out

    def our_sig(fn):
        return [(p_.name, None if p_.default is inspect.Parameter.empty else p_.default, p_.default is not inspect.Parameter.empty)
                for p_ in inspect.signature(fn).parameters.values()]

    tank = ["get_pixel_grids", "bin_op_reduce", "idx_img2cam", "idx_cam2world", "idx_world2cam", "idx_cam2img", "get_reproj_dynamic",
            "vis_filter_dynamic"]
    want = ast_sigs(os.path.join(REF_ROOT, "misc", "fusion.py"), tank)
    assert sorted(want) == sorted(tank)
    for name in tank:
        got = our_sig(getattr(fusion, name))
        assert got[:len(want[name])] == want[name], f"fusion.{name}: {got} vs reference {want[name]}"
        assert all(has_default for _, _, has_default in got[len(want[name]):]), f"fusion.{name}: appended parameters need defaults"
    dtu = ["reproject_with_depth", "check_geometric_consistency", "read_camera_parameters", "read_pair_file", "read_img", "save_mask"]
    want = ast_sigs(os.path.join(REF_ROOT, "test_dtu_dypcd.py"), dtu)
    assert sorted(want) == sorted(dtu)
    for name in dtu:
        got = our_sig(getattr(dtu_fusion, name))
        w_ = want[name]
        if name == "check_geometric_consistency":          # the reference's last parameter (confidence, unused) has no default; ours accepts its absence
            assert [g_[0] for g_ in got[:len(w_)]] == [x[0] for x in w_], (got, w_)
        else:
            assert got[:len(w_)] == w_, f"dtu_fusion.{name}: {got} vs reference {w_}"
        assert all(has_default for _, _, has_default in got[len(w_):]), f"dtu_fusion.{name}: appended parameters need defaults"
    # the driver's call sites name no other attribute of the module (test_tank.py:455-571)
    src = open(os.path.join(REF_ROOT, "test_tank.py")).read()
    import re
    used = set(re.findall(r"\bfusion\.([A-Za-z_][A-Za-z0-9_]*)", src))
    assert used and all(hasattr(fusion, u) for u in used), sorted(u for u in used if not hasattr(fusion, u))

