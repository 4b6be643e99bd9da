import sys, os, math
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import torch, torch.nn.functional as F
from common import build_model
from effi_mvs_plus_amd import autograd as A, ops, train_path as TP, synth
from oracle import effi_oracle as O
from test_gpu_train import rel, leaf
DEV = "cuda:0"
net, sd = build_model("8,8,8", seed=13, device=DEV)
net.train()
for m in net.modules():
    if isinstance(m, torch.nn.Dropout2d): m.p = 0.0
for B in (1, 2):
    g = torch.Generator().manual_seed(B)
    h, w, hd, cd = 16, 20, 48, 12
    lo, hi = 1 / 935.0, 1 / 425.0
    dv = torch.linspace(lo, hi, 384).view(1, 384).repeat(B, 1)
    cur, reg = torch.randn(B, 8, h, w, generator=g), torch.randn(B, 8, h, w, generator=g)
    hidden = torch.tanh(torch.randn(B, hd, h, w, generator=g)); inp = torch.relu(torch.randn(B, cd, h, w, generator=g))
    inv0 = torch.rand(B, 1, h, w, generator=g)
    itv = torch.full((B,), (hi - lo) / 384 * 4)
    gmin, gmax = torch.full((B, 1, 1, 1), 1 / hi), torch.full((B, 1, 1, 1), 1 / lo)
    gup = torch.randn(B, 2 * h, 2 * w, generator=g)
    # oracle
    sd2 = {k: v.clone() for k, v in sd.items()}
    pref = "update_block.0"
    lv = {k: sd2[k].requires_grad_(True) for k in sd2 if k.startswith(pref)}
    cc, rc, hc, ic = leaf(cur), leaf(reg), leaf(hidden), leaf(inp)
    pro = [rc.permute(0, 2, 3, 1).reshape(B * h * w, 1, 1, 8), cc.permute(0, 2, 3, 1).reshape(B * h * w, 1, 1, 8)]
    scale = lambda d: O.disp_to_depth(d, gmin, gmax)
    with O.training(0.0):
        _, masks, invs = O.update_block(sd2, pref, hc, lambda depth, it: O.getcost(depth, pro, itv.view(B, 1, 1, 1), 3, gmax, gmin, [B, h, w]), inv0, ic, 3, scale)
        up = O.upsample_depth(invs[-1], masks[-1], ratio=2)
    (up * gup).sum().backward()
    # hip
    blk = net.update_block[0]
    for p_ in blk.parameters(): p_.grad = None
    cd_, rd_, hd_, id_ = leaf(cur, DEV), leaf(reg, DEV), leaf(hidden, DEV), leaf(inp, DEV)
    cost_fn = lambda inv, i: A.getcost(cd_, rd_, inv, dv.to(DEV), itv.to(DEV), gmin.to(DEV), gmax.to(DEV), 3)
    _, masks_d, invs_d = TP.update_block(blk, hd_, cost_fn, inv0.to(DEV), id_, 3)
    up_d = A.convex_upsample(invs_d[-1], masks_d[-1])
    (up_d * gup.to(DEV)).sum().backward()
    print("B", B, "up", f"{rel(up_d, up):.1e}", "mask", f"{rel(masks_d[-1], masks[-1]):.1e}", "d hidden", f"{rel(hd_.grad, hc.grad):.1e}", "d inp", f"{rel(id_.grad, ic.grad):.1e}",
          "d cur", f"{rel(cd_.grad, cc.grad):.1e}")
    for k, p_ in blk.named_parameters():
        e = rel(p_.grad, lv[pref + "." + k].grad)
        if e > 1e-4: print("    ", k, f"{e:.2e}")
