import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import torch
from common import build_model
from effi_mvs_plus_amd import synth, ops, train_path as TP, autograd as A
from effi_mvs_plus_amd.models import mvs_loss
from oracle import effi_oracle as O
from test_gpu_train import _loss_inputs, DLOSS, rel
DEV = "cuda:0"
H, W, nd, B, N = 128, 160, (8, 8, 8), 1, 2
net, sd = build_model("8,8,8", seed=13, device=DEV)
net.train()
for m in net.modules():
    if isinstance(m, torch.nn.Dropout2d): m.p = 0.0
imgs, pm, dv = synth.synth_sample(H, W, N, seed=30)
gt, mask = _loss_inputs(H, W, B, 2)
torch.backends.cudnn.enabled = False
stash_h, stash_o = [], []
orig_mh, orig_omh = TP.mask_head, O.mask_head
def mh(block, net_):
    hid = A.conv2d([net_], block.mask[0].weight, block.mask[0].bias, ops.ACT_RELU)
    m_ = 0.25 * A.conv2d([hid], block.mask[2].weight, block.mask[2].bias, ops.ACT_NONE)
    for t_ in (net_, hid, m_): t_.retain_grad()
    stash_h.append((net_, hid, m_)); return m_
def omh(sd_, prefix, net_):
    import torch.nn.functional as F
    hid = F.relu(O.conv2d(net_, sd_, prefix + ".0", 1)); m_ = 0.25 * O.conv2d(hid, sd_, prefix + ".2", 0)
    for t_ in (net_, hid, m_): t_.retain_grad()
    stash_o.append((net_, hid, m_)); return m_
TP.mask_head, O.mask_head = mh, omh
imgs_d = imgs.to(DEV)
feats = [net.feature(imgs_d[:, v]) for v in range(N)]
ctx = net.cnet_depth(imgs_d[:, 0])
out = net.forward_hot(feats, ctx, {k: v.to(DEV) for k, v in pm.items()}, dv.to(DEV))
loss, _ = mvs_loss(out["depth"], {k: v.to(DEV) for k, v in gt.items()}, {k: v.to(DEV) for k, v in mask.items()}, DLOSS)
loss.backward()
sd3 = {k: v.detach().clone().requires_grad_(v.is_floating_point() and "running" not in k) for k, v in sd.items()}
fl = [{k: v.detach().cpu().clone() for k, v in f.items()} for f in feats]
cl = {k: v.detach().cpu().clone() for k, v in ctx.items()}
with O.training(0.0):
    o2 = O.hot_path(sd3, fl, cl, pm, dv, ndepths=nd)
    l2, _ = O.mvs_loss(o2["depth"], gt, mask, DLOSS)
l2.backward()
print("loss", float(loss.detach()), float(l2.detach()))
for s_, ((n1, h1, m1), (n2, h2, m2)) in enumerate(zip(stash_h, stash_o)):
    print("stage", s_ + 1, "net", f"{rel(n1, n2):.1e}", "hid", f"{rel(h1, h2):.1e}", "mask", f"{rel(m1, m2):.1e}", "| g mask", f"{rel(m1.grad, m2.grad):.1e}",
          "g hid", f"{rel(h1.grad, h2.grad):.1e}", "g net", f"{rel(n1.grad, n2.grad):.1e}")
    hidz = (h2 == 0).float().mean().item()
    e = (h1.grad.cpu() - h2.grad).abs(); pk = float(h2.grad.abs().max())
    bad = torch.nonzero(e > 1e-3 * pk)
    print("     hid zeros frac", f"{hidz:.3f}", "g hid elements off:", len(bad), "of", e.numel(), bad[:5].tolist())
    for (b_, c_, y_, x_) in bad[:5].tolist():
        print("       at", (b_, c_, y_, x_), "hid hip/cpu", float(h1[b_, c_, y_, x_]), float(h2[b_, c_, y_, x_]), "g hip/cpu", float(h1.grad[b_, c_, y_, x_]), float(h2.grad[b_, c_, y_, x_]))
    blk = net.update_block[s_]
    print("     mask.0.w", f"{rel(blk.mask[0].weight.grad, sd3[f'update_block.{s_}.mask.0.weight'].grad):.1e}", "mask.0.b", f"{rel(blk.mask[0].bias.grad, sd3[f'update_block.{s_}.mask.0.bias'].grad):.1e}",
          "mask.2.w", f"{rel(blk.mask[2].weight.grad, sd3[f'update_block.{s_}.mask.2.weight'].grad):.1e}")
print("---- recompute the stage-1 mask.0 backward by hand from the stashed tensors")
import torch.nn.functional as F
(n1, h1, m1), (n2, h2, m2) = stash_h[0], stash_o[0]
blk = net.update_block[0]
Wd = blk.mask[0].weight.detach().clone().requires_grad_(True); bd = blk.mask[0].bias.detach().clone().requires_grad_(True)
xin = n1.detach().clone().requires_grad_(True)
y = A.conv2d([xin], Wd, bd, ops.ACT_RELU)
y.backward(h1.grad)
print("HIP recompute vs model p.grad:  W", f"{rel(Wd.grad, blk.mask[0].weight.grad):.1e}", "b", f"{rel(bd.grad, blk.mask[0].bias.grad):.1e}")
print("HIP recompute vs oracle:        W", f"{rel(Wd.grad, sd3['update_block.0.mask.0.weight'].grad):.1e}", "b", f"{rel(bd.grad, sd3['update_block.0.mask.0.bias'].grad):.1e}", "y", f"{rel(y, h1):.1e}")
Wc = sd3['update_block.0.mask.0.weight'].detach().clone().requires_grad_(True); bc = sd3['update_block.0.mask.0.bias'].detach().clone().requires_grad_(True)
yc = F.relu(F.conv2d(n2.detach(), Wc, bc, padding=1)); yc.backward(h2.grad)
print("CPU recompute vs oracle p.grad: W", f"{rel(Wc.grad, sd3['update_block.0.mask.0.weight'].grad):.1e}")
print("alias check: update_block[0] is update_block_depth1:", net.update_block[0] is net.update_block_depth1,
      " oracle grads of the alias key:", sd3['update_block_depth1.mask.0.weight'].grad is None)
print("---- same inputs through torch on the GPU (native kernels)")
Wt = blk.mask[0].weight.detach().clone().requires_grad_(True); bt = blk.mask[0].bias.detach().clone().requires_grad_(True)
yt = F.relu(F.conv2d(n1.detach(), Wt, bt, padding=1)); yt.backward(h1.grad)
print("torch-GPU(same HIP inputs) vs HIP recompute: W", f"{rel(Wd.grad, Wt.grad):.1e}", "b", f"{rel(bd.grad, bt.grad):.1e}", " y", f"{rel(y, yt):.1e}")
print("torch-GPU(same HIP inputs) vs oracle:        W", f"{rel(Wt.grad, sd3['update_block.0.mask.0.weight'].grad):.1e}")
gp_h = ops.pointwise(ops.PW_ACT_BWD[ops.ACT_RELU], h1.grad.contiguous(), h1.detach())
gp_t = h1.grad * (h1.detach() > 0)
print("relu-masked grad: HIP pointwise vs torch", float((gp_h - gp_t).abs().max()), " |g_hid| peak", float(h1.grad.abs().max()), "mean", float(h1.grad.abs().mean()))
flips = ((h1.detach().cpu() > 0) != (h2.detach() > 0))
print("relu flips between HIP and CPU hid:", int(flips.sum()), " |g| at flips:", h2.grad[flips].abs().tolist()[:5])
print("dW peak", float(Wt.grad.abs().max()), "db peak", float(bt.grad.abs().max()))
print("---- pieces")
gp = gp_t.contiguous()
print("channel_sum vs torch.sum:", rel(ops.channel_sum(gp), gp.sum((0, 2, 3))), " bd.grad vs torch.sum:", rel(bd.grad, gp.sum((0, 2, 3))), " bt.grad vs torch.sum", rel(bt.grad, gp.sum((0, 2, 3))))
dw = torch.zeros(96, 48, 9, device=DEV)
ops.conv_wgrad(gp[0], n1.detach()[0].contiguous(), dw, 0, 1, 3)
print("conv_wgrad vs torch-GPU dW:", rel(dw.view(96, 48, 3, 3), Wt.grad), " Wd.grad vs conv_wgrad:", rel(Wd.grad, dw.view(96, 48, 3, 3)))
print("h1.grad: contiguous", h1.grad.is_contiguous(), h1.grad.stride(), h1.grad.shape, " n1", n1.stride(), n1.is_contiguous(), "xin", xin.stride())
