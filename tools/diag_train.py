import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import torch
from common import build_model
from effi_mvs_plus_amd import synth, ops
from effi_mvs_plus_amd.models import mvs_loss
from test_gpu_train import _loss_inputs, _oracle_training_pass, DLOSS, rel
DEV="cuda:0"
import itertools
H, W, nd = 128, 160, (8,8,8)
B, N = int(sys.argv[1]), int(sys.argv[2])
net, sd = build_model("8,8,8", seed=13, device=DEV)
net.train()
for m in net.modules():
    if isinstance(m, torch.nn.Dropout2d): m.p = 0.0
samples = [synth.synth_sample(H, W, N, seed=30 + b) for b in range(B)]
imgs = torch.cat([s_[0] for s_ in samples]); pm = {k: torch.cat([s_[1][k] for s_ in samples]) for k in samples[0][1]}; dv = torch.cat([s_[2] for s_ in samples])
gt, mask = _loss_inputs(H, W, B, 2)
_, _, l32, _ = _oracle_training_pass(net, sd, imgs, pm, dv, gt, mask, nd)
_, _, l64, _ = _oracle_training_pass(net, sd, imgs, pm, dv, gt, mask, nd, dtype=torch.float64)
torch.backends.cudnn.enabled = False
runs = {}
for tag, prec in (("fp32-a", "fp32"), ("split", "fp32"), ("fp32-b", "fp32")):
    ops.set_precision(prec)
    net.load_state_dict({k: v.to(DEV) for k, v in sd.items()})
    net.zero_grad()
    out = net(imgs.to(DEV), {k: v.to(DEV) for k, v in pm.items()}, dv.to(DEV))
    loss, _ = mvs_loss(out["depth"], {k: v.to(DEV) for k, v in gt.items()}, {k: v.to(DEV) for k, v in mask.items()}, DLOSS)
    loss.backward()
    runs[tag] = {k: p_.grad.detach().clone() for k, p_ in net.named_parameters()}
rows = []
for k in l64:
    if k not in runs["fp32-a"]: continue
    e_ref = rel(l32[k].grad, l64[k].grad)
    rows.append((max(rel(runs[t][k], l64[k].grad) for t in runs) / max(1e-3, 8 * e_ref), k, e_ref, [rel(runs[t][k], l64[k].grad) for t in runs], rel(runs["fp32-b"][k], runs["fp32-a"][k])))
rows.sort(reverse=True)
print("ratio to bound | ref32-vs-64 | [fp32-a, split, fp32-b] vs fp64 | fp32 run-to-run | parameter")
for r, k, e_ref, es, rr in rows[:6]:
    print(f"{r:6.2f}  {e_ref:.2e}  [{es[0]:.2e} {es[1]:.2e} {es[2]:.2e}]  {rr:.1e}  {k}")
