import os, sys, torch
import torch.nn.functional as F
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
from common import build_model
from effi_mvs_plus_amd import ops
dev = "cuda:0"
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
net, sd = build_model("8,8,8", seed=seed, device=dev)
reg = net.cost_regularization
g = torch.Generator().manual_seed(0)
x = torch.randn(1, 8, 16, 20, generator=g).to(dev)          # [1(C), D, h, w]
ops.set_precision("fp32")
def ref(m, t):   # torch reference of a Conv3d/Deconv3d wrapper
    y = m.conv(t.unsqueeze(0))
    if m.bn is not None: y = m.bn(y)
    return (F.relu(y) if m.relu else y)[0]
with torch.no_grad():
    c0h, c0r = reg.conv0.run([x]), ref(reg.conv0, x)
    print("conv0", float((c0h - c0r).abs().max()), float(c0r.abs().max()))
    c1h, c1r = reg.conv1.run([c0r]), ref(reg.conv1, c0r)
    print("conv1", float((c1h - c1r).abs().max()), float(c1r.abs().max()))
    c2h, c2r = reg.conv2.run([c1r]), ref(reg.conv2, c1r)
    print("conv2 s2", float((c2h - c2r).abs().max()), float(c2r.abs().max()), tuple(c2r.shape))
    c3h, c3r = reg.conv3.run([c2r]), ref(reg.conv3, c2r)
    print("conv3", float((c3h - c3r).abs().max()), float(c3r.abs().max()))
    c4h, c4r = reg.conv4.run([c3r]), ref(reg.conv4, c3r)
    print("conv4 s2", float((c4h - c4r).abs().max()), float(c4r.abs().max()), tuple(c4r.shape))
    c5h, c5r = reg.conv5.run([c4r]), ref(reg.conv5, c4r)
    print("conv5", float((c5h - c5r).abs().max()), float(c5r.abs().max()))
    c6h, c6r = reg.conv6.run(c5r), ref(reg.conv6, c5r)
    print("conv6", float((c6h - c6r).abs().max()), float(c6r.abs().max()))
    c7h, c7r = reg.conv7.run(c6r), ref(reg.conv7, c6r)
    print("conv7", float((c7h - c7r).abs().max()), float(c7r.abs().max()))
    for name in [n for n, _ in reg.named_children()]: print(name, end=" ")
    print()
