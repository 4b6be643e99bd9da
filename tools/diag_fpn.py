import sys, os, copy
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import torch
from common import build_model
from effi_mvs_plus_amd import synth
from test_gpu_train import rel
DEV = "cuda:0"
net, sd = build_model("8,8,8", seed=13)
imgs, pm, dv = synth.synth_sample(128, 160, 3, seed=30)
for name in ("cnet_depth", "feature"):
    cpu = copy.deepcopy(getattr(net, name)).train()
    gpu = copy.deepcopy(getattr(net, name)).to(DEV).train()
    x = imgs[:, 0]
    oc = cpu(x)
    og = gpu(x.to(DEV))
    g = torch.Generator().manual_seed(1)
    R = {k: torch.randn(v.shape, generator=g) for k, v in oc.items()}
    sum((oc[k] * R[k]).sum() for k in oc).backward()
    sum((og[k] * R[k].to(DEV)).sum() for k in og).backward()
    worst = max((rel(pg.grad, pc.grad), k) for (k, pc), (_, pg) in zip(cpu.named_parameters(), gpu.named_parameters()))
    print(name, "forward", max(rel(og[k], oc[k]) for k in oc), "worst param grad", worst)
    torch.backends.cudnn.enabled = False
    gpu2 = copy.deepcopy(getattr(net, name)).to(DEV).train()
    og2 = gpu2(x.to(DEV))
    sum((og2[k] * R[k].to(DEV)).sum() for k in og2).backward()
    worst2 = max((rel(pg.grad, pc.grad), k) for (k, pc), (_, pg) in zip(cpu.named_parameters(), gpu2.named_parameters()))
    print(name, "  with MIOpen disabled: worst param grad", worst2)
    torch.backends.cudnn.enabled = True
