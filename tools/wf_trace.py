import os, sys, torch
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"]); sys.path.insert(0, os.path.join(os.environ["GRAFT_REPO_ROOT"], "tests"))
from common import build_model
from effi_mvs_plus_amd import synth
from effi_mvs_plus_amd.graph import ForwardGraph
dev = "cuda:0"
net, sd = build_model("48,8,8", seed=1, device=dev)
imgs, pm, dv = synth.synth_sample(1184, 1600, 5, seed=0)
with torch.no_grad():
    fg = ForwardGraph(net, imgs.to(dev), {k: v.to(dev) for k, v in pm.items()}, dv.to(dev))
    for _ in range(4):
        fg.replay()
    torch.cuda.synchronize()
