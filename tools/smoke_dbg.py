import os, sys, torch
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
from common import build_model
from effi_mvs_plus_amd import synth, ops
from oracle import effi_oracle as O
dev = "cuda:0"
net, sd = build_model("8,8,8", seed=1, device=dev)
imgs, pm, dv = synth.synth_sample(128, 160, 4, seed=0)
rng = synth.DEPTH_MAX_MM - synth.DEPTH_MIN_MM
with torch.no_grad():
    feats = [O.feature_net(sd, "feature", imgs[:, v]) for v in range(imgs.size(1))]
    ctx = O.feature_net(sd, "cnet_depth", imgs[:, 0])
    want = O.hot_path(sd, feats, ctx, pm, dv, ndepths=(8, 8, 8))
    args = ([{k: v.to(dev) for k, v in f.items()} for f in feats], {k: v.to(dev) for k, v in ctx.items()}, {k: v.to(dev) for k, v in pm.items()}, dv.to(dev))
    for mode in ("split", "fp32"):
        ops.set_precision(mode)
        for br in (True, False):
            ops.set_branches(br)
            for rep in range(2):
                got = net.forward_hot(*args)
                torch.cuda.synchronize()
                errs = [((a.cpu() - b).abs().mean() / rng).item() for a, b in zip(got["depth"], want["depth"])]
                print(mode, "branches", br, "rep", rep, " ".join(f"{e:.1e}" for e in errs))
