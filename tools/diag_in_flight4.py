"""Full eager passes on three streams at once (no graphs) vs sequential."""
import sys, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from common import build_model
from effi_mvs_plus_amd import ops, synth
DEV = "cuda:0"
net, sd = build_model("8,8,8", seed=6, device=DEV)
samples = []
with torch.no_grad():
    for seed in (31, 32, 33):
        imgs, pm, dv = synth.synth_sample(192, 256, 3, seed=seed)
        imgs = imgs.to(DEV)
        feats = [net.feature(imgs[:, v]) for v in range(3)]
        ctx = net.cnet_depth(imgs[:, 0])
        samples.append((feats, ctx, {k: v.to(DEV) for k, v in pm.items()}, dv.to(DEV)))
    want = [[d.clone() for d in net.forward_hot(*smp)["depth"]] for smp in samples]
    torch.cuda.synchronize()
    lanes = [torch.cuda.Stream() for _ in range(3)]
    cur = torch.cuda.current_stream()
    bad = {}
    for r in range(30):
        for st in lanes:
            st.wait_stream(cur)
        kept = []
        for i in range(6):
            with torch.cuda.stream(lanes[i % 3]):
                kept.append((i % 3, [d for d in net.forward_hot(*samples[i % 3])["depth"]]))
        for st in lanes:
            cur.wait_stream(st)
        torch.cuda.synchronize()
        for slot, depths in kept:
            for k, (a_, b_) in enumerate(zip(depths, want[slot])):
                if not torch.equal(a_, b_):
                    bad[(slot, k)] = bad.get((slot, k), 0) + 1
    print("eager on 3 streams: mismatches over 180 passes:", bad)
