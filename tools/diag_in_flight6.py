"""Does the corruption need memory reuse inside a captured pass?  Same stress with every tensor allocated during capture kept alive."""
import sys, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from common import build_model
from effi_mvs_plus_amd import ops, synth
from effi_mvs_plus_amd.graph import _clone_tree
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
    keep = []
    real = {n: getattr(torch, n) for n in ("empty", "zeros", "empty_like", "zeros_like", "ones")}

    def wrap(f):
        def g(*a, **k):
            t = f(*a, **k)
            keep.append(t)
            return t
        return g

    def build(retain):
        if retain:
            for n, f in real.items():
                setattr(torch, n, wrap(f))
        graphs, outs, ins = [], [], []
        try:
            for smp in samples:
                inp = _clone_tree(smp)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, capture_error_mode="thread_local"):
                    out = net.forward_hot(*inp)
                graphs.append(g); outs.append(out); ins.append(inp)
        finally:
            for n, f in real.items():
                setattr(torch, n, f)
        return graphs, outs, ins

    def stress(graphs, outs, label):
        lanes = [torch.cuda.Stream() for _ in range(3)]
        cur = torch.cuda.current_stream()
        bad = 0
        for r in range(60):
            for st in lanes:
                st.wait_stream(cur)
            kept = []
            for i in range(12):
                with torch.cuda.stream(lanes[i % 3]):
                    graphs[i % 3].replay()
                    kept.append((i % 3, [d.clone() for d in outs[i % 3]["depth"]]))
            for st in lanes:
                cur.wait_stream(st)
            torch.cuda.synchronize()
            for slot, depths in kept:
                if not all(torch.equal(a_, b_) for a_, b_ in zip(depths, want[slot])):
                    bad += 1
        print(label, bad, "of 720")

    for retain in (False, True):
        graphs, outs, ins = build(retain)
        torch.cuda.synchronize()
        stress(graphs, outs, f"retain all capture-time allocations = {retain} ({len(keep)} tensors kept): mismatches")
