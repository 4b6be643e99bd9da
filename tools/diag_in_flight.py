import sys, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from common import build_model
from effi_mvs_plus_amd import ops, synth
from effi_mvs_plus_amd.graph import HotPathGraph
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
    def eq(a, b):
        return [bool(torch.equal(x, y)) for x, y in zip(a, b)]
    want = [[d.clone() for d in net.forward_hot(*smp)["depth"]] for smp in samples]
    again = [[d.clone() for d in net.forward_hot(*smp)["depth"]] for smp in samples]
    print("eager twice:", [all(eq(a, b)) for a, b in zip(want, again)])
    ops.set_branches(True)
    eb = [[d.clone() for d in net.forward_hot(*smp)["depth"]] for smp in samples]
    torch.cuda.synchronize()
    print("eager branches on vs off:", [eq(a, b) for a, b in zip(eb, want)])
    g = HotPathGraph(net, *samples[0], slots=3)
    ops.set_branches(False)
    for i, smp in enumerate(samples):
        g.load(i, *smp)
    torch.cuda.synchronize()
    for rep in range(2):
        seq = []
        for i in range(3):
            seq.append([d.clone() for d in g.replay(i)["depth"]])
            torch.cuda.synchronize()
        print("graph(branches) sequential vs eager:", [eq(a, b) for a, b in zip(seq, want)])
    def concurrent(g, label, rounds=40):
        lanes = [torch.cuda.Stream() for _ in range(3)]
        cur = torch.cuda.current_stream()
        bad = {}
        for r in range(rounds):
            for st in lanes:
                st.wait_stream(cur)
            kept = []
            for i in range(12):
                with torch.cuda.stream(lanes[i % 3]):
                    out = g.replay(i % 3)
                    kept.append((i % 3, [d.clone() for d in out["depth"]]))
            for st in lanes:
                cur.wait_stream(st)
            torch.cuda.synchronize()
            for slot, depths in kept:
                e = eq(depths, want[slot])
                for k, ok in enumerate(e):
                    if not ok:
                        bad[(slot, k)] = bad.get((slot, k), 0) + 1
        print(label, "mismatches (slot, depth index) -> count over", rounds * 12, "replays:", bad)

    import os
    if os.environ.get("DIAG_LINEAR_ONLY") is None:
        concurrent(g, "two-stream graphs, 3 in flight:")
    g2 = HotPathGraph(net, *samples[0], slots=3)          # linear graphs
    for i, smp in enumerate(samples):
        g2.load(i, *smp)
    torch.cuda.synchronize()
    concurrent(g2, "linear graphs, 3 in flight:")
    print("done")
