"""Chains of the cross-scale block's operators as graphs, three in flight: which prefix of the chain is enough for the corruption?"""
import sys, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from common import build_model
from effi_mvs_plus_amd import ops, synth
DEV = "cuda:0"
net, sd = build_model("8,8,8", seed=6, device=DEV)
a, b = net.CSP_R[1], net.CSP_C[1]
D, h, w = 8, 48, 64
g = torch.Generator().manual_seed(0)
sets = []
for i in range(3):
    x = torch.randn(1, D, 2 * h, 2 * w, generator=g).to(DEV)
    pa = torch.randn(1, D, h, w, generator=g).to(DEV)
    pb = torch.randn(1, D, h, w, generator=g).to(DEV)
    sets.append((x, pa, pb))
(w0a, b0a), (w0b, b0b) = a.conv0._packed(), b.conv0._packed()
(wca, bca), (wcb, bcb) = a.conv_cost._packed(), b.conv_cost._packed()
(w1a, b1a), (w1b, b1b) = a._roll_packed(), b._roll_packed()
(w2a, b2a), (w2b, b2b) = a.conv2._packed(), b.conv2._packed()


def chain(s, upto):
    fa, fb = ops.conv3d_k3_pair(s[0], w0a, b0a, s[0], w0b, b0b, 8, sxy=2, relu=True)
    if upto == 1:
        return [fa, fb]
    ga, gb = ops.conv3d_k3_pair(s[1], wca, bca, s[2], wcb, bcb, 8, sxy=1, relu=True)
    if upto == 2:
        return [fa, fb, ga, gb]
    c1a, c1b = ops.conv3d_k3s1_roll_pair([fa, ga], w1a, b1a, [fb, gb], w1b, b1b, 8, relu=True)
    if upto == 3:
        return [c1a, c1b]
    oa, ob = ops.deconv3d_k3_pair(c1a, w2a, b2a, c1b, w2b, b2b, 1, sz=1, relu=True)
    return [oa, ob, c1a, c1b]


from effi_mvs_plus_amd.graph import HotPathGraph
lanes = [torch.cuda.Stream() for _ in range(3)]
cur = torch.cuda.current_stream()
with torch.no_grad():
    smp = []
    for seed in (31, 32):
        imgs, pm, dv = synth.synth_sample(192, 256, 3, seed=seed)
        imgs = imgs.to(DEV)
        feats = [net.feature(imgs[:, v]) for v in range(3)]
        ctx = net.cnet_depth(imgs[:, 0])
        smp.append((feats, ctx, {k: v.to(DEV) for k, v in pm.items()}, dv.to(DEV)))
    agg = HotPathGraph(net, *smp[0], slots=2)
    for i_ in range(2):
        agg.load(i_, *smp[i_])
    want_agg = [[d.clone() for d in agg.replay(i_)["depth"]] for i_ in range(2)]
    want = [t.clone() for t in chain(sets[0], 4)]
    torch.cuda.synchronize()
    g_ = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g_, capture_error_mode="thread_local"):
        for _ in range(6):
            o = chain(sets[0], 4)
    torch.cuda.synchronize()
    bad_v, bad_a = 0, 0
    for r in range(80):
        for st in lanes:
            st.wait_stream(cur)
        kept, kept_a = [], []
        for i in range(4):
            with torch.cuda.stream(lanes[0]):
                g_.replay()
                kept.append([t.clone() for t in o])
            for l_ in (1, 2):
                with torch.cuda.stream(lanes[l_]):
                    kept_a.append((l_ - 1, [d.clone() for d in agg.replay(l_ - 1)["depth"]]))
        for st in lanes:
            cur.wait_stream(st)
        torch.cuda.synchronize()
        for out in kept:
            if not all(torch.equal(x_, y_) for x_, y_ in zip(out, want)):
                bad_v += 1
        for slot, out in kept_a:
            if not all(torch.equal(x_, y_) for x_, y_ in zip(out, want_agg[slot])):
                bad_a += 1
    print(f"victim = cross-scale chain graph next to two full passes: {bad_v} mismatches of 320; the full passes themselves: {bad_a} of 640")
