"""Op-level stress: each operator of the cross-scale blocks on three streams at once vs its own sequential result."""
import sys, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from common import build_model
from effi_mvs_plus_amd import ops, synth, packing
DEV = "cuda:0"
net, sd = build_model("8,8,8", seed=6, device=DEV)
a, b = net.CSP_R[1], net.CSP_C[1]
D, h, w = 8, 48, 64            # half-resolution volume of the stage (x is [1, D, 2h, 2w])
g = torch.Generator().manual_seed(0)
sets = []
for i in range(3):
    x = torch.randn(1, D, 2 * h, 2 * w, generator=g).to(DEV)
    pa = torch.randn(1, D, h, w, generator=g).to(DEV)
    pb = torch.randn(1, D, h, w, generator=g).to(DEV)
    fa = torch.relu(torch.randn(8, D, h, w, generator=g)).to(DEV)
    ga = torch.relu(torch.randn(8, D, h, w, generator=g)).to(DEV)
    fb = torch.relu(torch.randn(8, D, h, w, generator=g)).to(DEV)
    gb = torch.relu(torch.randn(8, D, h, w, generator=g)).to(DEV)
    sets.append((x, pa, pb, fa, ga, fb, gb))
(w0a, b0a), (w0b, b0b) = a.conv0._packed(), b.conv0._packed()
(wca, bca), (wcb, bcb) = a.conv_cost._packed(), b.conv_cost._packed()
(w1a, b1a), (w1b, b1b) = a._roll_packed(), b._roll_packed()
(w2a, b2a), (w2b, b2b) = a.conv2._packed(), b.conv2._packed()
opsl = {
    "conv3d_k3_pair s2": lambda s: ops.conv3d_k3_pair(s[0], w0a, b0a, s[0], w0b, b0b, 8, sxy=2, relu=True),
    "conv3d_k3_pair s1": lambda s: ops.conv3d_k3_pair(s[1], wca, bca, s[2], wcb, bcb, 8, sxy=1, relu=True),
    "roll pair": lambda s: ops.conv3d_k3s1_roll_pair([s[3], s[4]], w1a, b1a, [s[5], s[6]], w1b, b1b, 8, relu=True),
    "deconv pair": lambda s: ops.deconv3d_k3_pair(s[3], w2a, b2a, s[5], w2b, b2b, 1, sz=1, relu=True),
}
# the warp + correlation of stages 2 / 3 and the pair lookup, on real features / cameras
samples = []
with torch.no_grad():
    for seed in (31, 32, 33):
        imgs, pm, dv = synth.synth_sample(192, 256, 3, seed=seed)
        imgs = imgs.to(DEV)
        feats = [net.feature(imgs[:, v]) for v in range(3)]
        nh = ops.to_nhwc([f["stage2"][0].contiguous() for f in feats])
        rt = ops.compose_rel_proj(pm["stage2"][0].to(DEV).contiguous())
        hh, ww = feats[0]["stage2"].shape[-2:]
        depth = (500.0 + 300.0 * torch.rand(hh, ww, generator=g)).to(DEV)
        vw = torch.rand(2, hh // 2, ww // 2, generator=g).to(DEV)
        itv = torch.tensor([1e-5], device=DEV)
        vol_a = torch.randn(8, hh // 2, ww // 2, generator=g).to(DEV)
        vol_b = torch.randn(8, hh // 2, ww // 2, generator=g).to(DEV)
        samp = (1.0 / (1 / 900.0 + (1 / 450.0 - 1 / 900.0) * torch.rand(8, hh, ww, generator=g))).to(DEV)
        lo = torch.tensor([450.0], device=DEV); hi = torch.tensor([900.0], device=DEV)
        samples.append((nh, rt, depth, itv, vw, vol_a, vol_b, samp, lo, hi, hh, ww))
opsl2 = {
    "warpcorr_dyn": lambda s: ops.warpcorr_dyn(s[0][0], s[0][1:], s[1], s[2], s[3], s[4], 8),
    "vol_lookup1d_pair": lambda s: ops.vol_lookup1d_pair(s[5], s[6], s[7], s[8], s[9], s[10] // 2, s[11] // 2),
}
lanes = [torch.cuda.Stream() for _ in range(3)]
cur = torch.cuda.current_stream()
with torch.no_grad():
    for name, fn in list(opsl.items()) + list(opsl2.items()):
        sets_ = samples if name in opsl2 else sets
        want = [[t.clone() for t in fn(s)] for s in sets_]
        torch.cuda.synchronize()
        graphs, outs = [], []
        for s_ in sets_:
            g_ = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g_, capture_error_mode="thread_local"):
                for _ in range(8):
                    o = fn(s_)
            graphs.append(g_); outs.append(o)
        torch.cuda.synchronize()
        bad = 0
        for r in range(60):
            for st in lanes:
                st.wait_stream(cur)
            kept = []
            for i in range(12):
                with torch.cuda.stream(lanes[i % 3]):
                    graphs[i % 3].replay()
                    kept.append((i % 3, [t.clone() for t in outs[i % 3]]))
            for st in lanes:
                cur.wait_stream(st)
            torch.cuda.synchronize()
            for slot, out in kept:
                if not all(torch.equal(x_, y_) for x_, y_ in zip(out, want[slot])):
                    bad += 1
        print(f"{name} (graphs of 8 launches, 3 in flight): {bad} mismatches of 720")
        del graphs, outs
