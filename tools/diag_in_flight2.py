"""Which tensor differs first when three captured passes run concurrently?  (intermediates of forward_hot)"""
import sys, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from common import build_model
from effi_mvs_plus_amd import ops, synth
from effi_mvs_plus_amd.graph import ReplayGraph
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
    fn = lambda *a: net.forward_hot(*a, want_intermediates=True)

    def flat(out):
        d = {f"depth{i}": t for i, t in enumerate(out["depth"])}
        d.update(out["intermediates"])
        return {k: v.clone() for k, v in d.items()}

    want = [flat(fn(*smp)) for smp in samples]
    g = ReplayGraph(fn, samples[0], slots=3)
    for i, smp in enumerate(samples):
        g.load(i, *smp)
    torch.cuda.synchronize()
    lanes = [torch.cuda.Stream() for _ in range(3)]
    cur = torch.cuda.current_stream()
    bad, mag, cnt, where = {}, {}, {}, []
    for r in range(40):
        for st in lanes:
            st.wait_stream(cur)
        kept = []
        for i in range(12):
            with torch.cuda.stream(lanes[i % 3]):
                kept.append((i % 3, flat(g.replay(i % 3))))
        for st in lanes:
            cur.wait_stream(st)
        torch.cuda.synchronize()
        for slot, d in kept:
            for k, v in d.items():
                if not torch.equal(v, want[slot][k]):
                    bad[k] = bad.get(k, 0) + 1
                    dd = (v - want[slot][k]).abs()
                    nbad = int((dd > 0).sum())
                    mag[k] = max(mag.get(k, 0.0), float(dd.max()))
                    cnt[k] = max(cnt.get(k, 0), nbad)
                    if k.startswith(("cur_volume", "reg_volume")) and len(where) < 6:
                        idx = (dd > 0).nonzero()
                        where.append((k, tuple(v.shape), idx.min(0).values.tolist(), idx.max(0).values.tolist(), nbad))
    print("mismatching tensors -> count over 480 replays:")
    for k in sorted(bad):
        print("  ", k, bad[k], "max |diff|", mag[k], "max #elements differing", cnt[k])
    for w_ in where:
        print("   where:", w_)
