"""Trace every HIP operator's outputs inside the captured pass: which operator's output differs FIRST under concurrency?"""
import sys, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from common import build_model
from effi_mvs_plus_amd import ops, synth, _lib
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

    trace = None
    names = [n for n in dir(ops) if not n.startswith("_") and callable(getattr(ops, n)) and getattr(getattr(ops, n), "__module__", "") == ops.__name__
             and n not in ("check", "mark", "set_marks", "set_profile", "get_profile", "set_precision", "get_precision", "uses_split",
                           "set_branches", "get_branches", "on_tensor_device", "ensure_workspace", "Branch", "KernelProfile")]
    real = {n: getattr(ops, n) for n in names}

    def wrap(n, f):
        def g(*a, **k):
            out = f(*a, **k)
            if trace is not None:
                ts = [out] if isinstance(out, torch.Tensor) else [t for t in (out if isinstance(out, (tuple, list)) else []) if isinstance(t, torch.Tensor)]
                for j, t in enumerate(ts):
                    trace.append((f"{len(trace):03d} {n}[{j}]", t))
            return out
        return g
    for n, f in real.items():
        setattr(ops, n, wrap(n, f))

    refs = []
    for smp in samples:
        trace = []
        net.forward_hot(*smp)
        refs.append([(k, t.clone()) for k, t in trace])
    torch.cuda.synchronize()
    graphs, traces, ins = [], [], []
    for smp in samples:
        inp = _clone_tree(smp)
        trace = []
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            net.forward_hot(*inp)
        graphs.append(g); traces.append(trace); ins.append(inp)
    trace = None
    torch.cuda.synchronize()
    lanes = [torch.cuda.Stream() for _ in range(3)]
    cur = torch.cuda.current_stream()
    first = {}
    shown = 0
    for r in range(400):
        for st in lanes:
            st.wait_stream(cur)
        for i in range(3):
            with torch.cuda.stream(lanes[i]):
                graphs[i].replay()
        for st in lanes:
            cur.wait_stream(st)
        torch.cuda.synchronize()
        for slot in range(3):
            for (k, t), (k2, t2) in zip(traces[slot], refs[slot]):
                if not torch.equal(t, t2):
                    first[k] = first.get(k, 0) + 1
                    if shown < 4:
                        shown += 1
                        dd = (t - t2).abs()
                        idx = (dd > 0).nonzero()
                        print("slot", slot, k, tuple(t.shape), "differing:", int((dd > 0).sum()), "min idx", idx.min(0).values.tolist(), "max idx",
                              idx.max(0).values.tolist(), "max |diff|", float(dd.max()))
                        ys = sorted(set(idx[:, 1].tolist())); xs = sorted(set(idx[:, 2].tolist())) if idx.shape[1] > 2 else []
                        print("    rows:", ys[:40], "cols:", xs[:48])
                        i0 = idx[0].tolist()
                        print("    first:", i0, "got", float(t[tuple(i0)]), "want", float(t2[tuple(i0)]))
                    break
    print("first differing operator output (call index, op[output]) -> count over 360 replays:")
    for k in sorted(first):
        print("  ", k, first[k])
