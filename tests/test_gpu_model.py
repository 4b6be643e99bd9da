"""GPU parity of the whole cascade (the path bench.py times) against the golden vectors generated from
the reference and against the CPU oracle, in the normalised units of SURVEY.md section 8(d):

    mean |d_hip - d_ref| / (depth_max - depth_min) <= 1e-3   for each of the 13 outputs, p99 <= 5e-3
    (max is not gated: floor / clamp / out-of-bounds discontinuities), confidence mean abs <= 1e-3.

The reference's own fp32-vs-fp64 noise in these units is 3e-6 .. 8e-6, so a healthy build lands near 1e-5.
"""
import pytest
import torch

from common import build_model, check_close, conv_tol, load_golden, t
from effi_mvs_plus_amd import synth

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
RANGE = synth.DEPTH_MAX_MM - synth.DEPTH_MIN_MM


def _norm_err(got, want):
    d = (got.detach().double().cpu() - want.double()).abs() / RANGE
    return d.mean().item(), torch.quantile(d.flatten(), 0.99).item(), d.max().item()


def _features_on_cpu(sd, imgs):
    from oracle import effi_oracle as O
    with torch.no_grad():
        feats = [O.feature_net(sd, "feature", imgs[:, v]) for v in range(imgs.size(1))]
        ctx = O.feature_net(sd, "cnet_depth", imgs[:, 0])
    return feats, ctx


@pytest.mark.parametrize("tag", ["small", "mid"])
def test_cascade_vs_golden(tag, precision):
    g = load_golden(f"g11_full_{tag}.npz")
    H, W, N = int(g["H"]), int(g["W"]), int(g["N"])
    nd = ",".join(str(int(x)) for x in g["ndepths"])
    net, sd = build_model(nd, seed=int(g["weight_seed"]), device=DEV)
    imgs, pm, dv = synth.synth_sample(H, W, N, seed=int(g["img_seed"]))
    feats, ctx = _features_on_cpu(sd, imgs)          # identical features on both sides: isolates the hot path
    with torch.no_grad():
        out = net.forward_hot([{k: t(v, DEV) for k, v in f.items()} for f in feats], {k: t(v, DEV) for k, v in ctx.items()},
                              {k: t(v, DEV) for k, v in pm.items()}, t(dv, DEV), want_intermediates=True)
    assert len(out["depth"]) == 13
    for k in ("view_weights", "reg_volume1", "cur_volume1", "reg_volume2", "cur_volume2", "reg_volume3", "cur_volume3"):
        if "inter_" + k in g:
            check_close(f"[{tag}] {k}", out["intermediates"][k], g["inter_" + k], rtol=1e-3, atol=1e-3, frac_ok=0.99)
    worst = 0.0
    for i, d in enumerate(out["depth"]):
        want = g[f"depth{i:02d}"]
        assert tuple(d.shape) == tuple(want.shape)
        mean, p99, mx = _norm_err(d, want)
        print(f"[cascade {tag}] depth[{i:2d}] {tuple(d.shape)} normalised err: mean={mean:.3e} p99={p99:.3e} max={mx:.3e}")
        assert mean <= 1e-3 and p99 <= 5e-3, f"depth[{i}] outside the stated fp32 tolerance"
        worst = max(worst, mean)
    conf_err = (out["photometric_confidence"].cpu() - g["photometric_confidence"]).abs().mean().item()
    print(f"[cascade {tag}] confidence mean abs err = {conf_err:.3e}; worst depth mean = {worst:.3e}")
    assert conf_err <= 1e-3


def test_cascade_vs_oracle_and_fp64_noise_floor(precision):
    """The HIP result is as close to the fp64 evaluation of the reference graph as the reference's own
    fp32 result is (so the remaining difference is rounding, not arithmetic)."""
    from oracle import effi_oracle as O
    net, sd = build_model("48,8,8", seed=3, device=DEV)
    imgs, pm, dv = synth.synth_sample(192, 256, 4, seed=11)
    feats, ctx = _features_on_cpu(sd, imgs)
    with torch.no_grad():
        ora32 = O.hot_path(sd, feats, ctx, pm, dv)
        sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
        ora64 = O.hot_path(sd64, [{k: v.double() for k, v in f.items()} for f in feats], {k: v.double() for k, v in ctx.items()},
                           {k: v.double() for k, v in pm.items()}, dv.double())
        out = net.forward_hot([{k: t(v, DEV) for k, v in f.items()} for f in feats], {k: t(v, DEV) for k, v in ctx.items()},
                              {k: t(v, DEV) for k, v in pm.items()}, t(dv, DEV))
    for i in (0, 4, 8, 12):
        hip64 = _norm_err(out["depth"][i], ora64["depth"][i])[0]
        ref64 = _norm_err(ora32["depth"][i], ora64["depth"][i])[0]
        hip32 = _norm_err(out["depth"][i], ora32["depth"][i])[0]
        print(f"[noise floor {precision}] depth[{i:2d}] mean normalised err: hip-vs-fp64={hip64:.3e} ref32-vs-fp64={ref64:.3e} hip-vs-ref32={hip32:.3e}")
        assert hip32 <= 1e-3
        assert hip64 <= max(10 * ref64, 1e-4)


def test_cascade_vs_oracle_large_tiles(precision):
    """768x1024 image: stage 3 (384x512) selects the 4-rows-per-wave conv tiling, stage 2 the 2-row one."""
    from oracle import effi_oracle as O
    net, sd = build_model("16,8,8", seed=9, device=DEV)
    imgs, pm, dv = synth.synth_sample(768, 1024, 3, seed=2)
    feats, ctx = _features_on_cpu(sd, imgs)
    with torch.no_grad():
        want = O.hot_path(sd, feats, ctx, pm, dv, ndepths=(16, 8, 8))
        out = net.forward_hot([{k: t(v, DEV) for k, v in f.items()} for f in feats], {k: t(v, DEV) for k, v in ctx.items()},
                              {k: t(v, DEV) for k, v in pm.items()}, t(dv, DEV))
    for i, d in enumerate(out["depth"]):
        mean, p99, mx = _norm_err(d, want["depth"][i])
        print(f"[cascade large {precision}] depth[{i:2d}] {tuple(d.shape)} normalised err: mean={mean:.3e} p99={p99:.3e} max={mx:.3e}")
        assert mean <= 1e-3 and p99 <= 5e-3
    assert (out["photometric_confidence"].cpu() - want["photometric_confidence"]).abs().mean() <= 1e-3


def test_switchable_kernel_forms_agree():
    """The A/B switches of the update block's fp32-map form (ops.set_sr(False): the split-resident form has one path) select other
    kernels for the same arithmetic: enc_tail = 1 (encoder tail in one kernel, intermediate maps in LDS) is bitwise the default two
    launches; head_taps = 0 (depth head as conv1 -> hidden map -> one-channel 3x3) differs from the default tap-projected head only by
    the summation order of conv2 (a tenth of the parity gates)."""
    from effi_mvs_plus_amd import ops
    net, sd = build_model("16,8,8", seed=4, device=DEV)
    imgs, pm, dv = synth.synth_sample(256, 320, 3, seed=5)
    feats, ctx = _features_on_cpu(sd, imgs)
    args = ([{k: t(v, DEV) for k, v in f.items()} for f in feats], {k: t(v, DEV) for k, v in ctx.items()},
            {k: t(v, DEV) for k, v in pm.items()}, t(dv, DEV))

    def run(**opts):
        ops.set_sr(False)
        try:
            with ops.options(**opts), torch.no_grad():
                return [d.clone() for d in net.forward_hot(*args)["depth"]]
        finally:
            ops.set_sr(True)

    base = run()
    tail = run(enc_tail=1)
    for i, (a, b) in enumerate(zip(base, tail)):
        assert torch.equal(a, b), f"depth[{i}] differs with the one-kernel encoder tail"
    two = run(head_taps=0)
    for i, (a, b) in enumerate(zip(base, two)):
        mean, p99, mx = _norm_err(a, b.cpu())
        assert mean <= 1e-4 and p99 <= 1e-3, (i, mean, p99, mx)       # rounding-level differences grow through nine rough-depth iterations


# ---- the configurations BASELINE.json names, at their own sizes (SURVEY.md section 8(d) "Configs restated") -----------------------
FULL_SIZE = {
    "cfg2 800x576 S=4 48,8,8": (576, 800, 5, "48,8,8"),
    "cfg2 800x576 S=4 48,32,8": (576, 800, 5, "48,32,8"),
    "cfg3 1600x1184 S=4 48,8,8": (1184, 1600, 5, "48,8,8"),
    "cfg4 1920x1056 S=6 96,8,8": (1056, 1920, 7, "96,8,8"),
}
_FULL_CACHE = {}


def _full_size_case(name):
    """Inputs and the oracle's outputs of one full-size configuration (computed once per session: the oracle pass takes
    2.6 / 6 / 9 s per view on the box's host cores, both precisions of the HIP path are checked against the same pass)."""
    if name not in _FULL_CACHE:
        from oracle import effi_oracle as O
        H, W, N, nd = FULL_SIZE[name]
        net, sd = build_model(nd, seed=1, device=DEV)
        imgs, pm, dv = synth.synth_sample(H, W, N, seed=0)
        feats, ctx = _features_on_cpu(sd, imgs)
        with torch.no_grad():
            want = O.hot_path(sd, feats, ctx, pm, dv, ndepths=tuple(int(x) for x in nd.split(",")))
        _FULL_CACHE[name] = (net, feats, ctx, pm, dv, want)
    return _FULL_CACHE[name]


@pytest.mark.parametrize("name", list(FULL_SIZE))
def test_full_size_cascade_vs_oracle(name, precision):
    """The HIP path against the CPU oracle on the SAME inputs at the sizes the benchmark configurations name: 13 depth maps
    with normalised mean <= 1e-3 and p99 <= 5e-3, confidence mean abs <= 1e-3 (the gates of SURVEY.md section 8(d))."""
    net, feats, ctx, pm, dv, want = _full_size_case(name)
    with torch.no_grad():
        out = net.forward_hot([{k: t(v, DEV) for k, v in f.items()} for f in feats], {k: t(v, DEV) for k, v in ctx.items()},
                              {k: t(v, DEV) for k, v in pm.items()}, t(dv, DEV))
    assert len(out["depth"]) == 13
    worst = (0.0, 0.0)
    for i, d in enumerate(out["depth"]):
        assert tuple(d.shape) == tuple(want["depth"][i].shape)
        mean, p99, mx = _norm_err(d, want["depth"][i])
        print(f"[full size | {name} | {precision}] depth[{i:2d}] {tuple(d.shape)} normalised err: mean={mean:.3e} p99={p99:.3e} max={mx:.3e}")
        assert mean <= 1e-3 and p99 <= 5e-3, f"{name}: depth[{i}] outside the stated fp32 tolerance"
        worst = (max(worst[0], mean), max(worst[1], p99))
    conf_err = (out["photometric_confidence"].cpu() - want["photometric_confidence"]).abs().mean().item()
    print(f"[full size | {name} | {precision}] worst depth mean={worst[0]:.3e} p99={worst[1]:.3e}; confidence mean abs err={conf_err:.3e}")
    assert conf_err <= 1e-3


@pytest.mark.parametrize("name", ["cfg2 800x576 S=4 48,8,8", "cfg3 1600x1184 S=4 48,8,8"])
def test_bf16_operand_mode_within_its_own_tolerance(name):
    """BASELINE.json's "bf16 (MFMA 3D-conv path)" configuration: the split-precision kernels with plain bf16 operands (hi*hi only,
    fp32 accumulation; ops.set_precision("bf16")).  Not fp32-grade: its stated tolerance is a normalised mean depth error <= 1e-2 per
    output; the default mode is held to 1e-3 by the tests above."""
    from effi_mvs_plus_amd import ops
    net, feats, ctx, pm, dv, want = _full_size_case(name)
    before = ops.get_precision()
    ops.set_precision("bf16")
    try:
        with torch.no_grad():
            out = net.forward_hot([{k: t(v, DEV) for k, v in f.items()} for f in feats], {k: t(v, DEV) for k, v in ctx.items()},
                                  {k: t(v, DEV) for k, v in pm.items()}, t(dv, DEV))
    finally:
        ops.set_precision(before)
    assert ops.get_precision() == before
    worst = 0.0
    for i, d in enumerate(out["depth"]):
        mean, p99, mx = _norm_err(d, want["depth"][i])
        print(f"[bf16 operands | {name}] depth[{i:2d}] normalised err: mean={mean:.3e} p99={p99:.3e} max={mx:.3e}")
        # "bf16 variant: final depth normalised mean <= 1e-2"; the twelve intermediate maps are held to 2e-2
        # (measured at 1600x1184: final 3.7e-3, worst intermediate 1.0e-2)
        assert mean <= (1e-2 if i == 12 else 2e-2), f"{name}: depth[{i}] outside the bf16 variant's tolerance"
        worst = max(worst, mean)
    assert worst > 1e-6, "bf16 operands must differ from the fp32-grade result (is the mode switch wired?)"


def test_data_parallel_wrapper_and_foreign_current_device():
    """The reference's DTU driver wraps the model in nn.DataParallel (test_dtu_dypcd.py:418): the wrapped model must give the
    unwrapped model's result.  (On a one-GPU box DataParallel has a single replica; the per-device workspace and the
    current-device guard of ops._t are what make more replicas safe -- tests/test_host_logic.py covers the registry.)"""
    import torch.nn as nn
    from effi_mvs_plus_amd import ops
    g = load_golden("g11_full_small.npz")
    net, sd = build_model("8,8,8", seed=int(g["weight_seed"]), device=DEV)
    imgs, pm, dv = synth.synth_sample(int(g["H"]), int(g["W"]), int(g["N"]), seed=int(g["img_seed"]))
    args = (imgs.to(DEV), {k: v.to(DEV) for k, v in pm.items()}, dv.to(DEV))
    with torch.no_grad():
        want = net(*args)
        dp = nn.DataParallel(net)
        dp.eval()
        got = dp(*args)
    for a, b in zip(got["depth"], want["depth"]):
        assert torch.equal(a, b)
    assert torch.equal(got["photometric_confidence"], want["photometric_confidence"])
    assert 0 in ops._WORKSPACES and ops._WORKSPACES[0].numel() * 4 >= 256 and float(ops._WORKSPACES[0].abs().sum()) == 0.0
    from effi_mvs_plus_amd import _lib
    assert _lib.lib().effi_get_workspace(0) == ops._WORKSPACES[0].data_ptr()


def test_batch_of_two_and_48_32_8_cascade():
    """B = 2 (the host loops over the batch) and BASELINE.json's 48/32/8 hypothesis counts."""
    from oracle import effi_oracle as O
    net, sd = build_model("48,32,8", seed=4, device=DEV)
    a = synth.synth_sample(128, 192, 3, seed=21)
    b = synth.synth_sample(128, 192, 3, seed=22)
    imgs = torch.cat([a[0], b[0]])
    pm = {k: torch.cat([a[1][k], a[1][k]]) for k in a[1]}
    dv = torch.cat([a[2], a[2] * 1.1])                       # different depth ranges per sample
    feats, ctx = _features_on_cpu(sd, imgs)
    with torch.no_grad():
        want = O.hot_path(sd, feats, ctx, pm, dv, ndepths=(48, 32, 8))
        out = net.forward_hot([{k: t(v, DEV) for k, v in f.items()} for f in feats], {k: t(v, DEV) for k, v in ctx.items()},
                              {k: t(v, DEV) for k, v in pm.items()}, t(dv, DEV))
    assert out["depth"][-1].shape == (2, 128, 192) and out["photometric_confidence"].shape == (2, 64, 96)
    for i, d in enumerate(out["depth"]):
        mean, p99, _ = _norm_err(d, want["depth"][i])
        assert mean <= 1e-3 and p99 <= 5e-3, (i, mean, p99)


def test_tanks_and_temples_shaped_cascade():
    """cfg4-like: 6 source views, 96 first-stage hypotheses (test_tank.sh:14-15), reduced resolution."""
    from oracle import effi_oracle as O
    net, sd = build_model("96,8,8", seed=6, device=DEV)
    imgs, pm, dv = synth.synth_sample(160, 256, 7, seed=5)
    feats, ctx = _features_on_cpu(sd, imgs)
    with torch.no_grad():
        want = O.hot_path(sd, feats, ctx, pm, dv, ndepths=(96, 8, 8))
        out = net.forward_hot([{k: t(v, DEV) for k, v in f.items()} for f in feats], {k: t(v, DEV) for k, v in ctx.items()},
                              {k: t(v, DEV) for k, v in pm.items()}, t(dv, DEV))
    for i, d in enumerate(out["depth"]):
        mean, p99, _ = _norm_err(d, want["depth"][i])
        assert mean <= 1e-3 and p99 <= 5e-3, (i, mean, p99)


@pytest.mark.parametrize("H,W", [(64, 96), (256, 320), (1184, 1600)])
def test_feature_pyramid_on_hip(H, W, precision):
    """Scope row n1: P_1to8_FeatureNet_Fast (feature net and context net) on the MFMA conv kernels vs the oracle."""
    from oracle import effi_oracle as O
    net, sd = build_model("8,8,8", seed=8, device=DEV)
    g = torch.Generator().manual_seed(H)
    img = torch.rand(1, 3, H, W, generator=g)
    with torch.no_grad():
        for name, mod in (("feature", net.feature), ("cnet_depth", net.cnet_depth)):
            want = O.feature_net(sd, name, img)
            got = mod(img.to(DEV))
            ref_torch = mod.forward_torch(img.to(DEV))
            for k in ("stage1", "stage2", "stage3"):
                assert tuple(got[k].shape) == tuple(want[k].shape)
                check_close(f"FPN {name}.{k} {H}x{W}", got[k], want[k], **conv_tol(precision, want[k], 1e-4, 2e-5, layers=10))
                check_close(f"FPN {name}.{k} {H}x{W} (stock torch on GPU)", ref_torch[k], want[k], rtol=1e-3, atol=1e-4)


def test_full_forward_including_fpn():
    """model(imgs, proj_matrices, depth_values) exactly as the reference's drivers call it
    (test_dtu_dypcd.py:439); FPN runs in stock PyTorch-ROCm, so features differ by MIOpen rounding."""
    g = load_golden("g11_full_small.npz")
    net, sd = build_model("8,8,8", seed=int(g["weight_seed"]), device=DEV)
    imgs, pm, dv = synth.synth_sample(int(g["H"]), int(g["W"]), int(g["N"]), seed=int(g["img_seed"]))
    with torch.no_grad():
        out = net(imgs.to(DEV), {k: v.to(DEV) for k, v in pm.items()}, dv.to(DEV))
    assert set(out) == {"depth", "photometric_confidence"} and len(out["depth"]) == 13
    for i, d in enumerate(out["depth"]):
        mean, p99, _ = _norm_err(d, g[f"depth{i:02d}"])
        assert mean <= 1e-3 and p99 <= 5e-3, (i, mean, p99)
    assert tuple(out["photometric_confidence"].shape) == tuple(g["photometric_confidence"].shape)


def test_hip_graph_replay_is_bitwise_equal_to_eager(precision):
    """effi_mvs_plus_amd.graph.HotPathGraph: capture once, replay with fresh inputs copied into the static buffers."""
    from effi_mvs_plus_amd.graph import HotPathGraph
    net, sd = build_model("8,8,8", seed=4, device=DEV)
    samples = []
    with torch.no_grad():
        for seed in (21, 22):
            imgs, pm, dv = synth.synth_sample(128, 160, 3, seed=seed)
            imgs = imgs.to(DEV)
            feats = [net.feature(imgs[:, v]) for v in range(3)]
            ctx = net.cnet_depth(imgs[:, 0])
            samples.append((feats, ctx, {k: v.to(DEV) for k, v in pm.items()}, dv.to(DEV)))
        g = HotPathGraph(net, *samples[0])
        for smp in (samples[1], samples[0], samples[1]):
            want = net.forward_hot(*smp)
            got = g(*smp)
            assert len(got["depth"]) == 13
            for a, b in zip(got["depth"], want["depth"]):
                assert torch.equal(a, b)
            assert torch.equal(got["photometric_confidence"], want["photometric_confidence"])
        with pytest.raises(ValueError):
            g(samples[0][0][:2], *samples[0][1:])


def test_views_in_flight_on_two_stream_graphs_are_bitwise_equal_to_eager():
    """The default mode of bench.py: one hipGraph per input slot captured with the pass's two internal streams, three views in flight
    on three streams, replayed round-robin several times over -- every view's 13 depth maps and confidence equal the eager
    single-stream pass bit for bit (no scratch shared between slots, no ordering assumed between views)."""
    from effi_mvs_plus_amd import ops
    from effi_mvs_plus_amd.graph import HotPathGraph
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
        ops.set_branches(True)
        try:
            g = HotPathGraph(net, *samples[0], slots=3)
        finally:
            ops.set_branches(False)
        for i, smp in enumerate(samples):
            g.load(i, *smp)
        torch.cuda.synchronize()
        lanes = [torch.cuda.Stream() for _ in range(3)]
        cur = torch.cuda.current_stream()
        for st in lanes:
            st.wait_stream(cur)
        bad = {}
        for _ in range(25):                   # 300 replays: the lane-exchange form of the stage-2/3 warp kernel failed 1-5 % of them
            kept = []
            for i in range(12):
                with torch.cuda.stream(lanes[i % 3]):
                    out = g.replay(i % 3)
                    kept.append((i % 3, [d.clone() for d in out["depth"]]))
            for st in lanes:
                cur.wait_stream(st)
            torch.cuda.synchronize()
            for slot, depths in kept:
                for k, (a, b) in enumerate(zip(depths, want[slot])):
                    if not torch.equal(a, b):
                        bad[(slot, k)] = bad.get((slot, k), 0) + 1
            for st in lanes:
                st.wait_stream(cur)
    assert not bad, f"(slot, depth index) -> replays that differ from the eager pass: {bad}"


def test_cpu_tensors_are_refused():
    from effi_mvs_plus_amd._lib import EffiLibraryError
    net, _ = build_model("8,8,8", seed=1, device=DEV)
    imgs, pm, dv = synth.synth_sample(64, 96, 3, seed=0)
    net.eval().cpu()
    with pytest.raises(EffiLibraryError):
        net(imgs, pm, dv)
    net.train()
    with pytest.raises(EffiLibraryError):                # the training path has no CPU fallback either
        net(imgs, pm, dv)


def test_size_independent_properties_at_full_resolution():
    """cfg3 shape (1600x1184, S=4, 48,8,8): too large for the CPU oracle in a test, so check properties
    the domain offers: determinism, finite outputs inside the depth range, hypotheses ordered, and
    invariance of the cost volume to a permutation of the source views (aggregation is a weighted mean)."""
    from effi_mvs_plus_amd import ops
    net, sd = build_model("48,8,8", seed=5, device=DEV)
    H, W, N = 1184, 1600, 5
    g = torch.Generator().manual_seed(0)
    feats = []
    for v in range(N):
        feats.append({"stage1": torch.randn(1, 32, H // 8, W // 8, generator=g).to(DEV),
                      "stage2": torch.randn(1, 16, H // 4, W // 4, generator=g).to(DEV),
                      "stage3": torch.randn(1, 8, H // 2, W // 2, generator=g).to(DEV)})
    ctx = {"stage1": torch.randn(1, 60, H // 8, W // 8, generator=g).to(DEV),
           "stage2": torch.randn(1, 40, H // 4, W // 4, generator=g).to(DEV),
           "stage3": torch.randn(1, 20, H // 2, W // 2, generator=g).to(DEV)}
    _, pm, dv = synth.synth_sample(32, 32, N, seed=0)
    pm = synth.synth_cameras(H, W, N)
    pm = {k: v.to(DEV) for k, v in pm.items()}
    dv = dv.to(DEV)
    with torch.no_grad():
        a = net.forward_hot(feats, ctx, pm, dv)
        b = net.forward_hot(feats, ctx, pm, dv)
    assert [tuple(d.shape) for d in a["depth"]] == [(1, 148, 200)] * 4 + [(1, 296, 400)] * 4 + [(1, 592, 800)] * 4 + [(1, 1184, 1600)]
    for x, y in zip(a["depth"], b["depth"]):
        assert torch.equal(x, y), "the path must be deterministic (no atomics, fixed reduction order)"
        assert torch.isfinite(x).all()
    d0 = a["depth"][0]
    assert d0.min() >= synth.DEPTH_MIN_MM - 1e-2 and d0.max() <= synth.DEPTH_MAX_MM + 1e-2   # convex combination of hypotheses
    assert a["photometric_confidence"].min() >= 0 and a["photometric_confidence"].max() <= 1 + 1e-5
    # permutation invariance of the stage-2 dynamic volume w.r.t. source-view order
    maps = [f["stage2"][0] for f in feats]
    nhwc = ops.to_nhwc(maps)
    rt = ops.compose_rel_proj(pm["stage2"][0].contiguous())
    cur = a["depth"][4][0].contiguous()
    itv = torch.full((1,), 1e-5, device=DEV)
    vw = torch.rand(4, 148, 200, device=DEV)
    s1, smp = ops.warpcorr_dyn(nhwc[0], nhwc[1:], rt, cur, itv, vw, 8)
    perm = [2, 0, 3, 1]
    s2, _ = ops.warpcorr_dyn(nhwc[0], [nhwc[1 + p] for p in perm], rt[perm].contiguous(), cur, itv, vw[perm].contiguous(), 8)
    check_close("view-permutation invariance", s2, s1, rtol=1e-4, atol=1e-5)
    assert (smp[:-1] >= smp[1:]).all(), "depth hypotheses must be ordered far -> near (ascending inverse depth)"
