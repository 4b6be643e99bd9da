"""Scope row n3: dynamic geometric-consistency filter + depth averaging of the T&T driver
(misc/fusion.py:117-181, test_tank.py:466-512).

CPU: the oracle restatement against the golden vectors produced by running the reference's own functions
(tests/golden/make_golden_fusion.py).  GPU: the fused HIP kernel through the C ABI against the same vectors and the oracle.
Masks are thresholded comparisons, so a pixel whose reprojection error sits within float rounding of a threshold may flip:
continuous outputs are held to tolerances, masks to an agreement fraction (and every disagreement must be such a boundary case).
"""
import pytest
import torch

from common import check_close, load_golden, t
from effi_mvs_plus_amd import synth
from oracle import effi_oracle as O

DEV = "cuda:0"
CASES = ["a", "b"]


def _inputs(g, tag):
    c = {k: g[f"{tag}_{k}"] for k in ("H", "W", "N", "seed", "prob", "dh", "dist", "dfilt", "relative")}
    c = {k: (v.item() if hasattr(v, "item") else v) for k, v in c.items()}
    d, cams = synth.synth_depth_maps(int(c["H"]), int(c["W"]), int(c["N"]), seed=int(c["seed"]))
    gen = torch.Generator().manual_seed(int(c["seed"]) + 100)
    conf = torch.rand(1, 2 * int(c["H"]), 2 * int(c["W"]), generator=gen)
    return c, d[0][None, None], d[1:][None, :, None], cams[0][None], cams[1:][None], conf


def test_oracle_elementwise_products_equal_the_batched_matmul_form():
    """bench.py's PyTorch-ROCm fusion baseline evaluates the per-pixel 3x3 / 4x4 products element-wise (no 19 M-batch GEMM);
    it must be the same function as the reference's broadcast ``@`` form that the golden vectors pin."""
    g = load_golden("g12_fusion.npz")
    c, ref_depth, src, ref_cam, src_cams, conf = _inputs(g, "a")
    args = (ref_depth, src, ref_cam, src_cams, conf, float(c["prob"]), int(c["dh"]), float(c["dist"]), float(c["dfilt"]), bool(c["relative"]))
    with torch.no_grad():
        a = O.fusion_dynamic_filter(*args)
        with O.elementwise_mm():
            b = O.fusion_dynamic_filter(*args)
    assert not O.ELEMENTWISE_MM
    assert torch.allclose(a["reproj_xyd"], b["reproj_xyd"], rtol=1e-5, atol=2e-3)
    assert torch.allclose(a["depth"], b["depth"], rtol=1e-6, atol=1e-3)
    assert (a["mask"] == b["mask"]).float().mean().item() >= 0.999


def test_literal_syncs_do_not_change_values():
    """The oracle's literal mode re-enables the reference's NaN probe and torch.unique assert (34 host syncs per view on a GPU):
    same values, counted."""
    from common import build_model
    net, sd = build_model("8,8,8", seed=3)
    imgs, pm, dv = synth.synth_sample(64, 96, 4, seed=2)
    with torch.no_grad():
        feats = [O.feature_net(sd, "feature", imgs[:, v]) for v in range(4)]
        ctx = O.feature_net(sd, "cnet_depth", imgs[:, 0])
        a = O.hot_path(sd, feats, ctx, pm, dv, ndepths=(8, 8, 8))
        with O.literal_syncs():
            b = O.hot_path(sd, feats, ctx, pm, dv, ndepths=(8, 8, 8))
        assert O.SYNC_COUNT == 3 * 3 + 22 and not O.LITERAL_SYNCS      # S = 3 sources x 3 stages warps + 22 lookups
    for x, y in zip(a["depth"], b["depth"]):
        assert torch.equal(x, y)


@pytest.mark.parametrize("tag", CASES)
def test_oracle_matches_reference_vectors(tag):
    g = load_golden("g12_fusion.npz")
    c, ref_depth, src, ref_cam, src_cams, conf = _inputs(g, tag)
    with torch.no_grad():
        out = O.fusion_dynamic_filter(ref_depth, src, ref_cam, src_cams, conf, float(c["prob"]), int(c["dh"]), float(c["dist"]),
                                      float(c["dfilt"]), bool(c["relative"]))
    assert torch.allclose(out["reproj_xyd"], g[f"{tag}_reproj_xyd"], rtol=1e-5, atol=1e-4)
    assert torch.allclose(out["depth"], g[f"{tag}_depth"], rtol=1e-6, atol=1e-4)
    assert torch.allclose(out["points"], g[f"{tag}_points"], rtol=1e-5, atol=1e-3)
    for k in ("geo_mask", "prob_mask", "mask"):
        agree = (out[k].to(torch.uint8) == g[f"{tag}_{k}"]).float().mean().item()
        assert agree >= 0.9999, (k, agree)


@pytest.mark.gpu
@pytest.mark.parametrize("tag", CASES)
def test_hip_filter_against_reference_vectors_and_oracle(tag):
    from effi_mvs_plus_amd import fusion, ops
    g = load_golden("g12_fusion.npz")
    c, ref_depth, src, ref_cam, src_cams, conf = _inputs(g, tag)
    r = ops.fusion_dynamic_filter(t(ref_depth[0, 0], DEV), t(src[0, :, 0], DEV), t(ref_cam[0], DEV), t(src_cams[0], DEV),
                                  t(conf[0], DEV), float(c["prob"]), int(c["dh"]), float(c["dist"]), float(c["dfilt"]),
                                  bool(c["relative"]), want_points=True, want_reproj=True)
    # continuous outputs: reprojected coordinates to 2e-3 px / 2e-3 mm (two projections through fp32 4x4 algebra at |X| ~ 700)
    check_close(f"[{tag}] reproj_xyd", r["reproj_xyd"], g[f"{tag}_reproj_xyd"][0], rtol=2e-6, atol=2e-3)
    gd = g[f"{tag}_depth"][0, 0]
    # averaged depth: equal wherever the set of views that passed the loosest threshold is the same (boundary flips change the set)
    d_err = (r["depth"].cpu() - gd).abs()
    frac_close = (d_err <= 2e-3).float().mean().item()
    print(f"[{tag}] averaged depth within 2e-3 mm: {frac_close:.6f}, max {d_err.max():.3e}")
    assert frac_close >= 0.999
    for k in ("geo_mask", "prob_mask", "mask"):
        agree = (r[k].cpu() == g[f"{tag}_{k}"][0, 0]).float().mean().item()
        print(f"[{tag}] {k} agreement with the reference: {agree:.6f}")
        assert agree >= 0.999, (k, agree)
    ok = d_err <= 2e-3
    p_err = (r["points"].cpu() - g[f"{tag}_points"][0]).abs().max(dim=0).values
    assert (p_err[ok] <= 5e-3).float().mean().item() >= 0.9999
    # batched API mirror (names of misc/fusion.py)
    out = fusion.dynamic_filter(t(ref_depth, DEV), t(src, DEV), t(ref_cam, DEV), t(src_cams, DEV), t(conf, DEV), float(c["prob"]),
                                int(c["dh"]), float(c["dist"]), float(c["dfilt"]), bool(c["relative"]))
    assert tuple(out["depth"].shape) == tuple(g[f"{tag}_depth"].shape) and out["mask"].dtype == torch.bool
    assert torch.equal(out["depth"][0, 0], r["depth"])
    xyd, ref_pts, back_pts = fusion.get_reproj_dynamic(t(ref_depth, DEV), t(src, DEV), t(ref_cam, DEV), t(src_cams, DEV))
    assert torch.equal(xyd[0], r["reproj_xyd"])
    # the two by-product point tensors of misc/fusion.py:117-156 (same shapes; values to fp32 rounding of the element-wise algebra)
    with torch.no_grad():
        _, want_ref, want_back = O.fusion_get_reproj_dynamic(ref_depth, src, ref_cam, src_cams)
    assert tuple(ref_pts.shape) == tuple(want_ref.shape) and tuple(back_pts.shape) == tuple(want_back.shape)
    check_close(f"{tag} ref_idx_cam", ref_pts, want_ref, rtol=1e-5, atol=1e-3)
    ok = torch.isfinite(want_back).all(-1).all(-1) & (want_back[..., 2, 0].abs() > 1.0)          # zero-padded samples: 0 * 1/(0+1e-9)
    check_close(f"{tag} src2ref_idx_cam", back_pts.cpu()[ok], want_back[ok], rtol=1e-4, atol=5e-2, frac_ok=0.999)


@pytest.mark.gpu
def test_hip_filter_edge_cases():
    """All-consistent views, a view that projects outside the image (zero-padded sample), no confidence map, 16 views."""
    from effi_mvs_plus_amd import ops
    from effi_mvs_plus_amd._lib import EffiLibraryError
    d, cams = synth.synth_depth_maps(40, 56, 17, seed=9, noise_mm=0.0, outlier_frac=0.0)
    cams = cams.clone()
    cams[3, 0, 0, 3] += 4000.0                       # source view 3 shifted far sideways: every sample falls outside
    with torch.no_grad():
        want = O.fusion_dynamic_filter(d[0][None, None], d[1:][None, :, None], cams[0][None], cams[1:][None],
                                       torch.ones(1, 40, 56), -1.0, 2, 4.0, 1.3)
    r = ops.fusion_dynamic_filter(t(d[0], DEV), t(d[1:], DEV), t(cams[0], DEV), t(cams[1:], DEV), None, 0.0, 2, 4.0, 1.3)
    assert (r["prob_mask"] == 1).all()
    agree = (r["geo_mask"].cpu() == want["geo_mask"][0, 0].to(torch.uint8)).float().mean().item()
    assert agree >= 0.999
    check_close("edge averaged depth", r["depth"], want["depth"][0, 0], rtol=1e-5, atol=2e-3, frac_ok=0.999)
    with pytest.raises(EffiLibraryError):            # more than 16 source views
        dd, cc = synth.synth_depth_maps(16, 16, 18, seed=1)
        ops.fusion_dynamic_filter(t(dd[0], DEV), t(dd[1:], DEV), t(cc[0], DEV), t(cc[1:], DEV))
