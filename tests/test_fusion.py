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
@pytest.mark.parametrize("tag", CASES)
def test_tank_driver_lines_run_unchanged_on_the_hip_fusion_module(tag):
    """The tensor lines of the reference's T&T driver, /root/reference/test_tank.py:486-509, typed here as the driver has them with
    ``fusion`` = effi_mvs_plus_amd.fusion (the maintainer's one-line change: ``from effi_mvs_plus_amd import fusion``): every
    ``fusion.*`` name the driver calls exists with the reference's signature, and the block reproduces the golden vectors that
    tests/golden/make_golden_fusion.py produced by running the same lines on the reference's misc/fusion.py."""
    import torch.nn.functional as F
    from effi_mvs_plus_amd import fusion
    g = load_golden("g12_fusion.npz")
    c, ref_depth, src, ref_cam, src_cams, conf = _inputs(g, tag)
    sample = {"ref_depth": t(ref_depth, DEV), "src_depths": t(src, DEV), "ref_cam": t(ref_cam, DEV), "src_cams": t(src_cams, DEV),
              "ref_conf": t(conf, DEV)}
    filter_dixt = {"prob_threshold": float(c["prob"]), "dh_view_num": int(c["dh"]), "dist_filter": float(c["dist"]),
                   "depth_filter": float(c["dfilt"])}
    relative = bool(c["relative"])
    prob_threshold = filter_dixt["prob_threshold"]
    # ---- test_tank.py:467-509 ----
    num_src_views = sample['src_depths'].shape[1]
    dy_range = num_src_views + 1
    h, w = sample['ref_depth'].shape[-2:]
    sample['ref_conf'] = F.interpolate(sample['ref_conf'].unsqueeze(1), size=[h, w], mode='nearest')
    prob_mask = sample['ref_conf'] > prob_threshold
    prob_mask = prob_mask.squeeze(1)
    ref_depth_d = sample['ref_depth']
    reproj_xyd, ref_idx_cam, src2ref_idx_cam = fusion.get_reproj_dynamic(
        *[sample[attr] for attr in ['ref_depth', 'src_depths', 'ref_cam', 'src_cams']])
    dh_view_num = filter_dixt["dh_view_num"]
    vis_masks, vis_mask = fusion.vis_filter_dynamic(sample['ref_depth'], reproj_xyd, ref_idx_cam, src2ref_idx_cam, dist_base=filter_dixt["dist_filter"],
                                                    rel_diff_base=filter_dixt["depth_filter"], thres_view=dh_view_num, relative=relative)
    reproj_depth = reproj_xyd[:, :, -1]
    assert vis_mask.shape[2] != 0
    reproj_depth[~vis_mask.squeeze(2)] = 0
    geo_mask_sums = vis_masks.sum(dim=1)
    geo_mask_sum = vis_mask.sum(dim=1)
    depth_est_averaged = (torch.sum(reproj_depth, dim=1, keepdim=True) + ref_depth_d) / (geo_mask_sum + 1)
    geo_mask = geo_mask_sum >= dy_range
    for i in range(dh_view_num, dy_range):
        geo_mask = torch.logical_or(geo_mask, geo_mask_sums[:, i - dh_view_num] >= i)
    mask = fusion.bin_op_reduce([prob_mask, geo_mask], torch.min)
    idx_img = fusion.get_pixel_grids(*depth_est_averaged.size()[-2:]).unsqueeze(0)
    idx_cam = fusion.idx_img2cam(idx_img, depth_est_averaged, sample['ref_cam'])
    points = fusion.idx_cam2world(idx_cam, sample['ref_cam'])[..., :3, 0].permute(0, 3, 1, 2)
    # ---- against the reference's outputs for the same lines ----
    assert vis_masks.dtype == torch.bool and tuple(vis_masks.shape) == tuple(g[f"{tag}_vis_masks"].shape)
    assert tuple(vis_mask.shape) == (1, num_src_views, 1, h, w)
    agree = (vis_masks.cpu() == g[f"{tag}_vis_masks"].bool()).float().mean().item()
    print(f"[{tag}] vis_masks agreement with the reference: {agree:.6f}")
    assert agree >= 0.9995                  # a pixel whose reprojection error sits within rounding of a threshold may flip
    d_err = (depth_est_averaged.cpu() - g[f"{tag}_depth"]).abs()
    assert (d_err <= 2e-3).float().mean().item() >= 0.999
    for name, got in (("geo_mask", geo_mask), ("prob_mask", prob_mask), ("mask", mask)):
        a_ = (got.reshape(1, 1, h, w).cpu() == g[f"{tag}_{name}"].bool()).float().mean().item()
        assert a_ >= 0.999, (name, a_)
    ok = (d_err <= 2e-3)[0, 0]
    check_close(f"[{tag}] idx_cam (idx_img2cam)", idx_cam.cpu()[0][ok], g[f"{tag}_idx_cam"][0][ok], rtol=1e-5, atol=5e-3)
    check_close(f"[{tag}] points (idx_cam2world)", points.cpu()[0][:, ok], g[f"{tag}_points"][0][:, ok], rtol=1e-5, atol=5e-3)
    # the two transforms the driver does not call directly, fed with the REFERENCE's intermediate so that each function is checked alone
    gi = t(g[f"{tag}_idx_cam"], DEV)
    world = fusion.idx_cam2world(gi, sample['ref_cam'])
    check_close(f"[{tag}] idx_cam2world alone", world[..., :3, 0].permute(0, 3, 1, 2), g[f"{tag}_points"], rtol=1e-5, atol=2e-3)
    w2c = fusion.idx_world2cam(world, sample['src_cams'][:, 0])
    check_close(f"[{tag}] idx_world2cam", w2c, g[f"{tag}_world2cam_src0"], rtol=1e-5, atol=2e-3)
    check_close(f"[{tag}] idx_cam2img", fusion.idx_cam2img(t(g[f"{tag}_world2cam_src0"], DEV), sample['src_cams'][:, 0]),
                g[f"{tag}_cam2img_src0"], rtol=1e-5, atol=2e-3)
    check_close(f"[{tag}] idx_img2cam alone", fusion.idx_img2cam(idx_img, t(g[f"{tag}_depth"], DEV), sample['ref_cam']), g[f"{tag}_idx_cam"],
                rtol=1e-5, atol=2e-3)
    # vis_filter_dynamic alone on the reference's reproj_xyd: thresholds are exact comparisons of the same fp32 numbers
    vm, _ = fusion.vis_filter_dynamic(sample['ref_depth'], t(g[f"{tag}_reproj_xyd"], DEV), None, None, dist_base=filter_dixt["dist_filter"],
                                      rel_diff_base=filter_dixt["depth_filter"], thres_view=dh_view_num, relative=relative)
    a_ = (vm.cpu() == g[f"{tag}_vis_masks"].bool()).float().mean().item()
    print(f"[{tag}] vis_filter_dynamic on the reference's reproj_xyd: agreement {a_:.7f}")
    assert a_ >= 0.99999


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


# ---- DTU branch (test_dtu_dypcd.py:164-350): PARITY UNPINNED -- cv2 is absent and the reference holds no fixtures; the checker is a
# restatement of the numpy lines (with their dtypes) and of OpenCV's published remap arithmetic (oracle/effi_dtu_filter_oracle.py) -----
def _dtu_case(H, W, N, seed):
    import numpy as np
    d, cams = synth.synth_depth_maps(H, W, N, seed=seed, noise_mm=0.03, outlier_frac=0.08, pixel_center=0.0)
    g = torch.Generator().manual_seed(seed + 7)
    conf = torch.rand(H // 2, W // 2, generator=g)
    K = [cams[v, 1, :3, :3].numpy().astype(np.float32) for v in range(N)]
    E = [cams[v, 0].numpy().astype(np.float32) for v in range(N)]
    return d, cams, conf, K, E


def test_dtu_filter_parity_unpinned_oracle_self_checks():
    """Properties the restated pieces must have whatever OpenCV's build does: remap at integer coordinates returns the pixel, at
    k/32 fractions the table's linear blend, zero outside; identical cameras and depth maps reproject every pixel onto itself."""
    import numpy as np
    from oracle import effi_dtu_filter_oracle as D
    rng = np.random.default_rng(0)
    img = rng.random((7, 9), dtype=np.float32)
    ys, xs = np.meshgrid(np.arange(7, dtype=np.float32), np.arange(9, dtype=np.float32), indexing="ij")
    assert np.array_equal(D.cv_remap_linear(img, xs, ys), img)
    half = D.cv_remap_linear(img, xs[:, :-1] + 0.5, ys[:, :-1])
    assert np.allclose(half, 0.5 * img[:, :-1] + 0.5 * img[:, 1:], rtol=1e-6)
    assert D.cv_remap_linear(img, np.full((1, 1), -3.0, np.float32), np.full((1, 1), 2.0, np.float32))[0, 0] == 0.0
    assert np.isclose(D.cv_remap_linear(img, np.full((1, 1), 8.5, np.float32), np.full((1, 1), 0.0, np.float32))[0, 0], 0.5 * img[0, 8])   # half outside
    # coordinates are quantised to 1/32 pixel: 0.51 samples like 0.5 (16/32), 0.52 like 17/32
    q = D.cv_remap_linear(img, np.array([[0.51, 0.52]], np.float32), np.zeros((1, 2), np.float32))
    assert np.isclose(q[0, 0], 0.5 * img[0, 0] + 0.5 * img[0, 1]) and np.isclose(q[0, 1], (15 / 32) * img[0, 0] + (17 / 32) * img[0, 1])
    d, cams, conf, K, E = _dtu_case(48, 64, 3, seed=1)
    dn = d[0].numpy()
    dep, xr, yr, xs_, ys_ = D.reproject_with_depth(dn, K[0], E[0], dn, K[0], E[0])
    gx, gy = np.meshgrid(np.arange(64), np.arange(48))
    assert np.abs(xr - gx).max() < 1e-3 and np.abs(yr - gy).max() < 1e-3 and np.abs(dep - dn).max() < 1e-3
    out = D.filter_depth_arrays(dn, K[0], E[0], [dn, dn], [K[0]] * 2, [E[0]] * 2, np.ones((24, 32), np.float32))
    assert out["geo_mask"].all() and out["final_mask"].all() and np.allclose(out["depth_est_averaged"], dn)


@pytest.mark.gpu
@pytest.mark.parametrize("H,W,N,seed", [(96, 128, 5, 3), (75, 101, 3, 5), (128, 160, 11, 8)])
def test_dtu_filter_parity_unpinned_kernel_vs_oracle(H, W, N, seed):
    """effi_fusion_dtu_filter_f32 against the numpy restatement: continuous outputs to fp32 rounding, masks to an agreement fraction
    (a pixel whose reprojection error sits within rounding of one of the ten thresholds may flip: the kernel inverts the camera
    matrices in double, the reference in single precision)."""
    import numpy as np
    from effi_mvs_plus_amd import dtu_fusion
    from oracle import effi_dtu_filter_oracle as D
    d, cams, conf, K, E = _dtu_case(H, W, N, seed)
    want = D.filter_depth_arrays(d[0].numpy(), K[0], E[0], [d[v].numpy() for v in range(1, N)], K[1:], E[1:], conf.numpy(), conf=0.5)
    got = dtu_fusion.filter_view(t(d[0], DEV), K[0], E[0], t(d[1:], DEV), K[1:], E[1:], t(conf, DEV), conf=0.5)
    for k in ("photo_mask", "geo_mask", "final_mask"):
        agree = float((got[k].cpu().numpy() == want[k]).mean())
        print(f"[dtu filter {H}x{W} N={N}] {k}: agreement {agree:.5f}, set fraction {want[k].mean():.3f}")
        assert agree >= 0.998, (k, agree)
    assert 0.05 < want["geo_mask"].mean() < 0.999, "the case must exercise both outcomes"
    same = got["geo_mask"].cpu().numpy() == want["geo_mask"]
    dd = np.abs(got["depth_est_averaged"].cpu().numpy() - want["depth_est_averaged"])
    assert dd[same].max() <= 2e-3 and np.median(dd) <= 1e-4, (dd[same].max(), np.median(dd))          # mm, depths ~ 600 mm
    pw = np.abs(got["xyz_world"].cpu().numpy() - want["xyz_world"])
    assert pw[:, same].max() <= 5e-3


@pytest.mark.gpu
@pytest.mark.parametrize("H,W,N,seed", [(96, 128, 3, 3), (75, 101, 3, 5)])
def test_dtu_filter_parity_unpinned_reference_named_functions(H, W, N, seed):
    """``reproject_with_depth`` / ``check_geometric_consistency`` (test_dtu_dypcd.py:164-233) under their own names and argument
    lists (numpy in, numpy out) against the numpy restatement; PARITY UNPINNED like the fused kernel whose arithmetic they share."""
    import numpy as np
    from effi_mvs_plus_amd import dtu_fusion
    from oracle import effi_dtu_filter_oracle as D
    d, cams, conf, K, E = _dtu_case(H, W, N, seed)
    dn = [d[v].numpy() for v in range(N)]
    want = D.reproject_with_depth(dn[0], K[0], E[0], dn[1], K[1], E[1])
    got = dtu_fusion.reproject_with_depth(dn[0], K[0], E[0], dn[1], K[1], E[1])
    assert len(got) == 5 and all(isinstance(a, np.ndarray) and a.dtype == np.float32 and a.shape == (H, W) for a in got)
    names = ("depth_reprojected", "x_reprojected", "y_reprojected", "x_src", "y_src")
    for nm, a, b in zip(names, got, want):
        err = np.abs(a - b)
        print(f"[dtu reproject {H}x{W}] {nm}: max {err.max():.3e} median {np.median(err):.3e}")
        # a 1/32-pixel quantisation step of the remap coordinate that falls the other way moves a sampled depth by a gradient step
        assert np.median(err) <= 1e-4 and (err <= 5e-3).mean() >= 0.999, nm
    wm, wmask, wdep, wxs, wys, wxr, wyr = D.check_geometric_consistency(dn[0].copy(), K[0], E[0], dn[2], K[2], E[2])
    gm, gmask, gdep, gxs, gys, gxr, gyr = dtu_fusion.check_geometric_consistency(dn[0].copy(), K[0], E[0], dn[2], K[2], E[2], None)
    assert len(gm) == len(wm) == dtu_fusion.e - dtu_fusion.s and gmask.dtype == bool
    for k in range(len(wm)):
        assert (gm[k] == wm[k]).mean() >= 0.998, k
    same = gmask == wmask
    assert np.abs(gdep - wdep)[same].max() <= 5e-3 and np.abs(gxr - wxr)[same].max() <= 5e-3 and np.abs(gyr - wyr)[same].max() <= 5e-3
    assert (gdep[~gmask] == 0).all() and (gxr[~gmask] == 0).all()
    # CUDA tensors in -> CUDA tensors out, same values
    tt = dtu_fusion.reproject_with_depth(t(d[0], DEV), K[0], E[0], t(d[1], DEV), K[1], E[1])
    assert all(isinstance(a, torch.Tensor) and a.is_cuda for a in tt) and np.array_equal(tt[0].cpu().numpy(), got[0])


@pytest.mark.gpu
def test_dtu_filter_parity_unpinned_scan_directory_end_to_end(tmp_path):
    """filter_depth with the reference's arguments on a synthetic scan directory in the reference's formats (pair.txt, *_cam.txt,
    JPEG images, PFM depth / confidence maps): three mask PNGs per view and a binary PLY whose vertices are the masked pixels."""
    import numpy as np
    from PIL import Image
    from effi_mvs_plus_amd import dtu_fusion
    from effi_mvs_plus_amd.datasets.data_io import save_pfm
    H, W, N = 64, 96, 4
    d, cams, conf, K, E = _dtu_case(H, W, N, seed=2)
    scan, pairs = tmp_path / "out" / "scan1", tmp_path / "data" / "scan1"
    for sub in ("cams", "images", "depth_est", "confidence"):
        (scan / sub).mkdir(parents=True, exist_ok=True)
    pairs.mkdir(parents=True)
    rng = np.random.default_rng(1)
    with open(pairs / "pair.txt", "w") as f:
        f.write(f"{N}\n")
        for v in range(N):
            srcs = [u for u in range(N) if u != v]
            f.write(f"{v}\n{len(srcs)} " + " ".join(f"{u} {100.0 - u}" for u in srcs) + "\n")
    for v in range(N):
        with open(scan / "cams" / f"{v:08d}_cam.txt", "w") as f:
            f.write("extrinsic\n" + "\n".join(" ".join(repr(float(x)) for x in row) for row in E[v]) + "\n\nintrinsic\n")
            f.write("\n".join(" ".join(repr(float(x)) for x in row) for row in K[v]) + "\n\n425.0 2.5\n")
        Image.fromarray(rng.integers(0, 256, (H, W, 3), dtype=np.uint8)).save(scan / "images" / f"{v:08d}.jpg")
        save_pfm(str(scan / "depth_est" / f"{v:08d}.pfm"), d[v].numpy())
        save_pfm(str(scan / "confidence" / f"{v:08d}.pfm"), torch.rand(H // 2, W // 2, generator=torch.Generator().manual_seed(v)).numpy())
    ply = str(tmp_path / "out" / "mvsnet001_l3.ply")
    dtu_fusion.filter_depth(str(pairs), str(scan), str(scan), ply, conf=0.3, device=DEV)
    n_final = 0
    for v in range(N):
        for kind in ("photo", "geo", "final"):
            m = np.array(Image.open(scan / "mask" / f"{v:08d}_{kind}.png"))
            assert m.shape == (H, W) and set(np.unique(m)) <= {0, 255}
        n_final += int((np.array(Image.open(scan / "mask" / f"{v:08d}_final.png")) > 0).sum())
    raw = open(ply, "rb").read()
    head, body = raw.split(b"end_header\n", 1)
    assert head.startswith(b"ply\nformat binary_little_endian 1.0\n") and f"element vertex {n_final}\n".encode() in head
    assert len(body) == n_final * 15 and n_final > 0
    xyz = np.frombuffer(body, dtype=[("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("r", "u1"), ("g", "u1"), ("b", "u1")])
    assert np.isfinite(xyz["z"]).all()
