"""Scope row n4: input pipeline + on-disk formats (datasets/data_io.py, general_eval.py, tank.py).

CPU: PFM reader / writer (product and oracle) against files written / read by the reference itself (g13_io.npz); cam / pair
parsing and the projection bookkeeping of the product against the oracle's line-by-line restatement; loud failure without the GPU.
GPU: the image kernels through the C ABI against the oracle's resize, and whole samples of both dataset classes built from a
synthetic scan directory against the oracle's pipeline.
"""
import os

import numpy as np
import pytest
import torch

from common import GOLDEN
from effi_mvs_plus_amd.datasets import data_io, general_eval, tank
from oracle import effi_io_oracle as IO

DEV = "cuda:0"
PFM_CASES = ["gray_s1", "gray_s2p5", "gray1_s1", "gray1_s2p5", "color_s1", "color_s2p5", "depth_s1", "depth_s2p5"]


@pytest.fixture(scope="module")
def g13():
    return dict(np.load(os.path.join(GOLDEN, "g13_io.npz")))


# ---- PFM ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("tag", PFM_CASES)
def test_save_pfm_writes_the_reference_bytes(g13, tag, tmp_path):
    img, scale = g13[f"{tag}_image"], float(tag.split("_s")[1].replace("p", "."))
    p = tmp_path / "x.pfm"
    data_io.save_pfm(str(p), img, scale if scale != 1 else 1)
    assert p.read_bytes() == g13[f"{tag}_file"].tobytes()
    assert IO.pfm_bytes(img, scale if scale != 1 else 1) == g13[f"{tag}_file"].tobytes()


@pytest.mark.parametrize("tag", PFM_CASES + ["be"])
def test_read_pfm_returns_the_reference_arrays(g13, tag, tmp_path):
    p = tmp_path / "x.pfm"
    p.write_bytes(g13[f"{tag}_file"].tobytes())
    for reader in (data_io.read_pfm, data_io.pfm_imread, IO.read_pfm):
        data, scale = reader(str(p))
        assert scale == float(g13[f"{tag}_scale"])
        assert data.dtype.kind == "f" and np.array_equal(np.asarray(data, dtype=np.float32), g13[f"{tag}_read"])


def test_pfm_errors_like_the_reference(tmp_path):
    p = tmp_path / "bad.pfm"
    p.write_bytes(b"P6\n3 4\n1.0\n")
    with pytest.raises(Exception, match="Not a PFM file"):
        data_io.read_pfm(str(p))
    p.write_bytes(b"Pf\n3x4\n1.0\n")
    with pytest.raises(Exception, match="Malformed PFM header"):
        data_io.read_pfm(str(p))
    with pytest.raises(Exception, match="float32"):
        data_io.save_pfm(str(p), np.zeros((2, 2), dtype=np.float64))
    with pytest.raises(Exception, match="dimensions"):
        data_io.save_pfm(str(p), np.zeros((2, 2, 2), dtype=np.float32))
    assert data_io.read_all_lines(__file__)[0].startswith('"""Scope row n4')


# ---- a synthetic scan directory ----------------------------------------------------------------
def _cam_text(rng, last_line):
    e = np.eye(4)
    e[:3, :3] = np.linalg.qr(rng.standard_normal((3, 3)))[0]
    e[:3, 3] = rng.standard_normal(3) * 300
    k = np.array([[2892.33, 0, 823.205], [0, 2883.175, 619.071], [0, 0, 1]]) + rng.standard_normal((3, 3)) * 1e-3
    rows = lambda m: "\n".join(" ".join(repr(float(v)) for v in r) for r in m)  # noqa: E731
    return f"extrinsic\n{rows(e)}\n\nintrinsic\n{rows(k)}\n\n{last_line}\n"


def _write_scan(root, scan, n_img, sizes, last_line, cams_dir="cams", rng=None):
    from PIL import Image
    rng = rng or np.random.default_rng(0)
    os.makedirs(os.path.join(root, scan, "images"), exist_ok=True)
    os.makedirs(os.path.join(root, scan, cams_dir), exist_ok=True)
    raws = []
    for v in range(n_img):
        h, w = sizes[v % len(sizes)]
        # smooth content (JPEG-friendly) + decode what was stored, so that test and product see the same bytes
        yy, xx = np.mgrid[0:h, 0:w]
        img = np.stack([127 + 120 * np.sin(xx / (7.0 + v) + c) * np.cos(yy / (11.0 + c)) for c in range(3)], -1).astype(np.uint8)
        p = os.path.join(root, scan, "images", f"{v:08d}.jpg")
        Image.fromarray(img).save(p, quality=95)
        raws.append(np.array(Image.open(p)))
        with open(os.path.join(root, scan, cams_dir, f"{v:08d}_cam.txt"), "w") as f:
            f.write(_cam_text(rng, last_line))
    with open(os.path.join(root, scan, "pair.txt"), "w") as f:
        f.write(f"{n_img}\n")
        for v in range(n_img):
            others = [u for u in range(n_img) if u != v][: (0 if v == n_img - 1 else 2 + v % 2)]
            f.write(f"{v}\n{len(others)} " + " ".join(f"{u} {100.0 - u}" for u in others) + "\n")
    return raws


def test_pair_and_cam_parsing_match_the_oracle(tmp_path, capsys):
    root = str(tmp_path)
    _write_scan(root, "scan1", 5, [(60, 80)], "425.0 2.5 192.0 935.0")
    pairs = general_eval.parse_pair_file(os.path.join(root, "scan1", "pair.txt"), nviews=4)
    assert pairs == IO.read_pairs(os.path.join(root, "scan1", "pair.txt"), nviews=4)
    assert len(pairs) == 4 and all(len(s) >= 4 for _, s in pairs)         # last view has no sources; short lists are padded
    cam = os.path.join(root, "scan1", "cams", "00000002_cam.txt")
    for ndepths, isc in ((192, 1.06), (384, 0.8)):
        got, want = general_eval.parse_cam_file(cam, ndepths, isc), IO.read_cam_dtu(cam, ndepths, isc)
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1]) and got[2:] == want[2:]
    got, want = tank.parse_cam_file(cam), IO.read_cam_tank(cam)
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1]) and got[2:] == want[2:] and got[3] == 935.0
    short = os.path.join(root, "short_cam.txt")
    open(short, "w").write(_cam_text(np.random.default_rng(1), "0.5 2.25"))
    assert tank.parse_cam_file(short)[2:] == (0.5, 2.25) == IO.read_cam_tank(short)[2:]
    assert general_eval.parse_cam_file(short, 192, 1.0)[3] == 2.5       # fewer than 3 fields: fixed 2.5 interval (general_eval.py:72)
    # np.fromstring(sep=' '), which the reference uses, gives the same float32 values as both parsers
    lines = [ln.rstrip() for ln in open(cam).readlines()]
    ref_e = np.fromstring(" ".join(lines[1:5]), dtype=np.float32, sep=" ").reshape(4, 4)
    assert np.array_equal(ref_e, got[1])
    assert general_eval.scaled_size(1200, 1600, 1184, 1600) == (1184, 1600, 1184 / 1200, 1.0)
    assert general_eval.scaled_size(1080, 1920, 1056, 1920)[:2] == (1056, 1920)
    d1 = general_eval.inverse_depth_samples(425.0, 2.65, 192, "first")
    d2 = general_eval.inverse_depth_samples(425.0, 2.65, 192, "last")
    assert d1.dtype == np.float32 and np.array_equal(d2, np.linspace(1 / (2.65 * 192 + 425.0), 1 / 425.0, 192, dtype=np.float32))
    assert d1[0] > d1[-1] and d2[0] < d2[-1] and d1[0] == np.float32(1 / 425.0)


def test_dataset_lists_and_fails_loudly_without_gpu(tmp_path, capsys):
    from effi_mvs_plus_amd._lib import EffiLibraryError
    root = str(tmp_path)
    _write_scan(root, "scan9", 4, [(64, 96)], "425.0 2.5 192.0 935.0")
    ds = general_eval.MVSDataset(root, ["scan9"], "test", 3, 192, 1.06, max_h=64, max_w=96, device="cpu")
    assert len(ds) == 3 and ds.metas[0] == ("scan9", 0, [1, 2, 1], "scan9") and ds.interval_scale == {"scan9": 1.06}
    assert "metas: 3" in capsys.readouterr().out
    with pytest.raises(AssertionError):
        general_eval.MVSDataset(root, ["scan9"], "train", 3, max_h=64, max_w=96)
    with pytest.raises(EffiLibraryError):
        ds[0]                                   # the image arithmetic is a HIP kernel: no CPU fallback
    m = np.arange(2 * 2 * 16, dtype=np.float32).reshape(2, 2, 4, 4)
    st = general_eval.stage_projections(m)
    assert list(st) == ["stage0", "stage1", "stage2", "stage3", "stage4"]
    assert torch.equal(st["stage3"][:, 1, :2], torch.from_numpy(m[:, 1, :2] * 2)) and torch.equal(st["stage3"][:, 0], torch.from_numpy(m[:, 0]))
    assert torch.equal(st["stage1"][:, 1, 2:], torch.from_numpy(m[:, 1, 2:]))


def test_dataloader_workers_stay_on_the_host(tmp_path, capsys):
    """The reference's drivers wrap the dataset in DataLoader(num_workers=4/8) (test_dtu_dypcd.py:406, test_tank.py:209): forked
    workers must not touch the GPU, so inside a worker __getitem__ returns the decoded bytes and prepare_sample() finishes the
    sample in the main process."""
    from torch.utils.data import DataLoader
    root = str(tmp_path)
    raws = _write_scan(root, "scan4", 4, [(64, 96), (48, 64)], "425.0 2.5 192.0 935.0")
    ds = general_eval.MVSDataset(root, ["scan4"], "test", 3, 96, 1.06, max_h=64, max_w=96)       # device="cuda" (the default)
    for batch in DataLoader(ds, batch_size=1, shuffle=False, num_workers=2, drop_last=False):
        assert "imgs" not in batch and len(batch["imgs_u8"]) == 3
        assert batch["imgs_u8"][0].dtype == torch.uint8 and batch["imgs_u8"][0].dim() == 4 and batch["imgs_u8"][0].shape[0] == 1
        assert tuple(batch["imgs_hw"].shape) == (1, 3, 2) and batch["std_hw"].tolist() == [[64, 96]]
        assert tuple(batch["proj_matrices"]["stage2"].shape) == (1, 3, 2, 4, 4) and tuple(batch["depth_values"].shape) == (1, 96)
    ref = ds.metas[0][1]
    assert torch.equal(general_eval.MVSDataset(root, ["scan4"], "test", 3, 96, 1.06, max_h=64, max_w=96, device="host")[0]["imgs_u8"][0],
                       torch.from_numpy(raws[ref]))
    done = {"imgs": torch.zeros(1)}
    assert general_eval.prepare_sample(done) is done                    # already complete: untouched


# ---- GPU -----------------------------------------------------------------------------------------
@pytest.mark.gpu
def test_prepare_sample_of_worker_batches_equals_the_main_process_sample(tmp_path, capsys):
    from torch.utils.data import DataLoader
    root = str(tmp_path)
    _write_scan(root, "scan4", 4, [(64, 96), (48, 64)], "425.0 2.5 192.0 935.0")
    mk = lambda **kw: general_eval.MVSDataset(root, ["scan4"], "test", 3, 96, 1.06, max_h=64, max_w=96, **kw)  # noqa: E731
    direct = mk(device=DEV)
    for idx, batch in enumerate(DataLoader(mk(device=DEV), batch_size=1, shuffle=False, num_workers=2)):
        got = general_eval.prepare_sample(batch, device=DEV)
        want = direct[idx]
        assert got["imgs"].is_cuda and tuple(got["imgs"].shape) == (1,) + tuple(want["imgs"].shape)
        assert torch.equal(got["imgs"][0], want["imgs"])
        for k in want["proj_matrices"]:
            assert torch.equal(got["proj_matrices"][k][0], want["proj_matrices"][k])
        assert torch.equal(got["depth_values"][0], want["depth_values"])
    _write_scan(os.path.join(root, "intermediate"), "Family", 3, [(135, 240)], "0.5 0.01 192 2.25", cams_dir="cams_1")
    td = tank.MVSDataset(root, n_views=3, ndepths=64, split="intermediate", scan=["Family"], device=DEV)
    batch = next(iter(DataLoader(td, batch_size=1, num_workers=1)))
    assert torch.equal(tank.prepare_sample(batch, device=DEV)["imgs"][0], td[0]["imgs"])


@pytest.mark.gpu
@pytest.mark.parametrize("src,dst", [((1200, 1600), (1184, 1600)), ((1080, 1920), (1056, 1920)), ((1080, 2048), (1056, 1920)),
                                     ((37, 53), (64, 96)), ((64, 96), (64, 96)), ((50, 70), (17, 23))])
def test_image_prepare_matches_oracle_resize(src, dst):
    from effi_mvs_plus_amd import ops
    rng = np.random.default_rng(src[0] + dst[1])
    raw = rng.integers(0, 256, size=(src[0], src[1], 3), dtype=np.uint8)
    want = IO.resize_linear(raw.astype(np.float32) / 255., dst[1], dst[0]).transpose(2, 0, 1)
    got = ops.image_prepare(torch.from_numpy(raw).to(DEV), dst[0], dst[1]).cpu().numpy()
    assert got.shape == want.shape
    assert np.abs(got - want).max() <= 2e-7                     # same fp32 formula; fused multiply-adds are off in both
    if src == dst:
        assert np.array_equal(got, (raw.astype(np.float32) / 255.).transpose(2, 0, 1))      # identity resize is exact
    back = ops.resize_planar(torch.from_numpy(np.ascontiguousarray(want)).to(DEV), src[0], src[1]).cpu().numpy()
    want_back = IO.resize_linear(np.ascontiguousarray(want.transpose(1, 2, 0)), src[1], src[0]).transpose(2, 0, 1)
    assert np.abs(back - want_back).max() <= 2e-7
    gray = ops.image_prepare(torch.from_numpy(np.ascontiguousarray(raw[..., 0])).to(DEV), dst[0], dst[1]).cpu().numpy()
    assert np.array_equal(gray[0], got[0])


@pytest.mark.gpu
@pytest.mark.parametrize("sizes", [[(120, 160)], [(120, 160), (96, 128)]])       # second case: views of different sizes
def test_general_eval_sample_matches_oracle(tmp_path, sizes, capsys):
    root = str(tmp_path)
    raws = _write_scan(root, "scan4", 5, sizes, "425.0 2.5 192.0 935.0")
    ds = general_eval.MVSDataset(root, ["scan4"], "test", 4, 96, 1.06, dispmaxfirst="last", max_h=100, max_w=160, device=DEV)
    for idx in (0, 1):
        s = ds[idx]
        _, ref, srcs, _ = ds.metas[idx]
        ids = [ref] + srcs[:3]
        want = IO.general_eval_sample([raws[v] for v in ids], [os.path.join(root, "scan4", "cams", f"{v:08d}_cam.txt") for v in ids],
                                      96, 1.06, 100, 160, "last")
        assert s["imgs"].is_cuda and tuple(s["imgs"].shape) == want["imgs"].shape and s["imgs"].shape[-2:] == (96, 160)
        assert np.abs(s["imgs"].cpu().numpy() - want["imgs"]).max() <= 4e-7
        for k in ("stage0", "stage1", "stage2", "stage3", "stage4"):
            assert np.array_equal(s["proj_matrices"][k].numpy(), want["proj_matrices"][k])
        assert np.array_equal(s["depth_values"].numpy(), want["depth_values"])
        assert s["filename"] == "scan4/{}/" + f"{ref:08d}" + "{}"


@pytest.mark.gpu
def test_tank_sample_matches_oracle(tmp_path):
    root = str(tmp_path)
    raws = _write_scan(os.path.join(root, "intermediate"), "Family", 3, [(135, 240)], "0.5 0.01 192 2.25", cams_dir="cams_1")
    ds = tank.MVSDataset(root, n_views=3, ndepths=64, split="intermediate", scan=["Family"], device=DEV)
    assert len(ds) == 2
    s = ds[0]
    cams = [os.path.join(root, "intermediate", "Family", "cams_1", f"{v:08d}_cam.txt") for v in (0, 1, 2)]
    want = IO.tank_sample([raws[0], raws[1], raws[2]], cams, 64, (1920, 1080))
    assert tuple(s["imgs"].shape) == (3, 3, 1056, 1920)
    assert np.abs(s["imgs"].cpu().numpy() - want["imgs"]).max() <= 4e-7
    for k in want["proj_matrices"]:
        assert np.array_equal(s["proj_matrices"][k].numpy(), want["proj_matrices"][k])
    assert np.array_equal(s["depth_values"].numpy(), want["depth_values"]) and s["depth_values"][0] == np.float32(1 / 2.25)
