"""CPU oracle for scope row n4 (input pipeline + on-disk formats).  TEST INFRASTRUCTURE ONLY -- see oracle/effi_oracle.py for
who may import this.  numpy restatement of the reference's datasets/ code, every function citing the lines it follows.

Parity pinning:
  * PFM reader / writer: PINNED -- tests/golden/g13_io.npz holds files written and arrays read by the reference's own
    datasets/data_io.py, run in the build container by tests/golden/make_golden_io.py.
  * cam.txt / pair.txt parsing, projection dictionary, inverse-depth samples: restated line by line; the reference's dataset
    modules import cv2 at the top (datasets/general_eval.py:3, datasets/tank.py:5), cv2 is not installed here, so they cannot
    be run: PARITY UNPINNED beyond the restatement.
  * image resize: the reference calls cv2.resize(..., INTER_LINEAR) on float32 images.  cv2 is absent from this image, so
    ``resize_linear`` restates OpenCV 4.x's published algorithm for that case (pixel centres at half integers, coordinate in
    double rounded to float, source index clamped, horizontal pass then vertical pass in fp32).  PARITY UNPINNED.
"""
import re
import sys

import numpy as np


# ---- datasets/data_io.py -------------------------------------------------------------------------
def read_pfm(filename):
    """datasets/data_io.py:61-95"""
    f = open(filename, "rb")
    header = f.readline().decode("utf-8").rstrip()
    if header not in ("PF", "Pf"):
        raise Exception("Not a PFM file.")
    m = re.match(r"^(\d+)\s(\d+)\s$", f.readline().decode("utf-8"))
    if not m:
        raise Exception("Malformed PFM header.")
    width, height = int(m.group(1)), int(m.group(2))
    scale = float(f.readline().rstrip())
    data = np.fromfile(f, ("<" if scale < 0 else ">") + "f")
    f.close()
    data = data.reshape((height, width, 3) if header == "PF" else (height, width))
    return np.flipud(data), abs(scale)


def pfm_bytes(image, scale=1):
    """The bytes datasets/data_io.py:98-126 (save_pfm) writes for a float32 image."""
    image = np.flipud(image)
    color = image.ndim == 3 and image.shape[2] == 3
    if image.dtype.byteorder == "<" or (image.dtype.byteorder == "=" and sys.byteorder == "little"):
        scale = -scale
    head = ("PF\n" if color else "Pf\n") + "{} {}\n".format(image.shape[1], image.shape[0]) + ("%f\n" % scale)
    return head.encode("utf-8") + image.tobytes()


# ---- cv2.resize(img, (w, h)), INTER_LINEAR, float32 (published algorithm; see the header) ------------
def _taps(dst_n, src_n):
    scale = 1.0 / (float(dst_n) / float(src_n))
    f = ((np.arange(dst_n, dtype=np.float64) + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = f - s.astype(np.float32)
    lo = s < 0
    s[lo], f[lo] = 0, 0.0
    hi = s >= src_n - 1
    s[hi], f[hi] = src_n - 1, 0.0
    return s, np.minimum(s + 1, src_n - 1), (np.float32(1.0) - f).astype(np.float32), f.astype(np.float32)


def resize_linear(img, dst_w, dst_h):
    """img float32 [h,w] or [h,w,c] -> [dst_h,dst_w(,c)]"""
    img = np.asarray(img, dtype=np.float32)
    x0, x1, a0, a1 = _taps(dst_w, img.shape[1])
    y0, y1, b0, b1 = _taps(dst_h, img.shape[0])
    ex = (lambda v: v[None, :, None]) if img.ndim == 3 else (lambda v: v[None, :])
    ey = (lambda v: v[:, None, None]) if img.ndim == 3 else (lambda v: v[:, None])
    rows = img[:, x0] * ex(a0) + img[:, x1] * ex(a1)          # horizontal pass, fp32
    return (rows[y0] * ey(b0) + rows[y1] * ey(b1)).astype(np.float32)


# ---- datasets/general_eval.py ----------------------------------------------------------------------
def read_pairs(path, nviews=None):
    """datasets/general_eval.py:39-52"""
    metas = []
    with open(path) as f:
        num_viewpoint = int(f.readline())
        for _ in range(num_viewpoint):
            ref_view = int(f.readline().rstrip())
            src_views = [int(x) for x in f.readline().rstrip().split()[1::2]]
            if len(src_views) > 0:
                if nviews is not None and len(src_views) < nviews:
                    src_views += [src_views[0]] * (nviews - len(src_views))
                metas.append((ref_view, src_views))
    return metas


def _matrix(lines, shape):
    # np.fromstring(' '.join(lines), dtype=np.float32, sep=' ') in the reference: decimal text -> double -> float32
    return np.array([float(t) for t in " ".join(lines).split()], dtype=np.float64).astype(np.float32).reshape(shape)


def read_cam_dtu(path, ndepths, interval_scale):
    """datasets/general_eval.py:60-81"""
    lines = [line.rstrip() for line in open(path).readlines()]
    extrinsics = _matrix(lines[1:5], (4, 4))
    intrinsics = _matrix(lines[7:10], (3, 3))
    intrinsics[:2, :] /= 4.0
    depth_min = float(lines[11].split()[0])
    depth_interval = 2.5
    if len(lines[11].split()) >= 3:
        num_depth = lines[11].split()[2]
        depth_max = depth_min + int(float(num_depth)) * depth_interval
        depth_interval = (depth_max - depth_min) / ndepths
    depth_interval *= interval_scale
    return intrinsics, extrinsics, depth_min, depth_interval


def scale_mvs_input(img, intrinsics, max_w, max_h, base=32):
    """datasets/general_eval.py:94-117"""
    h, w = img.shape[:2]
    scale = 1.0 * max_w / w
    new_w = (scale * w) // base * base
    scale = 1.0 * max_h / h
    new_h = scale * h // base * base
    intrinsics[0, :] *= 1.0 * new_w / w
    intrinsics[1, :] *= 1.0 * new_h / h
    return resize_linear(img, int(new_w), int(new_h)), intrinsics


def _stage_dict(proj_matrices, scales):
    out = {}
    for k, s in scales.items():
        m = proj_matrices.copy()
        m[:, 1, :2, :] = proj_matrices[:, 1, :2, :] * s
        out[k] = m
    return out


def general_eval_sample(images_u8, cam_paths, ndepths, interval_scale, max_h, max_w, dispmaxfirst="first"):
    """datasets/general_eval.py:120-228 for one sample whose decoded images (uint8 [h,w,3]) and cam files are given
    (per-sample standard size: fix_res=False)."""
    imgs, mats, depth_values, std = [], [], None, None
    for i, (raw, cam) in enumerate(zip(images_u8, cam_paths)):
        img = np.array(raw, dtype=np.float32) / 255.                                      # :83-88
        intrinsics, extrinsics, depth_min, depth_interval = read_cam_dtu(cam, ndepths, interval_scale)
        img, intrinsics = scale_mvs_input(img, intrinsics, max_w, max_h)
        if i == 0:
            std = img.shape[:2]
        if img.shape[:2] != std:                                                          # :160-166
            intrinsics[0, :] *= 1.0 * std[1] / img.shape[1]
            intrinsics[1, :] *= 1.0 * std[0] / img.shape[0]
            img = resize_linear(img, std[1], std[0])
        imgs.append(img)
        m = np.zeros((2, 4, 4), dtype=np.float32)
        m[0, :4, :4] = extrinsics
        m[1, :3, :3] = intrinsics
        mats.append(m)
        if i == 0:                                                                        # :178-185
            depth_max = depth_interval * ndepths + depth_min
            disp_min, disp_max = 1 / depth_max, 1 / depth_min
            depth_values = (np.linspace(disp_max, disp_min, ndepths, dtype=np.float32) if dispmaxfirst == "first"
                            else np.linspace(disp_min, disp_max, ndepths, dtype=np.float32))
    scales = {"stage0": 0.25, "stage1": 0.5, "stage2": 1, "stage3": 2, "stage4": 4}       # :199-210
    return {"imgs": np.stack(imgs).transpose([0, 3, 1, 2]), "proj_matrices": _stage_dict(np.stack(mats), scales),
            "depth_values": depth_values}


# ---- datasets/tank.py --------------------------------------------------------------------------------
def read_cam_tank(path):
    """datasets/tank.py:79-99"""
    lines = [line.rstrip() for line in open(path).readlines()]
    extrinsics = _matrix(lines[1:5], (4, 4))
    intrinsics = _matrix(lines[7:10], (3, 3))
    v = lines[11].split()
    return intrinsics, extrinsics, float(v[0]), float(v[1]) if len(v) < 4 else float(v[3])


def tank_sample(images_u8, cam_paths, ndepths, orig_wh, img_wh=(1920, 1056)):
    """datasets/tank.py:118-183 for one sample"""
    imgs, mats, depth_values = [], [], None
    for i, (raw, cam) in enumerate(zip(images_u8, cam_paths)):
        img = resize_linear(np.array(raw, dtype=np.float32) / 255., 1920, 1056)           # :101-107,134
        intrinsics, extrinsics, dmin, dmax = read_cam_tank(cam)
        intrinsics[0] *= img_wh[0] / orig_wh[0]
        intrinsics[1] *= img_wh[1] / orig_wh[1]
        imgs.append(img)
        m = np.zeros((2, 4, 4), dtype=np.float32)
        m[0, :4, :4] = extrinsics
        m[1, :3, :3] = intrinsics
        mats.append(m)
        if i == 0:
            depth_values = np.linspace(1 / dmax, 1 / dmin, ndepths, dtype=np.float32)
    scales = {"stage0": 0.0625, "stage1": 0.125, "stage2": 0.25, "stage3": 0.5, "stage4": 1}
    return {"imgs": np.stack(imgs).transpose([0, 3, 1, 2]), "proj_matrices": _stage_dict(np.stack(mats), scales),
            "depth_values": depth_values}
