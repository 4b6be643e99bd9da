"""DTU-style evaluation dataset (reference: datasets/general_eval.py:8-228): same constructor, same sample dictionary
{"imgs" [N,3,H,W], "proj_matrices" {"stage0".."stage4": [N,2,4,4]}, "depth_values" [ndepths], "filename"}.

Host side: pair.txt / cam.txt parsing, intrinsics scaling, the per-stage projection dictionary, the inverse-depth samples.
Device side: one kernel per view for /255 + bilinear resize + HWC -> CHW (``ops.image_prepare``); the sample's tensors are
returned on ``device`` (default "cuda"), ready for the model.  Without the GPU / HIP library ``__getitem__`` raises.
"""
import os

import numpy as np
import torch
from torch.utils.data import Dataset

from .. import ops
from .data_io import read_pfm

STAGE_SCALES = {"stage0": 0.25, "stage1": 0.5, "stage2": 1.0, "stage3": 2.0, "stage4": 4.0}   # general_eval.py:201-210


def parse_pair_file(path, nviews=None, verbose=True):
    """pair.txt -> [(ref_view, [src views by score])]; views without sources are dropped, short lists are padded with their
    first entry up to ``nviews`` (general_eval.py:39-52)."""
    out = []
    with open(path) as f:
        n = int(f.readline())
        for _ in range(n):
            ref = int(f.readline().rstrip())
            srcs = [int(x) for x in f.readline().rstrip().split()[1::2]]
            if not srcs:
                continue
            if nviews is not None and len(srcs) < nviews:
                if verbose:
                    print("{}< num_views:{}".format(len(srcs), nviews))
                srcs = srcs + [srcs[0]] * (nviews - len(srcs))
            out.append((ref, srcs))
    return out


def parse_cam_file(path, ndepths, interval_scale):
    """cam.txt -> (intrinsics [3,3] with rows 0-1 / 4, extrinsics [4,4], depth_min, depth_interval)  (general_eval.py:60-81)."""
    with open(path) as f:
        lines = [line.rstrip() for line in f.readlines()]
    extrinsics = np.array(" ".join(lines[1:5]).split(), dtype=np.float32).reshape(4, 4)
    intrinsics = np.array(" ".join(lines[7:10]).split(), dtype=np.float32).reshape(3, 3)
    intrinsics[:2, :] /= 4.0
    fields = lines[11].split()
    depth_min = float(fields[0])
    depth_interval = 2.5                                  # the file's own interval is ignored (general_eval.py:72)
    if len(fields) >= 3:
        depth_max = depth_min + int(float(fields[2])) * depth_interval
        depth_interval = (depth_max - depth_min) / ndepths
    return intrinsics, extrinsics, depth_min, depth_interval * interval_scale


def scaled_size(h, w, max_h, max_w, base=32):
    """Target size of scale_mvs_input: each side scaled to its maximum, rounded down to a multiple of ``base``
    (general_eval.py:104-109; the float floor-division is kept)."""
    new_w = (1.0 * max_w / w * w) // base * base
    new_h = 1.0 * max_h / h * h // base * base
    return int(new_h), int(new_w), 1.0 * new_h / h, 1.0 * new_w / w


def stage_projections(proj_matrices, scales=STAGE_SCALES):
    """[N,2,4,4] -> {"stageK": same with intrinsic rows 0-1 x scale}  (general_eval.py:199-217)."""
    out = {}
    for name, s in scales.items():
        m = proj_matrices.copy()
        m[:, 1, :2, :] = proj_matrices[:, 1, :2, :] * s
        out[name] = torch.from_numpy(m).contiguous().float()
    return out


def inverse_depth_samples(depth_min, depth_interval, ndepths, dispmaxfirst):
    """general_eval.py:178-185"""
    depth_max = depth_interval * ndepths + depth_min
    disp_min, disp_max = 1 / depth_max, 1 / depth_min
    if dispmaxfirst == "first":
        return np.linspace(disp_max, disp_min, ndepths, dtype=np.float32)
    return np.linspace(disp_min, disp_max, ndepths, dtype=np.float32)


def decode_image(filename):
    """JPG / PNG -> uint8 [h,w,3] (the reference converts to float on the host, general_eval.py:83-88; here the bytes go to
    the device as they are)."""
    from PIL import Image
    with Image.open(filename) as img:
        arr = np.array(img)
    if arr.dtype != np.uint8:
        raise TypeError(f"{filename}: 8-bit image expected, got {arr.dtype}")
    return arr


class MVSDataset(Dataset):
    def __init__(self, datapath, listfile, mode, nviews, ndepths=192, interval_scale=1.06, dispmaxfirst="first", **kwargs):
        super().__init__()
        self.datapath = datapath
        self.listfile = listfile
        self.mode = mode
        self.nviews = nviews
        self.ndepths = ndepths
        self.interval_scale = interval_scale
        self.dispmaxfirst = dispmaxfirst
        self.max_h, self.max_w = kwargs["max_h"], kwargs["max_w"]
        self.fix_res = kwargs.get("fix_res", False)     # one standard size for the whole scene instead of one per sample
        self.fix_wh = False
        self.device = kwargs.get("device", "cuda")
        self._std = (0, 0)                              # the reference keeps this in module globals s_h, s_w
        assert self.mode == "test"
        self.metas = self.build_list()

    def build_list(self):
        metas, scales = [], {}
        for scan in self.listfile:
            scales[scan] = self.interval_scale if isinstance(self.interval_scale, float) else self.interval_scale[scan]
            for ref, srcs in parse_pair_file(os.path.join(self.datapath, "{}/pair.txt".format(scan)), self.nviews):
                metas.append((scan, ref, srcs, scan))
        self.interval_scale = scales
        print("dataset", self.mode, "metas:", len(metas), "interval_scale:{}".format(self.interval_scale))
        return metas

    def __len__(self):
        return len(self.metas)

    def read_cam_file(self, filename, interval_scale):
        return parse_cam_file(filename, self.ndepths, interval_scale)

    def read_depth(self, filename):
        return np.array(read_pfm(filename)[0], dtype=np.float32)

    def view_paths(self, scan, vid):
        img = os.path.join(self.datapath, "{}/images_post/{:0>8}.jpg".format(scan, vid))
        if not os.path.exists(img):
            img = os.path.join(self.datapath, "{}/images/{:0>8}.jpg".format(scan, vid))
        return img, os.path.join(self.datapath, "{}/cams/{:0>8}_cam.txt".format(scan, vid))

    def __getitem__(self, idx):
        scan, ref_view, src_views, scene_name = self.metas[idx]
        view_ids = [ref_view] + src_views[:self.nviews - 1]
        imgs, proj_matrices, depth_values = None, [], None
        for i, vid in enumerate(view_ids):
            img_filename, cam_filename = self.view_paths(scan, vid)
            raw = decode_image(img_filename)
            intrinsics, extrinsics, depth_min, depth_interval = self.read_cam_file(cam_filename, self.interval_scale[scene_name])
            h1, w1, sy, sx = scaled_size(raw.shape[0], raw.shape[1], self.max_h, self.max_w)
            intrinsics[0, :] *= sx
            intrinsics[1, :] *= sy
            if self.fix_res:                      # the first view ever read fixes the size of the whole scene
                self._std = (h1, w1)
                self.fix_res, self.fix_wh = False, True
            if i == 0 and not self.fix_wh:
                self._std = (h1, w1)
            s_h, s_w = self._std
            dev_raw = torch.from_numpy(raw).to(self.device, non_blocking=True)
            if imgs is None:
                imgs = torch.empty(len(view_ids), 3, s_h, s_w, device=self.device, dtype=torch.float32)
            if (h1, w1) == (s_h, s_w):
                ops.image_prepare(dev_raw, s_h, s_w, out=imgs[i])
            else:
                # the reference resizes twice (to the view's own size, then to the standard one): same two passes here
                first = ops.image_prepare(dev_raw, h1, w1)
                imgs[i] = ops.resize_planar(first, s_h, s_w)
                intrinsics[0, :] *= 1.0 * s_w / w1
                intrinsics[1, :] *= 1.0 * s_h / h1
            proj_mat = np.zeros((2, 4, 4), dtype=np.float32)
            proj_mat[0, :4, :4] = extrinsics
            proj_mat[1, :3, :3] = intrinsics
            proj_matrices.append(proj_mat)
            if i == 0:
                depth_values = inverse_depth_samples(depth_min, depth_interval, self.ndepths, self.dispmaxfirst)
        return {"imgs": imgs,
                "proj_matrices": stage_projections(np.stack(proj_matrices)),
                "depth_values": torch.from_numpy(depth_values.copy()).contiguous().float(),
                "filename": scan + "/{}/" + "{:0>8}".format(view_ids[0]) + "{}"}
