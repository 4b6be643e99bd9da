"""Tanks-and-Temples evaluation dataset (reference: datasets/tank.py:13-183): same constructor and sample dictionary as the
reference; every image goes to 1920 x 1056 (``read_img(filename, (1920, 1056))``, tank.py:134) on the device."""
import os

import numpy as np
import torch
from torch.utils.data import Dataset

from .. import ops
from .general_eval import decode_image, host_only, parse_pair_file, prepare_sample, stage_projections  # noqa: F401

STAGE_SCALES = {"stage0": 0.0625, "stage1": 0.125, "stage2": 0.25, "stage3": 0.5, "stage4": 1.0}       # tank.py:161-175
INTERMEDIATE = ["Family", "Francis", "Horse", "Lighthouse", "M60", "Panther", "Playground", "Train"]
ADVANCED = ["Auditorium", "Ballroom", "Courtroom", "Museum", "Palace", "Temple"]
WIDE_SCANS = ("Lighthouse", "M60", "Panther")          # 2048 x 1080 originals, all others 1920 x 1080 (tank.py:32-60)
NET_SIZE = (1920, 1056)                                # (w, h) every image is resized to (tank.py:134)


def parse_cam_file(path):
    """cam.txt -> (intrinsics [3,3] unscaled, extrinsics [4,4], depth_min, depth_max); the last line holds either
    "min max" / "min interval max"-style 2-3 fields (second = max) or 4 fields (fourth = max)  (tank.py:79-99)."""
    with open(path) as f:
        lines = [line.rstrip() for line in f.readlines()]
    extrinsics = np.array(" ".join(lines[1:5]).split(), dtype=np.float32).reshape(4, 4)
    intrinsics = np.array(" ".join(lines[7:10]).split(), dtype=np.float32).reshape(3, 3)
    fields = lines[11].split()
    return intrinsics, extrinsics, float(fields[0]), float(fields[1] if len(fields) < 4 else fields[3])


class MVSDataset(Dataset):
    def __init__(self, datapath, n_views=3, ndepths=192, img_wh=(1920, 1056), split="intermediate", scan=["Family"],
                 device="cuda"):
        self.stages = 4
        self.datapath = datapath
        self.img_wh = img_wh
        self.input_scans = scan
        self.split = split
        self.device = device
        self.build_metas()
        self.n_views = n_views
        self.ndepths = ndepths

    def build_metas(self):
        self.metas = []
        self.scans = self.input_scans
        names = (INTERMEDIATE + ADVANCED + ["Truck", "Ignatius"]) if self.split == "intermediate" else ADVANCED
        self.image_sizes = {n: ((2048, 1080) if n in WIDE_SCANS else (1920, 1080)) for n in names}
        for scan in self.scans:
            split = "intermediate" if scan in INTERMEDIATE else ("advanced" if scan in ADVANCED else "")
            for ref, srcs in parse_pair_file(os.path.join(self.datapath, split, scan, "pair.txt")):
                self.metas += [(scan, -1, ref, srcs, split)]

    def read_cam_file(self, filename):
        return parse_cam_file(filename)

    def center_img(self, img):
        """(img - mean) / (std + 1e-8) per channel of an [h,w,c] array (tank.py:110-113; unused by __getitem__ there too)."""
        var = np.var(img, axis=(0, 1), keepdims=True)
        mean = np.mean(img, axis=(0, 1), keepdims=True)
        return (img - mean) / (np.sqrt(var) + 0.00000001)

    def __len__(self):
        return len(self.metas)

    def __getitem__(self, idx):
        scan, _, ref_view, src_views, split = self.metas[idx]
        view_ids = [ref_view] + src_views[:self.n_views - 1]
        img_w, img_h = self.image_sizes[scan]
        cams = "cams_1" if split in ("intermediate", "advanced") else "cams"
        on_host = host_only(self.device)          # DataLoader worker / device="host": see general_eval.prepare_sample
        imgs = None if on_host else torch.empty(len(view_ids), 3, NET_SIZE[1], NET_SIZE[0], device=self.device, dtype=torch.float32)
        raws = []
        proj_matrices, depth_values = [], None
        for i, vid in enumerate(view_ids):
            raw = decode_image(os.path.join(self.datapath, split, scan, f"images/{vid:08d}.jpg"))
            if on_host:
                raws.append(torch.from_numpy(raw))
            else:
                ops.image_prepare(torch.from_numpy(raw).to(self.device, non_blocking=True), NET_SIZE[1], NET_SIZE[0], out=imgs[i])
            intrinsics, extrinsics, depth_min_, depth_max_ = self.read_cam_file(
                os.path.join(self.datapath, split, scan, f"{cams}/{vid:08d}_cam.txt"))
            intrinsics[0] *= self.img_wh[0] / img_w
            intrinsics[1] *= self.img_wh[1] / img_h
            proj_mat = np.zeros((2, 4, 4), dtype=np.float32)
            proj_mat[0, :4, :4] = extrinsics
            proj_mat[1, :3, :3] = intrinsics
            proj_matrices.append(proj_mat)
            if i == 0:
                depth_values = np.linspace(1 / depth_max_, 1 / depth_min_, self.ndepths, dtype=np.float32)
        sample = {"proj_matrices": stage_projections(np.stack(proj_matrices), STAGE_SCALES),
                  "depth_values": torch.from_numpy(depth_values.copy()).contiguous().float(),
                  "filename": scan + "/{}/" + "{:0>8}".format(view_ids[0]) + "{}"}
        if on_host:
            sample.update(imgs_u8=raws, imgs_hw=torch.tensor([[NET_SIZE[1], NET_SIZE[0]]] * len(view_ids), dtype=torch.int64),
                          std_hw=torch.tensor([NET_SIZE[1], NET_SIZE[0]], dtype=torch.int64))
        else:
            sample["imgs"] = imgs
        return sample
