"""The DTU driver's depth-map filter + point-cloud export on the HIP path (DTU branch): the function
names of the reference's ``test_dtu_dypcd.py:123-350`` (``read_camera_parameters``, ``read_pair_file``, ``reproject_with_depth``'s and
``check_geometric_consistency``'s work inside ONE fused kernel per reference view, ``filter_depth``, ``dypcd_filter``).

PARITY UNPINNED (the bar of this row stays "partial"): the reference samples with ``cv2.remap`` and resizes with ``cv2.resize``; cv2 is
not installed in the build image and the reference holds no fixtures for this code.  The kernel (csrc/fusion.hip,
``effi_fusion_dtu_filter_f32``) follows the numpy lines dtype for dtype and OpenCV's published INTER_LINEAR arithmetic; it is checked
against ``oracle/effi_dtu_filter_oracle.py`` (tests/test_fusion.py::test_dtu_filter_parity_unpinned_*).

The reference runs this filter on the host, one scan per ``multiprocessing.Pool`` worker, re-reading every depth map ~11 times from
disk; here a scan's depth maps, confidences and cameras are uploaded once and every reference view is one kernel launch.
File formats (PFM depth / confidence maps, ``*_cam.txt``, ``pair.txt``, mask PNGs, binary PLY with x y z red green blue) are the
reference's; the PLY is written without ``plyfile`` (absent here) in the layout ``PlyData([el]).write`` produces.
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

from . import ops
from .datasets.data_io import read_pfm

# module constants of the reference (test_dtu_dypcd.py:33-37)
s, e = 1, 11
dist_base, diff_base = 1 / 2, 0.25


def read_camera_parameters(filename):
    """test_dtu_dypcd.py:123-131 -> (intrinsics [3,3], extrinsics [4,4]) float32."""
    with open(filename) as f:
        lines = [line.rstrip() for line in f.readlines()]
    extrinsics = np.array(" ".join(lines[1:5]).split(), dtype=np.float32).reshape((4, 4))
    intrinsics = np.array(" ".join(lines[7:10]).split(), dtype=np.float32).reshape((3, 3))
    return intrinsics, extrinsics


def read_pair_file(filename):
    """test_dtu_dypcd.py:150-160 -> [(ref_view, [src_views...]), ...] (reference views without sources are skipped)."""
    data = []
    with open(filename) as f:
        num_viewpoint = int(f.readline())
        for _ in range(num_viewpoint):
            ref_view = int(f.readline().rstrip())
            src_views = [int(x) for x in f.readline().rstrip().split()[1::2]]
            if len(src_views) > 0:
                data.append((ref_view, src_views))
    return data


def read_img(filename):
    """test_dtu_dypcd.py:134-138: image as float32 in [0, 1]."""
    from PIL import Image
    return np.array(Image.open(filename), dtype=np.float32) / 255.


def save_mask(filename, mask):
    """test_dtu_dypcd.py:144-148."""
    from PIL import Image
    assert mask.dtype == bool
    Image.fromarray(mask.astype(np.uint8) * 255).save(filename)


def _cam_tensor(intrinsics, extrinsics, device):
    cam = torch.zeros(2, 4, 4, dtype=torch.float32)
    cam[0] = torch.from_numpy(np.ascontiguousarray(extrinsics, dtype=np.float32))
    cam[1, :3, :3] = torch.from_numpy(np.ascontiguousarray(intrinsics, dtype=np.float32))
    return cam.to(device)


def _dev_map(a, device):
    return a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(device)


def _like_input(t_, like):
    return t_ if isinstance(like, torch.Tensor) else t_.cpu().numpy()


def reproject_with_depth(depth_ref, intrinsics_ref, extrinsics_ref, depth_src, intrinsics_src, extrinsics_src, device="cuda"):
    """test_dtu_dypcd.py:164-205 with the reference's arguments: project the reference pixels into the source view, sample its
    depth (cv2.remap's INTER_LINEAR arithmetic), project back -> (depth_reprojected, x_reprojected, y_reprojected, x_src, y_src),
    [h,w] float32 each.  numpy arrays in -> numpy arrays out (the reference's types); CUDA tensors in -> CUDA tensors out.  One
    kernel (``effi_fusion_dtu_reproject_f32``); PARITY UNPINNED (module docstring)."""
    dev = depth_ref.device if isinstance(depth_ref, torch.Tensor) else torch.device(device)
    with torch.cuda.device(dev):
        out5, _ = ops.fusion_dtu_reproject(_dev_map(depth_ref, dev).contiguous(), _dev_map(depth_src, dev).contiguous(),
                                           _cam_tensor(np.asarray(intrinsics_ref), np.asarray(extrinsics_ref), dev),
                                           _cam_tensor(np.asarray(intrinsics_src), np.asarray(extrinsics_src), dev))
    return tuple(_like_input(out5[k], depth_ref) for k in range(5))


def check_geometric_consistency(depth_ref, intrinsics_ref, extrinsics_ref, depth_src, intrinsics_src, extrinsics_src, confidence=None,
                                device="cuda"):
    """test_dtu_dypcd.py:208-233 with the reference's arguments (``confidence`` is unused there too) -> (masks: list of e - s bool
    maps, mask = the last of them, depth_reprojected, x2d_src, y2d_src, x2d_reprojected, y2d_reprojected), the three reprojected
    maps zeroed outside ``mask``.  Types follow the input as in ``reproject_with_depth``; PARITY UNPINNED."""
    dev = depth_ref.device if isinstance(depth_ref, torch.Tensor) else torch.device(device)
    with torch.cuda.device(dev):
        out5, m = ops.fusion_dtu_reproject(_dev_map(depth_ref, dev).contiguous(), _dev_map(depth_src, dev).contiguous(),
                                           _cam_tensor(np.asarray(intrinsics_ref), np.asarray(extrinsics_ref), dev),
                                           _cam_tensor(np.asarray(intrinsics_src), np.asarray(extrinsics_src), dev),
                                           s=s, e=e, dist_base=dist_base, diff_base=diff_base)
    masks = [_like_input(m[k].bool(), depth_ref) for k in range(m.shape[0])]
    o = [_like_input(out5[k], depth_ref) for k in range(5)]
    return masks, masks[-1], o[0], o[3], o[4], o[1], o[2]


@ops.on_tensor_device
def filter_view(ref_depth_est, ref_intrinsics, ref_extrinsics, src_depth_ests, src_intrinsics, src_extrinsics, confidence, conf=0.5):
    """The array part of ``filter_depth`` for one reference view (test_dtu_dypcd.py:257-333) on the device: depth maps [h,w] /
    [V,h,w] and confidence (any size) as CUDA tensors, cameras as numpy or tensors -> dict(depth_est_averaged [h,w], photo_mask,
    geo_mask, final_mask [h,w] bool, xyz_world [3,h,w])."""
    dev = ref_depth_est.device
    ref_cam = _cam_tensor(np.asarray(ref_intrinsics), np.asarray(ref_extrinsics), dev)
    src_cams = torch.stack([_cam_tensor(np.asarray(k), np.asarray(x), dev) for k, x in zip(src_intrinsics, src_extrinsics)])
    r = ops.fusion_dtu_filter(ref_depth_est.contiguous(), src_depth_ests.contiguous(), ref_cam, src_cams, confidence, conf_threshold=conf,
                              conf_keep=0.75, s=s, e=e, dist_base=dist_base, diff_base=diff_base)
    return {"depth_est_averaged": r["depth"], "photo_mask": r["photo_mask"].bool(), "geo_mask": r["geo_mask"].bool(),
            "final_mask": r["mask"].bool(), "xyz_world": r["points"]}


def write_ply(filename, vertexs, vertex_colors):
    """What ``PlyData([PlyElement.describe(vertex_all, 'vertex')]).write(filename)`` writes for the reference's vertex array
    (test_dtu_dypcd.py:337-349): binary PLY in native byte order, properties x y z (float) red green blue (uchar)."""
    vertex_all = np.empty(len(vertexs), dtype=[("x", "f4"), ("y", "f4"), ("z", "f4"), ("red", "u1"), ("green", "u1"), ("blue", "u1")])
    vertex_all["x"], vertex_all["y"], vertex_all["z"] = vertexs[:, 0], vertexs[:, 1], vertexs[:, 2]
    vertex_all["red"], vertex_all["green"], vertex_all["blue"] = vertex_colors[:, 0], vertex_colors[:, 1], vertex_colors[:, 2]
    order = "binary_little_endian" if sys.byteorder == "little" else "binary_big_endian"
    header = ("ply\nformat {} 1.0\nelement vertex {}\nproperty float x\nproperty float y\nproperty float z\n"
              "property uchar red\nproperty uchar green\nproperty uchar blue\nend_header\n").format(order, len(vertex_all))
    with open(filename, "wb") as f:
        f.write(header.encode("ascii"))
        f.write(vertex_all.tobytes())


def filter_depth(pair_folder, scan_folder, out_folder, plyfilename, conf=0.5, device="cuda"):
    """test_dtu_dypcd.py:236-350 with the reference's arguments (+ ``conf`` = its ``args.conf``): reads ``pair.txt``, cameras, images
    and the PFM depth / confidence maps the forward pass wrote, filters every reference view on the device, writes the three mask
    PNGs per view and the fused point cloud."""
    pair_data = read_pair_file(os.path.join(pair_folder, "pair.txt"))
    views = sorted({v for ref, srcs in pair_data for v in [ref] + srcs})
    cams = {v: read_camera_parameters(os.path.join(scan_folder, "cams/{:0>8}_cam.txt".format(v))) for v in views}
    # every map of the scan is uploaded ONCE (the reference re-reads a source view's PFM for each of its ~10 reference views)
    depth = {v: torch.from_numpy(np.ascontiguousarray(read_pfm(os.path.join(out_folder, "depth_est/{:0>8}.pfm".format(v)))[0])).to(device)
             for v in views}
    vertexs, vertex_colors = [], []
    os.makedirs(os.path.join(out_folder, "mask"), exist_ok=True)
    for ref_view, src_views in pair_data:
        ref_intrinsics, ref_extrinsics = cams[ref_view]
        ref_img = read_img(os.path.join(scan_folder, "images/{:0>8}.jpg".format(ref_view)))
        confidence = torch.from_numpy(np.ascontiguousarray(read_pfm(os.path.join(out_folder, "confidence/{:0>8}.pfm".format(ref_view)))[0])).to(device)
        r = filter_view(depth[ref_view], ref_intrinsics, ref_extrinsics, torch.stack([depth[v] for v in src_views]),
                        [cams[v][0] for v in src_views], [cams[v][1] for v in src_views], confidence, conf)
        photo_mask, geo_mask, final_mask = (r[k].cpu().numpy() for k in ("photo_mask", "geo_mask", "final_mask"))
        save_mask(os.path.join(out_folder, "mask/{:0>8}_photo.png".format(ref_view)), photo_mask)
        save_mask(os.path.join(out_folder, "mask/{:0>8}_geo.png".format(ref_view)), geo_mask)
        save_mask(os.path.join(out_folder, "mask/{:0>8}_final.png".format(ref_view)), final_mask)
        print("processing {}, ref-view{:0>2}, photo/geo/final-mask:{}/{}/{}".format(scan_folder, ref_view, photo_mask.mean(),
                                                                                    geo_mask.mean(), final_mask.mean()))
        xyz = r["xyz_world"].cpu().numpy()
        vertexs.append(xyz[:, final_mask].transpose((1, 0)))
        vertex_colors.append((ref_img[final_mask] * 255).astype(np.uint8))
    write_ply(plyfilename, np.concatenate(vertexs, axis=0), np.concatenate(vertex_colors, axis=0))
    print("saving the final model to", plyfilename)


def dypcd_filter(testlist, testpath, outdir, conf=0.5, device="cuda"):
    """test_dtu_dypcd.py:352-383: every scan of ``testlist`` (the reference fans scans out to a process pool because its filter is
    CPU-bound; on the device the scans are simply walked)."""
    for scene in testlist:
        save_name = "mvsnet{:0>3}_l3.ply".format(int(scene[4:])) if scene.startswith("scan") and scene[4:].isdigit() else "{}.ply".format(scene)
        filter_depth(os.path.join(testpath, scene), os.path.join(outdir, scene), os.path.join(outdir, scene), os.path.join(outdir, save_name),
                     conf=conf, device=device)
