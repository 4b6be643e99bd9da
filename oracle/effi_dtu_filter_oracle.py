"""CPU oracle for the DTU branch of scope row n3: the dynamic geometric-consistency filter + depth averaging + back-projection of
the reference's test_dtu_dypcd.py:164-350.  TEST INFRASTRUCTURE ONLY -- see oracle/effi_oracle.py for who may import this.

PARITY UNPINNED.  The reference's functions are plain numpy EXCEPT two OpenCV calls -- cv2.remap(INTER_LINEAR) (:184) and
cv2.resize (:259) -- and its module imports cv2 and plyfile at the top; neither is installed in this image and the reference holds no
fixtures for this code, so the module cannot be imported or run here.  What this file is:
  * the numpy lines restated one for one, INCLUDING their dtypes: pixel grids are int64 (np.arange), so every product with a float32
    depth map is float64 (numpy promotion) -- the whole projection chain runs in double and is rounded to float32 exactly where the
    reference calls .astype(np.float32); matrix inverses are np.linalg.inv of float32 matrices (single-precision LAPACK);
  * ``cv_remap_linear``: OpenCV 4.x's published algorithm for remap(float32 image, float32 maps, INTER_LINEAR, BORDER_CONSTANT 0)
    (modules/imgproc/src/imgwarp.cpp): coordinates are rounded to 1/32 pixel (cvRound(x * 32), round half to even), the integer part
    is saturated to int16, the four bilinear weights come from a 32 x 32 table of float32 products (1 - a/32 | a/32) x (1 - b/32 | b/32),
    taps outside the image contribute the border value 0;
  * cv2.resize of the confidence map: effi_io_oracle.resize_linear (the restatement used for scope row n4, also unpinned).
"""
import math

import numpy as np

from .effi_io_oracle import resize_linear

INTER_BITS = 5
INTER_TAB_SIZE = 1 << INTER_BITS


def cv_remap_linear(src, map_x, map_y):
    """cv2.remap(src, map_x, map_y, interpolation=cv2.INTER_LINEAR) for float32 single-channel ``src`` and float32 maps
    (borderMode = BORDER_CONSTANT, borderValue = 0: the defaults the reference relies on, test_dtu_dypcd.py:184)."""
    src = np.ascontiguousarray(src, dtype=np.float32)
    h, w = src.shape
    with np.errstate(invalid="ignore", over="ignore"):
        fx = np.nan_to_num(map_x.astype(np.float64) * INTER_TAB_SIZE, nan=-2.0 ** 31, posinf=2.0 ** 31 - 1, neginf=-2.0 ** 31)
        fy = np.nan_to_num(map_y.astype(np.float64) * INTER_TAB_SIZE, nan=-2.0 ** 31, posinf=2.0 ** 31 - 1, neginf=-2.0 ** 31)
    sx = np.clip(np.rint(fx), -2.0 ** 31, 2.0 ** 31 - 1).astype(np.int64)          # cvRound: round half to even
    sy = np.clip(np.rint(fy), -2.0 ** 31, 2.0 ** 31 - 1).astype(np.int64)
    ax, ay = (sx & (INTER_TAB_SIZE - 1)), (sy & (INTER_TAB_SIZE - 1))
    x0 = np.clip(sx >> INTER_BITS, -32768, 32767)                                  # saturate_cast<short>
    y0 = np.clip(sy >> INTER_BITS, -32768, 32767)
    scale = np.float32(1.0 / INTER_TAB_SIZE)
    tx = (ax.astype(np.float32) * scale).astype(np.float32)
    ty = (ay.astype(np.float32) * scale).astype(np.float32)
    wx0, wx1 = (np.float32(1.0) - tx).astype(np.float32), tx
    wy0, wy1 = (np.float32(1.0) - ty).astype(np.float32), ty

    def tap(yy, xx):
        ok = (yy >= 0) & (yy < h) & (xx >= 0) & (xx < w)
        return np.where(ok, src[np.clip(yy, 0, h - 1), np.clip(xx, 0, w - 1)], np.float32(0.0)).astype(np.float32)

    w00, w01 = (wy0 * wx0).astype(np.float32), (wy0 * wx1).astype(np.float32)      # table entries: float32 products
    w10, w11 = (wy1 * wx0).astype(np.float32), (wy1 * wx1).astype(np.float32)
    out = tap(y0, x0) * w00 + tap(y0, x0 + 1) * w01 + tap(y0 + 1, x0) * w10 + tap(y0 + 1, x0 + 1) * w11
    return out.astype(np.float32)


def reproject_with_depth(depth_ref, intrinsics_ref, extrinsics_ref, depth_src, intrinsics_src, extrinsics_src):
    """test_dtu_dypcd.py:164-209 -> (depth_reprojected, x_reprojected, y_reprojected, x_src, y_src), all float32 [h,w]."""
    width, height = depth_ref.shape[1], depth_ref.shape[0]
    x_ref, y_ref = np.meshgrid(np.arange(0, width), np.arange(0, height))
    x_ref, y_ref = x_ref.reshape([-1]), y_ref.reshape([-1])
    xyz_ref = np.matmul(np.linalg.inv(intrinsics_ref), np.vstack((x_ref, y_ref, np.ones_like(x_ref))) * depth_ref.reshape([-1]))
    xyz_src = np.matmul(np.matmul(extrinsics_src, np.linalg.inv(extrinsics_ref)), np.vstack((xyz_ref, np.ones_like(x_ref))))[:3]
    K_xyz_src = np.matmul(intrinsics_src, xyz_src)
    with np.errstate(divide="ignore", invalid="ignore"):
        xy_src = K_xyz_src[:2] / K_xyz_src[2:3]
    x_src = xy_src[0].reshape([height, width]).astype(np.float32)
    y_src = xy_src[1].reshape([height, width]).astype(np.float32)
    sampled_depth_src = cv_remap_linear(depth_src, x_src, y_src)
    xyz_src = np.matmul(np.linalg.inv(intrinsics_src), np.vstack((xy_src, np.ones_like(x_ref))) * sampled_depth_src.reshape([-1]))
    xyz_reprojected = np.matmul(np.matmul(extrinsics_ref, np.linalg.inv(extrinsics_src)), np.vstack((xyz_src, np.ones_like(x_ref))))[:3]
    depth_reprojected = xyz_reprojected[2].reshape([height, width]).astype(np.float32)
    K_xyz_reprojected = np.matmul(intrinsics_ref, xyz_reprojected)
    K_xyz_reprojected[2:3][K_xyz_reprojected[2:3] == 0] += 0.00001
    with np.errstate(divide="ignore", invalid="ignore"):
        xy_reprojected = K_xyz_reprojected[:2] / K_xyz_reprojected[2:3]
    x_reprojected = xy_reprojected[0].reshape([height, width]).astype(np.float32)
    y_reprojected = xy_reprojected[1].reshape([height, width]).astype(np.float32)
    return depth_reprojected, x_reprojected, y_reprojected, x_src, y_src


def check_geometric_consistency(depth_ref, intrinsics_ref, extrinsics_ref, depth_src, intrinsics_src, extrinsics_src, s=1, e=11,
                                dist_base=1 / 2, diff_base=0.25):
    """test_dtu_dypcd.py:212-236 (module constants s, e, dist_base, diff_base of :33-37 as arguments)."""
    width, height = depth_ref.shape[1], depth_ref.shape[0]
    x_ref, y_ref = np.meshgrid(np.arange(0, width), np.arange(0, height))
    depth_reprojected, x2d_reprojected, y2d_reprojected, x2d_src, y2d_src = reproject_with_depth(
        depth_ref, intrinsics_ref, extrinsics_ref, depth_src, intrinsics_src, extrinsics_src)
    with np.errstate(invalid="ignore"):
        dist = np.sqrt((x2d_reprojected - x_ref) ** 2 + (y2d_reprojected - y_ref) ** 2)
        depth_diff = np.abs(depth_reprojected - depth_ref)
        masks = []
        mask = None
        for i in range(s, e):
            mask = np.logical_and(dist < i * dist_base, depth_diff < math.log(max(i, 1.05), 10) * diff_base)
            masks.append(mask)
    depth_reprojected[~mask] = 0
    x2d_reprojected[~mask] = 0
    y2d_reprojected[~mask] = 0
    return masks, mask, depth_reprojected, x2d_src, y2d_src, x2d_reprojected, y2d_reprojected


def filter_depth_arrays(ref_depth_est, ref_intrinsics, ref_extrinsics, src_depth_ests, src_intrinsics, src_extrinsics, confidence,
                        conf=0.5, s=1, e=11, dist_base=1 / 2, diff_base=0.25):
    """The array part of filter_depth for ONE reference view (test_dtu_dypcd.py:257-333; file reads / writes left out):
    -> dict(depth_est_averaged [h,w] f32, photo_mask, geo_mask, final_mask [h,w] bool, xyz_world [3,h,w] f64 of EVERY pixel --
    the reference keeps the columns where final_mask is set)."""
    h, w = ref_depth_est.shape
    confidence = resize_linear(confidence.astype(np.float32), int(w), int(h))          # cv2.resize(confidence, (w, h)), :259
    photo_mask = confidence > conf
    all_srcview_depth_ests = []
    geo_mask_sum = 0
    dy_range = e
    geo_mask_sums = [0] * (dy_range - s)
    for depth_src, K_src, E_src in zip(src_depth_ests, src_intrinsics, src_extrinsics):
        masks, geo_mask, depth_reprojected, *_ = check_geometric_consistency(ref_depth_est, ref_intrinsics, ref_extrinsics, depth_src,
                                                                           K_src, E_src, s, e, dist_base, diff_base)
        geo_mask_sum += geo_mask.astype(np.int32)
        for i in range(s, dy_range):
            geo_mask_sums[i - s] += masks[i - s].astype(np.int32)
        all_srcview_depth_ests.append(depth_reprojected)
    depth_est_averaged = (sum(all_srcview_depth_ests) + ref_depth_est) / (geo_mask_sum + 1)
    depth_est_averaged[confidence > 0.75] = ref_depth_est[confidence > 0.75]
    geo_mask = geo_mask_sum >= dy_range
    for i in range(s, dy_range):
        geo_mask = np.logical_or(geo_mask, geo_mask_sums[i - s] >= i)
    final_mask = np.logical_and(photo_mask, geo_mask)
    x, y = np.meshgrid(np.arange(0, w), np.arange(0, h))
    x, y, depth = x.reshape(-1), y.reshape(-1), depth_est_averaged.reshape(-1)
    xyz_ref = np.matmul(np.linalg.inv(ref_intrinsics), np.vstack((x, y, np.ones_like(x))) * depth)
    xyz_world = np.matmul(np.linalg.inv(ref_extrinsics), np.vstack((xyz_ref, np.ones_like(x))))[:3]
    return {"depth_est_averaged": depth_est_averaged.astype(np.float32), "photo_mask": photo_mask, "geo_mask": geo_mask,
            "final_mask": final_mask, "xyz_world": xyz_world.reshape(3, h, w), "confidence_resized": confidence}
