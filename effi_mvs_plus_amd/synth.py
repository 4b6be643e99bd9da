"""Synthetic inputs and seeded-random weights (no datasets or checkpoints travel to the GPU box).

``synth_sample`` is the rig fixed in SURVEY.md section 8(d); it reproduces the input contract of the
reference's dataset classes (datasets/general_eval.py:60-228): ``imgs [B,N,3,H,W]``,
``proj_matrices {stageK: [B,N,2,4,4]}`` with ``[:,:,0]`` the 4x4 extrinsic and ``[:,:,1,:3,:3]`` the
intrinsic, and ``depth_values [B,384]`` holding ascending inverse depths.
"""
from __future__ import annotations

import math
import re
import zlib

import torch

NUM_DEPTH_VALUES = 384
DEPTH_MIN_MM = 425.0
DEPTH_MAX_MM = 935.0


def synth_cameras(H, W, N, dtype=torch.float32):
    """Per-stage projection dict for N views on a ring around the scene."""
    fx = fy = 2892.33 * (W / 1600.0) / 4.0
    cx, cy = W / 8.0, H / 8.0
    ext = torch.zeros(N, 4, 4, dtype=torch.float64)
    for v in range(N):
        a = math.radians(6.0) * math.ceil(v / 2) * (1.0 if v % 2 == 1 else -1.0)
        if v == 0:
            a = 0.0
        ca, sa = math.cos(a), math.sin(a)
        R = torch.tensor([[ca, 0.0, sa], [0.0, 1.0, 0.0], [-sa, 0.0, ca]], dtype=torch.float64)
        C = torch.tensor([680.0 * sa, 10.0 * v, 680.0 * (1.0 - ca)], dtype=torch.float64)
        ext[v, :3, :3] = R
        ext[v, :3, 3] = -R @ C
        ext[v, 3, 3] = 1.0
    out = {}
    for name, scale in (("stage0", 0.25), ("stage1", 0.5), ("stage2", 1.0), ("stage3", 2.0), ("stage4", 4.0)):
        K = torch.zeros(N, 4, 4, dtype=torch.float64)
        K[:, 0, 0] = fx * scale
        K[:, 1, 1] = fy * scale
        K[:, 0, 2] = cx * scale
        K[:, 1, 2] = cy * scale
        K[:, 2, 2] = 1.0
        out[name] = torch.stack([ext, K], dim=1).unsqueeze(0).to(dtype)       # [1,N,2,4,4]
    return out


def synth_sample(H, W, N, seed=0, dtype=torch.float32):
    """(imgs [1,N,3,H,W], proj_matrices dict, depth_values [1,384]) on CPU."""
    g = torch.Generator().manual_seed(seed)
    imgs = torch.rand(1, N, 3, H, W, generator=g, dtype=torch.float32).to(dtype)
    depth_values = torch.linspace(1.0 / DEPTH_MAX_MM, 1.0 / DEPTH_MIN_MM, NUM_DEPTH_VALUES,
                                  dtype=torch.float64).to(dtype).unsqueeze(0)
    return imgs, synth_cameras(H, W, N, dtype), depth_values


def smooth_features(N, C, h, w, seed, dtype=torch.float32):
    """Seeded feature maps with spatial correlation (so that a warp onto the right depth actually
    correlates): low-res noise bilinearly upsampled plus a little white noise."""
    g = torch.Generator().manual_seed(seed)
    lo = torch.randn(N, C, max(h // 4, 2), max(w // 4, 2), generator=g)
    up = torch.nn.functional.interpolate(lo, size=(h, w), mode="bilinear", align_corners=True)
    up = up + 0.1 * torch.randn(N, C, h, w, generator=g)
    return [up[v:v + 1].contiguous().to(dtype) for v in range(N)]


def _canonical_key(k):
    """The reference registers some modules twice (update_block.N == update_block_depthN+1,
    CSP_R.N == CSP_RN+1, CSP_C.N == CSP_CN+1); aliases must receive identical values."""
    k = re.sub(r"^update_block_depth(\d)\.", lambda m: "update_block.%d." % (int(m.group(1)) - 1), k)
    k = re.sub(r"^CSP_([RC])(\d)\.", lambda m: "CSP_%s.%d." % (m.group(1), int(m.group(2)) - 1), k)
    return k


def randomize_state_dict(sd, seed=0):
    """Overwrite every entry of ``sd`` (any model with the reference's key naming) with seeded
    values: He-scaled conv weights, non-trivial biases, and non-trivial BatchNorm affine/running
    statistics so that BN folding is exercised.  Returns a new dict of CPU tensors."""
    out = {}
    for k in sd:
        v = sd[k]
        shape = tuple(v.shape)
        # one generator per key: the values do not depend on the order modules were registered in
        g = torch.Generator().manual_seed((seed * 1000003 + zlib.crc32(_canonical_key(k).encode())) % (2 ** 31))
        if k.endswith("num_batches_tracked"):
            out[k] = torch.tensor(100, dtype=torch.int64)
        elif k.endswith("running_var"):
            out[k] = torch.rand(shape, generator=g) + 0.5
        elif k.endswith("running_mean"):
            out[k] = 0.1 * torch.randn(shape, generator=g)
        elif ".bn." in k and k.endswith("weight"):
            out[k] = 0.6 + 0.8 * torch.rand(shape, generator=g)
        elif ".bn." in k and k.endswith("bias"):
            out[k] = 0.1 * torch.randn(shape, generator=g)
        elif k.endswith("bias"):
            out[k] = 0.05 * torch.randn(shape, generator=g)
        else:
            if len(shape) >= 3 and ("conv6" in k or "conv7" in k or k.endswith("conv2.conv.weight") and "CSP" in k):
                fan_in = shape[0] * math.prod(shape[2:]) / 4.0   # transposed conv: [Cin,Cout,k..], ~1/4..1/8 of taps hit
            else:
                fan_in = math.prod(shape[1:]) if len(shape) > 1 else shape[0]
            w = torch.randn(shape, generator=g) * math.sqrt(2.0 / max(fan_in, 1))
            if "depth_head.conv2" in k:
                w = w * 0.05        # keep GRU deltas to a few percent of the inverse-depth range
            out[k] = w
    return out


def synth_depth_maps(H, W, N, seed=0, noise_mm=0.4, outlier_frac=0.05, pixel_center=0.5):
    """Depth maps of ONE analytic scene (a tilted plane with a smooth bump field evaluated per view) for the ring of
    ``synth_cameras``: (depths [N,H,W] fp32, cams [N,2,4,4] fp32 at full resolution "stage3").  Used by the depth-fusion
    tests / bench: views agree up to ``noise_mm`` except in random blocks (``outlier_frac`` of the image) that are offset by
    several mm, so that the consistency masks are neither empty nor full."""
    g = torch.Generator().manual_seed(seed)
    cams = synth_cameras(H, W, N, torch.float64)["stage3"][0]          # [N,2,4,4]: extrinsic, intrinsic
    nrm = torch.tensor([0.08, -0.05, 1.0], dtype=torch.float64)
    nrm = nrm / nrm.norm()
    dpl = float(nrm[2]) * 680.0
    # pixel_center: where a pixel's ray passes -- 0.5 for the Tanks-and-Temples filter (misc/fusion.py adds 0.5), 0 for the DTU
    # filter (test_dtu_dypcd.py:169-173 uses the integer grid)
    ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float64) + pixel_center, torch.arange(W, dtype=torch.float64) + pixel_center,
                            indexing="ij")
    pix = torch.stack([xs, ys, torch.ones_like(xs)], 0).reshape(3, -1)
    depths = []
    for v in range(N):
        E, K = cams[v, 0], cams[v, 1, :3, :3]
        R, t = E[:3, :3], E[:3, 3]
        C = -R.t() @ t
        dirs = R.t() @ (torch.linalg.inv(K) @ pix)                       # world direction per unit camera depth
        tt = (dpl - nrm @ C) / (nrm @ dirs)                               # camera-frame depth of the plane hit
        d = tt.reshape(H, W).float()
        d = d + noise_mm * torch.randn(H, W, generator=g)
        blocks = torch.rand(max(H // 16, 1), max(W // 16, 1), generator=g) < outlier_frac
        off = (torch.rand(blocks.shape, generator=g) * 12.0 + 3.0) * blocks
        d = d + torch.nn.functional.interpolate(off[None, None], size=(H, W), mode="nearest")[0, 0]
        depths.append(d)
    return torch.stack(depths), cams.float()
