#!/usr/bin/env python3
"""Per-kernel HBM traffic from rocprofv3 --pmc passes -> profiles/pmc_traffic_<workload>.json (read by bench.py).

usage: tools/pmc_traffic.py FETCH_counter_collection.csv WRITE_counter_collection.csv out.json

FETCH_SIZE / WRITE_SIZE are reported in KB.  MI355X_MICROARCH.md (HBM section): on gfx950 FETCH_SIZE counts 128-B
requests at 64 B, i.e. it reads exactly HALF of the bytes of a wide coalesced read stream; WRITE_SIZE is exact; other
access shapes must be calibrated on a known byte count.  tools/microbench/fetch_calib.hip (tools/pmc_calibrate.sh,
profiles/r02_fetch_calibration.txt) reads 1 GiB exactly once in the access shapes of this repository's kernels; true bytes /
FETCH_SIZE came out as

    16 B per lane, 1 KiB per wave-instruction ................ 2.00      4 B per lane, 256 B per wave-instruction .. 2.00
    16 B per lane in 96-byte row segments (16-px tiles) ...... 1.00      160-byte segments ......................... 1.25
    288-byte segments (the 4 x 64 "wide" tiles) .............. 1.50      scattered 32-byte pieces (C = 8 taps) ..... 0.50

(the counter tallies 64 B per request whatever the request moved: 128-B requests read half, 32-B pieces read double).  The
factor of each kernel family below is the one of its dominant read shape (`fetch_scale`); `calibrated` says so.
Also prints the effective clock (GRBM_GUI_ACTIVE / 8 / time) when that counter is present.
"""
import collections
import csv
import json
import re
import sys


def key_of(name):
    n = name.replace("(anonymous namespace)::", "").replace("void ", "")
    m = re.match(r"conv2d_mfma_v2_kernel<(\d+), (\d+), (\d+), (\d+)", n)
    if m:
        return f"conv2d_k{m.group(1)}_nt{m.group(2)}_epi{m.group(4)}", 1.0          # 16-px tiles: ~96-byte row segments
    m = re.match(r"conv2d_k3_bf16x3_kernel<(\d+), (\d+), (\d+), (true|false), (true|false)(?:, (true|false))?", n)
    if m:
        wide, sr = m.group(5) == "true", m.group(6) == "true"
        # fp32 maps: (16 MR + 8)-pixel row segments of 4 bytes; split-resident maps: (16 MR + 2) pixels of 16 bytes = 288-byte segments
        # for the 16-wide tiles (calibrated x1.5), > 1 KB for the 4 x 64 tiles (long runs: x2)
        scale = ((2.0 if wide else 1.5) if sr else ({1: 1.0, 2: 1.25, 4: 1.5}.get(int(m.group(2)), 1.0) if wide else 1.0))
        epi = m.group(3)
        name = {"6": f"conv2d_k3k1_nt{m.group(1)}", "8": f"conv2d_k3k1up_nt{m.group(1)}"}.get(epi, f"conv2d_k3x3_nt{m.group(1)}_epi{epi}")
        return (name if m.group(4) == "false" else f"conv3d_x3_nt{m.group(1)}"), scale
    m = re.match(r"conv2d_k3_bf16x3_pair_kernel<(\d+), (\d+), (true|false)(?:, (true|false))?", n)
    if m:
        wide, sr = m.group(3) == "true", m.group(4) == "true"
        return f"conv2d_k3x3_pair_nt{m.group(1)}", ((2.0 if wide else 1.5) if sr else ({1: 1.0, 2: 1.25, 4: 1.5}.get(int(m.group(2)), 1.0) if wide else 1.0))
    m = re.match(r"conv2d_k3_bf16x3_encgen_pair_kernel<(\d+)", n)
    if m:
        # reads: 4-byte lanes along the rows of the cost volumes / the inverse-depth map (x2, as encoder_inputs); its writes are
        # the 16-byte split-resident stores
        return f"encgen_pair_nt{m.group(1)}", 2.0
    if n.startswith("encoder_inputs_kernel"):
        return "encoder_inputs", 2.0                                                   # 4 B per lane, 256-byte runs of the cost volumes
    m = re.match(r"conv3d_roll(?:_rp)?_bf16x3_pair_kernel<(\d+)", n)
    if m:
        return f"conv3d_roll_pair_oct{m.group(1)}", 1.0                                # 96-byte segments
    m = re.match(r"conv3d_k3_pair_kernel<(\d+), (\d+), (\d+)", n)
    if m:
        return f"conv3d_pair_c{m.group(1)}_s{m.group(2)}{m.group(3)}", 1.0
    if n.startswith("deconv3d_k3_pair_kernel"):
        return "deconv3d_pair_c1_s1", 1.0
    m = re.match(r"conv3d_roll_bf16x3_kernel<(\d+), (\d+)", n)
    if m:
        return f"conv3d_roll_oct{m.group(1)}_nt{m.group(2)}", 1.0
    m = re.match(r"conv3d_roll_rp_bf16x3_kernel<(\d+)", n)                             # row-pair form (cout <= 8): one N-tile
    if m:
        return f"conv3d_roll_oct{m.group(1)}_nt1", 1.0
    if n.startswith("head_update_kernel"):
        return "head_update", 2.0                                                      # float4 per lane streams
    m = re.match(r"deconv3d_s2_bf16x3_kernel", n)
    if m:
        return "deconv3d_x3", 1.0                                                      # 96-byte segments: counted exactly
    m = re.match(r"getcost_conv1x1_kernel", n)
    if m:
        return "getcost_conv1x1", 2.0
    m = re.match(r"conv2d_mfma_kernel<(\d+), (\d+), (\d+)", n)
    if m:
        return f"conv2d_k{m.group(1)}_nt{m.group(2)}_epi{m.group(3)}", 1.0
    m = re.match(r"conv2d_cout1_k3_kernel<(\d+)", n)
    if m:
        return f"conv2d_k3_nt1_epi{m.group(1)}", 1.0
    m = re.match(r"conv3d_k3_kernel<(\d+), (\d+), (\d+)", n)
    if m:
        return f"conv3d_c{m.group(1)}_s{m.group(2)}{m.group(3)}", 1.0
    m = re.match(r"deconv3d_k3_kernel<(\d+), (\d+)", n)
    if m:
        return f"deconv3d_c{1 if m.group(1) == '1' else 8}_s{m.group(2)}", 1.0
    if n.startswith("warpcorr_views_win_kernel"):
        return "warpcorr_views_c32", 2.0                                               # window rows: long contiguous runs
    m = re.match(r"warpcorr_dyn_win_kernel<(\d+)", n)                                  # LDS-window form: window rows copied as contiguous runs
    if m:
        return f"warpcorr_dyn_c{m.group(1)}", 2.0
    m = re.match(r"warpcorr_dyn_hyp_kernel<(\d+)", n)                                  # a lane reads the C * 4 contiguous bytes of a tap
    if m:
        return f"warpcorr_dyn_c{m.group(1)}", {16: 1.0, 8: 0.5}[int(m.group(1))]
    m = re.match(r"warpcorr_(views|dyn)_kernel<(\d+)", n)
    if m:
        return f"warpcorr_{m.group(1)}_c{m.group(2)}", {32: 2.0, 16: 1.0, 8: 0.5}[int(m.group(2))]     # taps of 128 / 64 / 32 bytes
    return "other:" + n.split("(")[0].split("<")[0][:48], 1.0          # everything else of a view (uncalibrated): the per-view total is complete


# families whose dominant read shape is one of the calibrated ones (see the module docstring)
CALIBRATED = {"encoder_inputs", "deconv3d_x3", "getcost_conv1x1", "warpcorr_views_c32", "warpcorr_dyn_c8", "warpcorr_dyn_c16", "head_update"}
CALIBRATED_PREFIXES = ("conv2d_k3x3", "conv3d_x3", "conv3d_roll", "conv2d_k3k1", "encgen_pair")


def collect(path):
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(set)
    dur = collections.defaultdict(float)
    for r in csv.DictReader(open(path)):
        k, scale = key_of(r["Kernel_Name"])
        per[k][r["Counter_Name"]] += float(r["Counter_Value"])
        per[k]["_scale"] = scale
        if r["Dispatch_Id"] not in disp[k]:
            disp[k].add(r["Dispatch_Id"])
            dur[k] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    return per, disp, dur


def main(fetch_csv, write_csv, out_json):
    pf, df, tf = collect(fetch_csv)
    pw, dw, _ = collect(write_csv)
    out = {}
    for k in sorted(pf, key=lambda k: -tf[k]):
        n = len(df[k])
        raw = pf[k]["FETCH_SIZE"] * 1024.0 / n
        scale = pf[k]["_scale"]
        wr = pw[k]["WRITE_SIZE"] * 1024.0 / max(len(dw[k]), 1) if k in pw else 0.0
        us = tf[k] / n / 1e3
        clk = pf[k].get("GRBM_GUI_ACTIVE", 0.0) / n / 8 / us / 1e3 if us else 0.0
        hit = pw[k].get("TCC_HIT_sum", 0.0)
        miss = pw[k].get("TCC_MISS_sum", 0.0)
        out[k] = {"launches_profiled": n, "avg_launch_us_profiled": us, "fetch_bytes_raw": raw, "fetch_scale": scale,
                  "fetch_bytes": raw * scale, "write_bytes": wr, "calibrated": k in CALIBRATED or k.startswith(CALIBRATED_PREFIXES),
                  "l2_hit_rate": hit / (hit + miss) if hit + miss else None}
        print(f"{k:24s} n={n:3d} {us:8.1f} us  fetch {raw * scale / 1e6:8.1f} MB (raw {raw / 1e6:7.1f}) write {wr / 1e6:7.1f} MB"
              f"  L2 hit {100 * (hit / (hit + miss) if hit + miss else 0):5.1f}%  clk~{clk:4.2f} GHz")
    # reference views in the profiled run: one stage1_hypotheses launch each
    views = len(df.get("other:stage1_hypotheses_kernel", ()))
    out["_meta"] = {"views": views, "note": "launches_profiled / views = launches per view; bench.py's roofline.traffic_per_view sums "
                                            "(fetch_bytes + write_bytes) x launches over all kernels / views"}
    tot = sum((v["fetch_bytes"] + v["write_bytes"]) * v["launches_profiled"] for k, v in out.items() if k != "_meta")
    print(f"views profiled: {views}; traffic per view {tot / max(views, 1) / 1e6:.1f} MB")
    with open(out_json, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main(*sys.argv[1:4])
