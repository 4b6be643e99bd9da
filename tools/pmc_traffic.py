#!/usr/bin/env python3
"""Per-kernel effective clock (GRBM_GUI_ACTIVE / 8 / time) and HBM traffic (FETCH_SIZE / WRITE_SIZE, KB units;
on gfx950 FETCH_SIZE under-reports wide coalesced reads by 2x -- MI355X_MICROARCH.md) from rocprofv3 --pmc CSVs."""
import collections
import csv
import sys


def short(n):
    return n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:52]


for f in sys.argv[1:]:
    rows = list(csv.DictReader(open(f)))
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(set)
    dur = collections.defaultdict(float)
    for r in rows:
        k = short(r["Kernel_Name"])
        if not any(s in k for s in ("conv2d", "conv3d", "warpcorr", "deconv", "getcost", "planar", "lookup", "softmax", "pixelwise", "split", "upsample", "aggregate")):
            continue
        per[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Dispatch_Id"] not in cnt[k]:
            cnt[k].add(r["Dispatch_Id"])
            dur[k] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    print(f)
    for k, v in sorted(per.items(), key=lambda kv: -dur[kv[0]])[:26]:
        n = len(cnt[k])
        us = dur[k] / n / 1e3
        s = f"{k:54s} n={n:3d} {us:8.1f} us "
        for c, val in v.items():
            val /= n
            if c == "GRBM_GUI_ACTIVE":
                s += f" clk={val / 8 / us / 1e3:5.2f}GHz"
            elif c in ("FETCH_SIZE", "WRITE_SIZE"):
                s += f" {c}={val / 1024:8.1f}MB"
            else:
                s += f" {c}={val:10.3g}"
        print(s)
