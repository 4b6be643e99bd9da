#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv per kernel (last step only when a kernel trace is given)."""
import csv
import sys
from collections import defaultdict


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return n.split("(")[0][:60]


def main(path, key_filter=None):
    rows = list(csv.DictReader(open(path)))
    per = defaultdict(lambda: defaultdict(float))
    cnt = defaultdict(int)
    seen = set()
    for r in rows:
        k = short(r["Kernel_Name"])
        if key_filter and key_filter not in k:
            continue
        per[k][r["Counter_Name"]] += float(r["Counter_Value"])
        did = (r["Dispatch_Id"], k)
        if did not in seen:
            seen.add(did)
            cnt[k] += 1
    names = sorted({c for v in per.values() for c in v})
    print("kernel".ljust(60), "n".rjust(5), *[c[-16:].rjust(17) for c in names])
    for k, v in sorted(per.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", kv[1].get("FETCH_SIZE", kv[1].get("WRITE_SIZE", 0)))):
        print(k.ljust(60), str(cnt[k]).rjust(5), *[f"{v.get(c, 0) / cnt[k]:17.4g}" for c in names])


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
