#!/usr/bin/env python3
"""Print the kernel timeline of the LAST hot-path step in a rocprofv3 --kernel-trace CSV
(one line per launch: duration, gap to the previous kernel, grid, kernel name)."""
import csv
import sys


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return n.split("(")[0][:64]


def main(path, summary_only=False):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    idx = [i for i, r in enumerate(rows) if "stage1_hypotheses" in r["Kernel_Name"]]
    a = idx[-1]
    last_end = int(rows[a]["Start_Timestamp"])
    t0 = last_end
    busy = gaps = 0
    agg = {}
    for r in rows[a:]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        gap = max(0, s - last_end)
        gaps += gap
        busy += e - s
        last_end = max(last_end, e)
        k = short(r["Kernel_Name"])
        d = agg.setdefault(k, [0, 0.0])
        d[0] += 1
        d[1] += (e - s) / 1e3
        if not summary_only:
            blocks = (int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"])) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
            print(f"{(e - s) / 1e3:9.1f} us  gap {gap / 1e3:6.1f}  blocks {blocks:6d}  {k}")
    print(f"--- {len(rows) - a} kernels, busy {busy / 1e6:.3f} ms, gaps {gaps / 1e6:.3f} ms, span {(last_end - t0) / 1e6:.3f} ms")
    for k, (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{us:9.1f} us  x{n:3d}  {k}")


if __name__ == "__main__":
    main(sys.argv[1], len(sys.argv) > 2)
