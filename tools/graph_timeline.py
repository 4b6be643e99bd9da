#!/usr/bin/env python3
"""Timeline of ONE hipGraph replay of the hot path in a rocprofv3 --kernel-trace CSV: start offset, duration, how many
other kernels run at the same time.  usage: graph_timeline.py kernel_trace.csv [replay index from the end, default 2]"""
import csv
import sys


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return n.split("(")[0][:60]


def main(path, back=2):
    rows = [r for r in csv.DictReader(open(path))]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    idx = [i for i, r in enumerate(rows) if "stage1_hypotheses" in r["Kernel_Name"]]
    a, b = idx[-back - 1], idx[-back]
    seg = rows[a:b]
    t0 = int(seg[0]["Start_Timestamp"])
    ev = [(int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0, short(r["Kernel_Name"])) for r in seg]
    end = max(e for _, e, _ in ev)
    # union busy time and time with >= 2 kernels
    pts = sorted([(s, 1) for s, _, _ in ev] + [(e, -1) for _, e, _ in ev])
    busy = both = 0
    lvl, last = 0, 0
    for t, d in pts:
        if lvl >= 1:
            busy += t - last
        if lvl >= 2:
            both += t - last
        lvl += d
        last = t
    for s, e, k in ev:
        conc = sum(1 for s2, e2, _ in ev if s2 < e and e2 > s) - 1
        print(f"{s / 1e3:9.1f} us  +{(e - s) / 1e3:7.1f}  overlap {conc}  {k}")
    print(f"--- {len(ev)} kernels, span {end / 1e6:.3f} ms, busy {busy / 1e6:.3f} ms, >=2 kernels {both / 1e6:.3f} ms, sum of durations "
          f"{sum(e - s for s, e, _ in ev) / 1e6:.3f} ms")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 2)
