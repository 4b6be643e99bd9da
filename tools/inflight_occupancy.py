#!/usr/bin/env python3
"""How busy is the GPU while views are in flight?  Reads a rocprofv3 kernel trace (CSV) of `bench.py --in-flight N` and reports, over
the densest window of the run (the timed replays), the union of kernel intervals (some kernel running), the time with >= 2 kernels
running, the idle time and the sum of durations.   usage: inflight_occupancy.py <p_kernel_trace.csv> [window_ms]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
win_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 30.0
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows)
def stats(lo, hi):
    sel = [(max(a, lo), min(b, hi)) for a, b in iv if b > lo and a < hi]
    events = []
    for a, b in sel:
        events.append((a, 1))
        events.append((b, -1))
    events.sort()
    busy = multi = depth = 0
    prev = lo
    for t, d in events:
        if depth >= 1:
            busy += t - prev
        if depth >= 2:
            multi += t - prev
        depth += d
        prev = t
    span = hi - lo
    return len(sel), busy / span, multi / span, sum(b - a for a, b in sel) / span


# slide a window over the run and report the one with the most concurrent execution (= the in-flight replays)
w = int(win_ms * 1e6)
t0, t1 = iv[0][0], iv[-1][1]
best = None
t = t0
while t + w <= t1:
    r = stats(t, t + w)
    if best is None or r[2] > best[1][2]:
        best = (t, r)
    t += w // 4
n, busy, multi, dens = best[1]
print(f"window of {win_ms:.0f} ms with the most concurrency (at +{(best[0] - t0) / 1e6:.0f} ms): {n} kernels, some kernel running {100 * busy:.1f} %, "
      f">= 2 running {100 * multi:.1f} %, idle {100 * (1 - busy):.1f} %, sum of durations / window = {dens:.2f}")
