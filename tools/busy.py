#!/usr/bin/env python3
"""tools/busy.py DIR -- from a rocprofv3 --kernel-trace run: how much of the steady-state wall time has at least one g1s kernel
running (the rest is GPU idle: launch gaps, host back-pressure), and the mean number of kernels running at once."""
import csv, glob, sys
d = sys.argv[1]
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(open(f)) if "g1s" in r["Kernel_Name"]]
rows.sort()
n = len(rows)
rows = rows[n // 4: n - n // 8]  # steady state
t0, t1 = rows[0][0], max(e for _, e in rows)
ev = []
for s, e in rows:
    ev.append((s, 1)); ev.append((e, -1))
ev.sort()
busy = 0; depth = 0; last = t0; area = 0
for t, dlt in ev:
    if depth > 0: busy += t - last
    area += depth * (t - last)
    depth += dlt; last = t
print(f"window {(t1 - t0) / 1e6:.2f} ms, busy {busy / (t1 - t0):.3f}, mean kernels in flight {area / (t1 - t0):.2f}")
gaps = []
cur_end = rows[0][1]
for s, e in rows[1:]:
    if s > cur_end: gaps.append(s - cur_end)
    cur_end = max(cur_end, e)
gaps.sort(reverse=True)
print("idle gaps: count", len(gaps), "total ms", sum(gaps) / 1e6, "largest us", [round(g / 1e3, 1) for g in gaps[:8]])
