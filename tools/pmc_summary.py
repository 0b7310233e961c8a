#!/usr/bin/env python3
"""Summarise rocprofv3 CSV output (kernel trace and/or counter collection) per kernel name."""
import csv, glob, sys, collections, os
d = sys.argv[1]
filt = sys.argv[2] if len(sys.argv) > 2 else "g1s"
for f in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    seen = set()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if filt not in k: continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (k, r["Dispatch_Id"])
        if key not in seen: seen.add(key); cnt[k] += 1
    for k in agg:
        print(k[:70], "dispatches", cnt[k])
        for c, v in sorted(agg[k].items()): print(f"   {c:32s} {v / cnt[k]:16.1f} per dispatch")
for f in sorted(glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)):
    print("== kernel_stats", f)
    for r in csv.DictReader(open(f)):
        if filt in r["Name"]: print("  ", r["Name"][:60], "calls", r["Calls"], "avg_ns", r["AverageNs"], "total_ns", r["TotalDurationNs"], "pct", r["Percentage"])
