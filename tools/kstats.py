#!/usr/bin/env python3
"""tools/kstats.py DIR -- print the g1s kernels of a rocprofv3 --kernel-trace --stats CSV run."""
import csv
import glob
import sys

d = sys.argv[1]
f = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)[0]
tot = 0.0
for r in csv.DictReader(open(f)):
    if "g1s" in r["Name"]:
        tot += float(r["TotalDurationNs"])
        print(f'{r["Name"][:64]:64s} calls {r["Calls"]:>4s}  avg {float(r["AverageNs"]) / 1e3:9.1f} us  total {float(r["TotalDurationNs"]) / 1e3:10.1f} us')
print(f"g1s kernels total {tot / 1e3:.1f} us")
