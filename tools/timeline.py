#!/usr/bin/env python3
"""tools/timeline.py DIR [N [SKIP_FROM_END]] -- print the last N g1s kernel launches of a rocprofv3 --kernel-trace run as a
timeline (start relative to the first shown, duration, stream / queue)."""
import csv, glob, sys
d = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "g1s" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
off = int(sys.argv[3]) if len(sys.argv) > 3 else 0  # skip that many launches from the end (the timed, serial job of bench.py)
rows = rows[len(rows) - n - off: len(rows) - off]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("void ", "").replace("g1s::", "").split("(")[0]
    print(f'{(s - t0) / 1e3:10.1f} {(e - s) / 1e3:8.1f}  q{r.get("Queue_Id", "?"):>3s}  {name}')
