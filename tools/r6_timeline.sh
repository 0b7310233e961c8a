#!/bin/bash
# tools/r6_timeline.sh TAG [env...] -- rocprofv3 --kernel-trace of the PIPELINED bench job (default streams): the launches of two
# batches from the middle of a timed step as a timeline (start, end, duration, queue): where the finder chain of batch N + 1 really
# runs relative to the accumulation launches of batch N
set -u
TAG=${1:-r06_tl}; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd "$ROOT"; mkdir -p gpurun_out/$TAG
(cd /tmp && TMPDIR=/tmp env "$@" timeout 300 rocprofv3 --output-format csv -d $ROOT/gpurun_out/$TAG -o p --kernel-trace -- python $ROOT/bench.py --steps 1 --warmup 1 --cycles 8 --no-cpu-baseline --no-all-flat > $ROOT/gpurun_out/$TAG/run.log 2>&1 < /dev/null)
python - <<PY
import csv, glob
f = glob.glob("gpurun_out/$TAG/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "g1s" in r["Kernel_Name"] or "ALL" in "${TL_ALL:-}"]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = len(rows)
mid = rows[n * 3 // 8 - 10: n * 3 // 8 + 20]
t0 = int(mid[0]["Start_Timestamp"])
for r in mid:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("void ", "").replace("g1s::", "").split("(")[0]
    print(f'{(s - t0) / 1e3:10.1f} {(e - t0) / 1e3:10.1f} {(e - s) / 1e3:8.1f}  q{r.get("Queue_Id", "?"):>3s}  {name}')
PY
find gpurun_out -name "*kernel_trace.csv" -size +8M -delete
