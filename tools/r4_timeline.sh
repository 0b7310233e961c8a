cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
python tools/ktime.py 2 > /dev/null 2>&1
python tools/ktime.py 4 2>/dev/null | tail -1
bash tools/prof.sh r04_timeline --kernel-trace --stats -- python $PWD/tools/ktime.py 4 > /dev/null
python tools/kstats.py gpurun_out/r04_timeline
python - <<'PY'
import csv,glob
f=glob.glob("gpurun_out/r04_timeline/**/*kernel_trace.csv",recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if "g1s" in r["Kernel_Name"]]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
t0=int(rows[0]["Start_Timestamp"])
for r in rows[-24:]:
    print("%-40s start %9.1f us dur %8.1f us"%(r["Kernel_Name"][:40],(int(r["Start_Timestamp"])-t0)/1e3,(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3))
PY
find gpurun_out -name "*kernel_trace.csv" -size +8M -delete
