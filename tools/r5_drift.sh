#!/bin/bash
# tools/r5_drift.sh TAG -- VERDICT r04 item 4: the driver's bench line, as the FIRST GPU process of a fresh box, with the GPU's
# clocks / power / busy and the host's CPU use sampled next to it; then the same command again (a warm box).
set -u
TAG=${1:-r05_drift}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd "$ROOT"
mkdir -p gpurun_out
for run in first second; do
  python tools/gpu_sampler.py gpurun_out/${TAG}_${run}_samples.txt 0.25 &
  SP=$!
  sleep 1
  date +%s.%N > gpurun_out/${TAG}_${run}_t0.txt
  timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-all-flat > gpurun_out/${TAG}_${run}.json 2> gpurun_out/${TAG}_${run}.err
  kill $SP; wait $SP 2>/dev/null
  python - <<PY
import json
j = json.load(open("gpurun_out/${TAG}_${run}.json"))
print("${run}: value %.0f  step_ms %s" % (j["value"], j["step_ms"]))
print("   host_fold_ms per step", j.get("step_host_fold_ms"))
PY
done
cat /sys/fs/cgroup/cpu.max 2>/dev/null; nproc
