#!/bin/bash
# the round's measurement set in one call (tools/profile_round.sh + the other workloads + the pipelined timeline + the host budget)
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
bash tools/profile_round.sh r03 > gpurun_out/r03_round.log 2>&1
bash tools/other_workloads.sh
G1S_K3=stream bash tools/prof.sh tl_0 --kernel-trace -- python $PWD/bench.py --steps 2 --warmup 1 --cycles 8 --no-cpu-baseline --no-all-flat > /dev/null
python tools/timeline.py gpurun_out/tl_0 60 400 > gpurun_out/r03_timeline_default_streams.txt
find gpurun_out/tl_0 -name "*.csv" -size +1M -delete
{
echo "tools/bench_fold.py 3840x2160 on the GPU box's host: CPU-seconds per frame of the two host halves"
timeout 600 python tools/bench_fold.py 2>/dev/null
for m in 1 8 16; do
  echo "== ordered merge, stage timers, merge pool of $m threads (G1S_MERGE_POOL=$m)"
  G1S_FOLD_PROFILE=1 G1S_FOLD_THREADS=32 G1S_MERGE_POOL=$m python tools/bench_fold.py 3840x2160 merge-only 2>&1 | tail -3
done
} > gpurun_out/r03_fold_budget.txt 2>&1
tail -30 gpurun_out/r03_round.log; cat gpurun_out/other_workloads.txt; cat gpurun_out/r03_fold_budget.txt
