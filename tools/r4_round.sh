#!/bin/bash
# tools/r4_round.sh TAG -- the measurement set of round 4 on the GPU box (as tools/profile_round.sh, for the wide chain):
#   gpurun_out/TAG_bench.json, TAG_kt1 (rocprofv3 --kernel-trace --stats, one stream), TAG_kt (default streams), PMC traffic passes
set -u
TAG=${1:-r04}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd "$ROOT"
mkdir -p gpurun_out
python tools/ktime.py 2 > /dev/null 2>&1   # (warm the box: the first process runs slow)
timeout 900 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
tail -c 400 gpurun_out/${TAG}_bench.json; echo
B="python $ROOT/bench.py --steps 2 --warmup 1 --cycles 8 --no-cpu-baseline --no-all-flat --no-table-check"
bash tools/prof.sh ${TAG}_kt --kernel-trace --stats -- $B > /dev/null
G1S_ONE_STREAM=1 G1S_D2H_SYNC=1 bash tools/prof.sh ${TAG}_kt1 --kernel-trace --stats -- $B > /dev/null
for c in FETCH_SIZE WRITE_SIZE; do
  n=$(echo $c | tr 'A-Z' 'a-z' | sed 's/_size//')
  bash tools/prof.sh ${TAG}_$n --pmc $c -- python $ROOT/tools/diff_pmc.py 2 > /dev/null
done
for d in kt kt1; do echo "== $d"; python tools/kstats.py gpurun_out/${TAG}_$d | tee gpurun_out/${TAG}_kernel_stats_$d.txt; done
for d in fetch write; do echo "== $d"; python tools/pmc_summary.py gpurun_out/${TAG}_$d | grep -v "^==" | tee gpurun_out/${TAG}_pmc_$d.txt; done
find gpurun_out -name "*kernel_trace.csv" -size +8M -delete
find gpurun_out -name "*counter_collection.csv" -size +8M -delete
