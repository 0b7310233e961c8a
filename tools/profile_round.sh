#!/bin/bash
# tools/profile_round.sh TAG -- the measurement set of a round, on the GPU box:
#   gpurun_out/TAG_bench.json           the default bench.py line
#   gpurun_out/TAG_kt/                  rocprofv3 --kernel-trace --stats of the same command (shorter job, no all-flat variant)
#   gpurun_out/TAG_kt1/                 same on one stream with the records copy ended before the next batch starts
#                                       (G1S_ONE_STREAM=1 G1S_D2H_SYNC=1: every kernel alone on the chip, no blit next to it)
#   gpurun_out/TAG_fetch/, TAG_write/   PMC passes (FETCH_SIZE, WRITE_SIZE), each in its own run
set -u
TAG=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd "$ROOT"
mkdir -p gpurun_out
timeout 900 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
tail -c 600 gpurun_out/${TAG}_bench.json; echo
B="python $ROOT/bench.py --steps 2 --warmup 1 --cycles 8 --no-cpu-baseline --no-all-flat"
bash tools/prof.sh ${TAG}_kt --kernel-trace --stats -- $B > /dev/null
G1S_ONE_STREAM=1 G1S_D2H_SYNC=1 bash tools/prof.sh ${TAG}_kt1 --kernel-trace --stats -- $B > /dev/null
# PMC passes: the lean driver (frames made on the CPU: no torch kernels under the profiler), two 64-frame launches, one stream
for c in FETCH_SIZE WRITE_SIZE; do
  n=$(echo $c | tr 'A-Z' 'a-z' | sed 's/_size//')
  bash tools/prof.sh ${TAG}_$n --pmc $c -- python $ROOT/tools/diff_pmc.py 2 > /dev/null
done
for d in kt kt1; do echo "== $d"; python tools/kstats.py gpurun_out/${TAG}_$d | tee gpurun_out/${TAG}_kernel_stats_$d.txt; done
for d in fetch write; do echo "== $d"; python tools/pmc_summary.py gpurun_out/${TAG}_$d | grep -v "^==" | tee gpurun_out/${TAG}_pmc_$d.txt; done
find gpurun_out -name "*kernel_trace.csv" -size +8M -delete
find gpurun_out -name "*counter_collection.csv" -size +8M -delete
