#!/bin/bash
# HBM traffic of the wide chain's kernels (FETCH_SIZE, WRITE_SIZE: separate --pmc passes over the lean driver) and what the lists hold
cd ${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r04}
mkdir -p gpurun_out
for c in FETCH_SIZE WRITE_SIZE; do
  n=$(echo $c | tr 'A-Z' 'a-z' | sed 's/_size//')
  bash tools/prof.sh ${TAG}_$n --pmc $c -- python $PWD/tools/diff_pmc.py 2 > /dev/null
  python tools/pmc_summary.py gpurun_out/${TAG}_$n | grep -v "^==" > gpurun_out/${TAG}_pmc_$n.txt
done
find gpurun_out -name "*counter_collection.csv" -size +8M -delete
cat gpurun_out/${TAG}_pmc_fetch.txt gpurun_out/${TAG}_pmc_write.txt
G1S_DBG_ONLY=1 BATCH=64 python tools/ktime.py 1 2>&1 | grep -E "wide list|deferred" | tee gpurun_out/${TAG}_lists.txt
G1S_DBG_ONLY=1 FLAT=1 BATCH=64 python tools/ktime.py 1 2>&1 | grep -E "wide list|deferred" | tee -a gpurun_out/${TAG}_lists.txt
