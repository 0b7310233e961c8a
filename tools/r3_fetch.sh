#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
for r in 1 0; do
G1S_F_REUSE=$r bash tools/prof.sh fetch_reuse$r --pmc FETCH_SIZE -- python $PWD/tools/diff_pmc.py 2 > /dev/null
echo reuse=$r; python tools/pmc_summary.py gpurun_out/fetch_reuse$r | grep -A1 k3s_fused
done
find gpurun_out -name "*counter_collection.csv" -size +1M -delete
