#!/bin/bash
# the pipelined job's kernel timeline (rocprofv3 --kernel-trace): what overlaps what
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
for s2 in 0 1; do
G1S_K3=stream G1S_SIDE2=$s2 bash tools/prof.sh tl_$s2 --kernel-trace -- python $PWD/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null
python tools/timeline.py gpurun_out/tl_$s2 60 400 > gpurun_out/timeline_side2_$s2.txt
find gpurun_out/tl_$s2 -name "*.csv" -size +1M -delete
done
cat gpurun_out/timeline_side2_0.txt; echo =====; cat gpurun_out/timeline_side2_1.txt
