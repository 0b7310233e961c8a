#!/bin/bash
# tools/r5_hostcores.sh -- what a rank asks of the host: bench.py's host_cores_busy (rank 0's CPU seconds / wall seconds over the timed steps) and value with
# the per-frame half of the fold on the host (default above 5 cores a rank) and on the device (G1S_LATEST=device)
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
O=gpurun_out/r05_host_cores.txt; : > $O
run() { echo "## $*" >> $O; env "$@" python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-all-flat 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('value', round(j['value']), 'Mpx/s  ms/step', round(j['ms_per_step'],1), ' host_cores_busy', j['host_cores_busy'], ' half:', j['config'].get('per_frame_fold_half'))" >> $O; }
for i in 1 2; do
  run G1S_LATEST=host
  run G1S_LATEST=device
  run G1S_LATEST=device G1S_FOLD_THREADS=1 G1S_MERGE_THREADS=1
done
cat $O
