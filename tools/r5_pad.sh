#!/bin/bash
# tools/r5_pad.sh -- the PIPELINED job (bench.py's whole-job value) with the accumulation launches held to three workgroups a CU by an LDS
# request beyond their tile (G1S_W_LDS_PAD / G1S_W_LDS_PAD_C): does the finder chain of the batch after, on the side stream, fill the room?
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
O=gpurun_out/${1:-r05d_lds_pad}.txt; : > $O
run() { echo "## $*" >> $O; env "$@" python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-all-flat 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(round(j['value']), j['ms_per_step'], j['roofline']['frac'])" >> $O; }
for i in 1 2; do
  run G1S_X=0
  run G1S_W_LDS_PAD=8192
  run G1S_W_LDS_PAD=8192 G1S_W_LDS_PAD_C=4096
  run G1S_W_LDS_PAD_C=4096
done
cat $O
