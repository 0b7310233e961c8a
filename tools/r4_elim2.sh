#!/bin/bash
# elimination runs, second set (chroma launch's parts): G1S_W_DBG bits through libg1s_v_wdbg.so
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
OUT=gpurun_out/${1:-r04_elim2}.txt
: > $OUT
for d in 0 64 128 192 4 68 196 16 2 8 32 0; do
  echo -n "dbg$d " >> $OUT
  G1S_LIB=$PWD/grav1synth_amd/libg1s_v_wdbg.so G1S_W_DBG=$d python tools/ktime.py 3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read())['kernels_us']; print({k:v for k,v in d.items() if 'k3w_pass' in k})" >> $OUT
done
cat $OUT
