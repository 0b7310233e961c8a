#!/bin/bash
# elimination runs of the wide kernels: parts left out (G1S_W_DBG bits, libg1s_v_wdbg.so = -DG1S_W_DBG_BUILD), us per 64-frame launch
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
OUT=gpurun_out/${1:-r04_elim}.txt
: > $OUT
for d in 0 1 2 4 8 16 32 3 7 15 24 31 63 0; do
  echo -n "dbg$d " >> $OUT
  G1S_LIB=$PWD/grav1synth_amd/libg1s_v_wdbg.so G1S_W_DBG=$d python tools/ktime.py 3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read())['kernels_us']; print({k:v for k,v in d.items() if 'k3w_pass' in k})" >> $OUT
done
cat $OUT
