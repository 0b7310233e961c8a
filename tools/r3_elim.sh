#!/bin/bash
# elimination runs on the stream kernel (the instrumented build libg1s_v_dbg.so): which part of an iteration is the time
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
: > gpurun_out/elim.txt
for d in 0 1 2 4 8 16 32 64 3 15 24 31 127 0; do
  G1S_LIB=$PWD/grav1synth_amd/libg1s_v_dbg.so G1S_K3=stream G1S_S_DBG=$d TAG=dbg$d timeout 120 python tools/ktime.py 3 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1])
k = j['kernels_us']
print(j['tag'], {n: v for n, v in k.items() if n.startswith('k3s')})" >> gpurun_out/elim.txt
done
cat gpurun_out/elim.txt
