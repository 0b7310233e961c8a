#!/bin/bash
# A/B: workgroups per launch and frame order of the wide launches (tools/ktime.py, 64 distinct 4K pairs, one stream)
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
OUT=gpurun_out/${1:-r04_ab2}.txt
: > $OUT
run() { echo -n "$* : " >> $OUT; env "$@" python tools/ktime.py 4 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernels_us']; print(d['sum_us'], {n:v for n,v in k.items() if 'k3w_pass' in n or 'moments' in n})" >> $OUT; }
run TAG=default
run G1S_W_WGS=1536 G1S_W_WGS_C=1536
run G1S_W_WGS=3072 G1S_W_WGS_C=2048
run G1S_W_WGS=4096 G1S_W_WGS_C=3072
run G1S_W_WGS=4352 G1S_W_WGS_C=4352
run G1S_W_WGS=6144 G1S_W_WGS_C=4096
run G1S_W_REV=1
run G1S_W_REV=3
run G1S_W_REV=2
run TAG=default
cat $OUT
