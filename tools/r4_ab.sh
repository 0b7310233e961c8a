#!/bin/bash
# A/B driver: tools/ktime.py under several environments; each line = one JSON of per-kernel HIP-event times (64 distinct 4K pairs)
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
OUT=gpurun_out/${1:-r04_ab}.txt
: > $OUT
run() { echo "## $*" >> $OUT; env "$@" python tools/ktime.py 3 2>/dev/null | tail -1 >> $OUT; }
run TAG=wide
run TAG=stream G1S_K3=stream
run TAG=wide_b32 BATCH=32
run TAG=wide_wgs1024 G1S_W_WGS=1024 G1S_W_WGS_C=1024
run TAG=wide_wgs4096 G1S_W_WGS=4096 G1S_W_WGS_C=2048
run TAG=wide_wgs8192 G1S_W_WGS=8192 G1S_W_WGS_C=4096
run TAG=wide_flat FLAT=1
run TAG=stream_flat FLAT=1 G1S_K3=stream
cat $OUT
