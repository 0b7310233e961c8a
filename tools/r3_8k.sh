#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
for m in fused stream; do WL=8k10_444 DISTINCT=16 BATCH=16 G1S_K3=$m TAG=$m timeout 400 python tools/ktime.py 3 2>/dev/null | tail -1; done
for m in fused stream; do WL=1080p8 DISTINCT=64 BATCH=128 G1S_K3=$m TAG=$m timeout 400 python tools/ktime.py 3 2>/dev/null | tail -1; done
