#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
for m in fused stream; do echo == $m 8k444; WL=8k10_444 DISTINCT=4 BATCH=4 G1S_DBG_ONLY=1 G1S_K3=$m timeout 400 python tools/ktime.py 1 2>&1 | grep -E "deferred" | head -3; done
for m in fused stream; do echo == $m 4k; DISTINCT=8 BATCH=8 G1S_DBG_ONLY=1 G1S_K3=$m timeout 400 python tools/ktime.py 1 2>&1 | grep -E "deferred" | head -3; done
