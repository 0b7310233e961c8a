#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
for d in 16 64 16 64; do DISTINCT=$d TAG=distinct$d timeout 200 python tools/ktime.py 3 2>/dev/null | tail -1; done
