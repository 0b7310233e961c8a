#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 1200 python -m pytest tests -m gpu -x -q -k "large_residuals or records_and_table or modes_agree or goldens or device_latest" 2>&1 | tail -3
for v in 1 0 1 0; do
  echo "== G1S_F_SPLIT444=$v"; G1S_F_SPLIT444=$v WL=8k10_444 DISTINCT=64 python tools/ktime.py 3 2>/dev/null | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.readline()); print({k:v for k,v in j['kernels_us'].items() if 'k3s' in k}, j['sum_us'])"
done
