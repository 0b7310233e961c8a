#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
for p in 0 12000 20000 30000 45000; do
  echo "== G1S_F_LDS_PAD=$p"; G1S_F_LDS_PAD=$p python tools/ktime.py 3 2>/dev/null | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.readline()); print({k:v for k,v in j['kernels_us'].items() if 'k3s' in k}, j['sum_us'])"
done
