#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 1200 python -m pytest tests -m gpu -x -q -k "certified_flat_finder or records_and_table or goldens or estimate" 2>&1 | tail -3
for r in 1 2 3; do python tools/ktime.py 3 2>/dev/null | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.readline()); print({k:v for k,v in j['kernels_us'].items() if 'k3s' in k or 'moments' in k}, j['sum_us'])"; done
WL=1080p8 python tools/ktime.py 3 2>/dev/null | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.readline()); print({k:v for k,v in j['kernels_us'].items() if 'moments' in k}, j['sum_us'])"
