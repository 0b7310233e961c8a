#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
run() { python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-all-flat 2>&1 | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.readline()); r=j['roofline']
print(round(j['value']), 'Mpx/s', round(j['ms_per_step'],1), 'ms/step  host fold ms/frame', round(r.get('host_fold_ms_per_frame',0),5), {k:v for k,v in r['kernels_us_per_launch'].items() if 'k4' in k or 'k3s' in k})"; }
echo "== host"; G1S_LATEST=host run
echo "== device"; G1S_LATEST=device run
G1S_LATEST=device timeout 600 python -m pytest tests -m gpu -x -q -k "device_latest or records_and_table or two_ranks or sharded" 2>&1 | tail -3
