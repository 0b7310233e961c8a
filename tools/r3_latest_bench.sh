#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
for w in host device host device; do
  echo "== G1S_LATEST=$w"
  G1S_LATEST=$w python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-all-flat 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.readline()); r=j['roofline']
print(round(j['value']), 'Mpx/s', round(j['ms_per_step'],1), 'ms/step  host fold ms/frame', round(r.get('host_fold_ms_per_frame',0),5), {k:v for k,v in r['kernels_us_per_launch'].items() if 'k4' in k or 'k3s' in k})"
done
