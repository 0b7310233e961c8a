#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
for b in 64 96 128 64 128; do
  echo "== --batch $b"; python bench.py --batch $b --steps 5 --warmup 2 --no-cpu-baseline --no-all-flat 2>/dev/null | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.readline()); r=j['roofline']; print(round(j['value']), round(j['ms_per_step'],1), round(r['frac'],4), r['frames_per_launch'], {k:v for k,v in r['kernels_us_per_launch'].items() if 'k3s' in k or 'moments' in k})"
done
