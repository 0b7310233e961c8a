#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
for w in 2048 4096 6144 8192 12288; do
  echo "== G1S_F_WGS=$w"; G1S_F_WGS=$w python tools/ktime.py 3 2>/dev/null | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.readline()); print({k:v for k,v in j['kernels_us'].items() if 'k3s' in k}, j['sum_us'])"
done
for w in 2048 4096 8192; do
  echo "== bench G1S_F_WGS=$w"; G1S_F_WGS=$w python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-all-flat 2>/dev/null | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.readline()); r=j['roofline']; print(round(j['value']), round(j['ms_per_step'],1), round(r['frac'],4), {k:v for k,v in r['kernels_us_per_launch'].items() if 'k3s' in k})"
done
