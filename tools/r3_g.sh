#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 900 python -m pytest tests -m gpu -x -q -k "records_and_table or modes_agree or ceiling or tuning or goldens or two_ranks" 2>&1 | tail -3
for r in 1 2; do
for v in "" "G1S_F_WGS_L=2048" "G1S_F_WGS_L=6144"; do
  echo "== $v"; env $v python tools/ktime.py 3 2>/dev/null | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.readline()); print({k:v for k,v in j['kernels_us'].items() if 'k3s' in k or 'finish' in k}, j['sum_us'])"
done
done
for v in "" "G1S_F_WGS_L=2048"; do
  echo "== bench $v"; env $v python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-all-flat 2>/dev/null | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.readline()); r=j['roofline']; print(round(j['value']), round(j['ms_per_step'],1), round(r['frac'],4))"
done
