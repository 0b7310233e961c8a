#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 900 python -m pytest tests -m gpu -x -q -k "records_and_table or goldens or modes_agree" 2>&1 | tail -3
for v in "" "" ""; do
  echo "== 8k10_444 $v"; env WL=8k10_444 DISTINCT=64 $v python tools/ktime.py 4 2>/dev/null | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.readline()); print({k:v for k,v in j['kernels_us'].items() if 'k3s' in k or 'finish' in k or 'moments' in k}, j['sum_us'])"
done
python bench.py --workload 8k10_444 --steps 4 --warmup 2 --no-cpu-baseline --no-all-flat 2>/dev/null | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.readline()); r=j['roofline']; print(round(j['value']), round(j['ms_per_step'],1), round(r['frac'],4), r['kernels_us_per_launch'])"
