#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 900 python -m pytest tests -m gpu -x -q -k "records_and_table or goldens or modes_agree or large_residuals" 2>&1 | tail -3
for round in 1 2; do
for n in main two one3; do
  lib=$PWD/grav1synth_amd/libg1s_v_$n.so; [ "$n" = main ] && lib=$PWD/grav1synth_amd/libg1s_diff.so
  echo "== $n"; G1S_LIB=$lib WL=8k10_444 DISTINCT=64 python tools/ktime.py 3 2>/dev/null | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.readline()); print({k:v for k,v in j['kernels_us'].items() if 'k3s' in k}, j['sum_us'])"
done
done
