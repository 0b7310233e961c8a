#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
one() { python bench.py "$@" --steps 5 --warmup 2 --no-cpu-baseline --no-all-flat 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.readline()); r=j['roofline']
print(round(j['value']), 'Mpx/s  frac', round(r['frac'],4), 'whole-job', round(j['hbm_roofline_frac_whole_job'],4), ' frames/launch', r['frames_per_launch'], r['kernels_us_per_launch'])"; }
for b in 64 128 256; do echo "== 1080p8 --batch $b"; one --workload 1080p8 --batch $b; done
for b in 128 256; do echo "== 4k10 --batch $b"; one --batch $b; done
