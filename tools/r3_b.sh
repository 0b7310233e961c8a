#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
python bench.py --steps 4 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.readline()); r=j['roofline']; print(round(j['value']), round(j['ms_per_step'],1), round(r['frac'],4), round(r['frac_all_flat'],4), r['avg_launch_ms'], r['frames_per_launch'], r['kernels_us_per_launch'], r['host_fold_ms_per_frame'])"
G1S_BENCH_SHARE_GPU=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29519 bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.readline()); r=j['roofline']; print(round(j['value']), j['n_gpus'], round(j['ms_per_step'],1), round(r['frac'],4), r['frames_per_launch'])"
