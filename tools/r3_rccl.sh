#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 900 python -m pytest tests -m gpu -x -q -k "rccl or two_ranks or sharded or native or latest_only" 2>&1 | tail -3
python tools/rccl_round_cost.py > gpurun_out/rccl_round.txt 2>&1; grep -E "Mpx/s|identical|Error|Traceback" -A3 gpurun_out/rccl_round.txt | head
G1S_ROUNDS_LOCAL=1 python tools/rccl_round_cost.py > gpurun_out/rccl_round2.txt 2>&1; grep -E "Mpx/s|identical|Error|Traceback" -A3 gpurun_out/rccl_round2.txt | head
G1S_BENCH_SHARE_GPU=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29519 bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline --no-all-flat 2>/dev/null | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.readline()); print(round(j['value']), j['n_gpus'], round(j['ms_per_step'],1), j['config'].get('exchange_ms_per_round'))"
