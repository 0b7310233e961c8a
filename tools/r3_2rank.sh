#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
G1S_BENCH_SHARE_GPU=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 3 --warmup 1 --cycles 16 --no-cpu-baseline > gpurun_out/bench_2ranks.json 2> gpurun_out/bench_2ranks.err
tail -c 900 gpurun_out/bench_2ranks.json; tail -5 gpurun_out/bench_2ranks.err
timeout 300 python tools/bench_fold.py > gpurun_out/fold_budget.json 2>/dev/null; cat gpurun_out/fold_budget.json
