#!/bin/bash
# tools/r5_other_workloads.sh -- the other BASELINE workloads, the all-flat variant, the fallback chain and the two-rank
# shared-GPU line on the current build (parity-test cases, not bench lines) -> gpurun_out/${OUT:-r06}_other_workloads.txt
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
python tools/ktime.py 2 > /dev/null 2>&1
one() { python bench.py "$@" --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.readline()); r=j['roofline']
print(round(j['value']), 'Mpx/s  whole-job frac', round(j['hbm_roofline_frac_whole_job'],4), ' roofline.frac', round(r['frac'],4), ' frames/launch', r['frames_per_launch'], ' flat', round(j['config']['flat_fraction'],3), r['kernels_us_per_launch'])"; }
{
echo "bench.py --workload W --steps 5 --warmup 2 --no-cpu-baseline: value, whole-job and kernel roofline fractions, per-kernel us per launch (HIP events, one stream)"
for w in 1080p8_lag2_luma 1080p8 8k10_444; do echo "== $w"; one --workload $w; done
echo "== 1080p8 --batch 64 (the default is the engine's choice: 128)"; one --workload 1080p8 --batch 64
echo "== 1080p8_lag2_luma --batch 64"; one --workload 1080p8_lag2_luma --batch 64
echo "== 8k10_444 --batch 64 (default: 32)"; one --workload 8k10_444 --batch 64
echo "== 4k10 --flat"; one --flat
echo "== 4k10 G1S_K3=stream (round 3's chain, the fallback)"; G1S_K3=stream one
echo "== 4k10 --batch 32"; one --batch 32
echo "== 4k10 G1S_LATEST=device (the per-frame half on the device)"; G1S_LATEST=device one
echo "== 4k10, two ranks sharing the one GPU (G1S_BENCH_SHARE_GPU=1, torch.distributed.run --nproc-per-node 2, bench.py --gpus 2)"
G1S_BENCH_SHARE_GPU=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 5 --warmup 2 2>/dev/null | tail -1 | tee gpurun_out/${OUT:-r06}_bench_2ranks_shared_gpu.json | python -c "
import json,sys
j=json.loads(sys.stdin.readline()); print(round(j['value']), 'Mpx/s', j['n_gpus'], 'ranks', round(j['ms_per_step'],2), 'ms/step', j['config']['parallelism'])"
} > gpurun_out/${OUT:-r06}_other_workloads.txt 2>&1
cat gpurun_out/${OUT:-r06}_other_workloads.txt
