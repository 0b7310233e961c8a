mkdir -p gpurun_out
python tools/ktime.py 2 > /dev/null 2>&1
for cfg in "16 8" "12 4" "10 6"; do set -- $cfg
  HALF_THREADS=$1 G1S_MERGE_THREADS=$2 G1S_MERGE_POOL=$2 python tools/host_budget_8ranks.py 12 1 > gpurun_out/r04_host_rank0_h$1_m$2.json 2>&1
done
PACE=0 python tools/host_budget_8ranks.py 12 1 > gpurun_out/r04_host_rank0_flat.json 2>&1
python bench.py > gpurun_out/r04f_bench.json 2>gpurun_out/r04f_bench.err
G1S_LATEST=device python bench.py > gpurun_out/r04f_bench_devlatest.json 2>gpurun_out/r04f_bench_devlatest.err
tail -c 600 gpurun_out/r04f_bench.json; echo; tail -c 600 gpurun_out/r04f_bench_devlatest.json
echo; WL=1080p8 BATCH=128 DISTINCT=128 python tools/ktime.py 3 2>/dev/null | tail -1 | tee gpurun_out/r04f_1080p8.txt
WL=8k10_444 BATCH=16 DISTINCT=16 python tools/ktime.py 3 2>/dev/null | tail -1 | tee gpurun_out/r04f_8k444.txt
