mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r04h_gputests.txt 2>&1
tail -3 gpurun_out/r04h_gputests.txt
python tools/half_scaling.py > gpurun_out/r04h_half_scaling.txt 2>&1; cat gpurun_out/r04h_half_scaling.txt
HALF_THREADS=8 G1S_MERGE_THREADS=6 python tools/host_budget_8ranks.py 12 1 > gpurun_out/r04h_host_rank0_h8_m6.json 2>&1
python tools/host_budget_8ranks.py 12 8 > gpurun_out/r04h_host_node_paced.json 2>&1
python bench.py > gpurun_out/r04h_bench.json 2>gpurun_out/r04h_bench.err
tail -c 1500 gpurun_out/r04h_bench.json
