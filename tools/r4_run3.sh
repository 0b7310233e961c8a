mkdir -p gpurun_out
for m in 1 8; do
  G1S_MERGE_THREADS=$m python tools/host_budget_8ranks.py 8 0 > gpurun_out/r04_host_merge_only_m$m.json 2>&1
done
PACE=0 G1S_MERGE_THREADS=8 python tools/host_budget_8ranks.py 8 0 > gpurun_out/r04_host_merge_only_flat_m8.json 2>&1
HALF_THREADS=10 G1S_MERGE_THREADS=6 python tools/host_budget_8ranks.py 12 1 > gpurun_out/r04_host_rank0_h10_m6.json 2>&1
HALF_THREADS=11 G1S_MERGE_THREADS=4 python tools/host_budget_8ranks.py 12 1 > gpurun_out/r04_host_rank0_h11_m4.json 2>&1
PACE=0 HALF_THREADS=11 G1S_MERGE_THREADS=4 python tools/host_budget_8ranks.py 12 1 > gpurun_out/r04_host_rank0_flat.json 2>&1
python tools/host_budget_8ranks.py 12 8 > gpurun_out/r04_host_node_paced.json 2>&1
PACE=0 python tools/host_budget_8ranks.py 12 8 > gpurun_out/r04_host_node_flat.json 2>&1
