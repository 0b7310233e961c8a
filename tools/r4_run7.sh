mkdir -p gpurun_out
for cfg in "8 6" "7 4"; do set -- $cfg
HALF_THREADS=$1 G1S_MERGE_THREADS=$2 python tools/host_budget_8ranks.py 12 1 > gpurun_out/r04i_host_rank0_h$1_m$2.json 2>&1
done
python tools/host_budget_8ranks.py 12 8 > gpurun_out/r04i_host_node_paced.json 2>&1
PACE=0 python tools/host_budget_8ranks.py 12 8 > gpurun_out/r04i_host_node_flat.json 2>&1
G1S_MERGE_THREADS=8 python tools/host_budget_8ranks.py 8 0 > gpurun_out/r04i_host_merge_only_m8.json 2>&1
G1S_MERGE_THREADS=1 PACE=0 python tools/host_budget_8ranks.py 8 0 > gpurun_out/r04i_host_merge_only_flat_m1.json 2>&1
G1S_MERGE_THREADS=8 PACE=0 python tools/host_budget_8ranks.py 8 0 > gpurun_out/r04i_host_merge_only_flat_m8.json 2>&1
python tools/host_cores.py > gpurun_out/r04i_host_cores.txt 2>&1
python tools/half_scaling.py >> gpurun_out/r04i_host_cores.txt 2>&1
