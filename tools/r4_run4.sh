mkdir -p gpurun_out
for m in 1 8; do
  echo "== merge alone, pool $m, no copy stream"; NO_D2H=1 G1S_FOLD_PROFILE=1 PACE=0 G1S_MERGE_THREADS=$m python tools/host_budget_8ranks.py 6 0 2>&1 | grep -E "ordered merge|frames_per_s\"|cpu_us"
  echo "== merge alone, pool $m, copy stream next to it"; G1S_FOLD_PROFILE=1 PACE=0 G1S_MERGE_THREADS=$m python tools/host_budget_8ranks.py 6 0 2>&1 | grep -E "ordered merge|frames_per_s\"|cpu_us"
done
echo "== tools/bench_fold.py"; python tools/bench_fold.py 2>&1 | tail -2
