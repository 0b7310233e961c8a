#!/bin/bash
# the ordered merge on the GPU box's host cores, by pool size, with the stage timers
mkdir -p gpurun_out
{
for t in 1 32 256; do
  echo "== G1S_FOLD_THREADS=$t"
  G1S_FOLD_PROFILE=1 G1S_FOLD_THREADS=$t python tools/bench_fold.py 3840x2160 merge-only 2>&1 | tail -3
done
for m in 4 8 16 32; do
  echo "== G1S_MERGE_POOL=$m"
  G1S_FOLD_PROFILE=1 G1S_FOLD_THREADS=32 G1S_MERGE_POOL=$m python tools/bench_fold.py 3840x2160 merge-only 2>&1 | tail -3
done
} > gpurun_out/r3_fold.txt 2>&1
cat gpurun_out/r3_fold.txt
python -m pytest tests -m gpu -x -q -k "two_ranks or scene or sharded or records_and_table" 2>&1 | tail -3
