mkdir -p gpurun_out
bash tools/r4_quick.sh
WL=1080p8 BATCH=128 DISTINCT=128 python tools/ktime.py 3 2>/dev/null | tail -1
WL=8k10_444 BATCH=16 DISTINCT=16 python tools/ktime.py 3 2>/dev/null | tail -1
