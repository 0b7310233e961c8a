mkdir -p gpurun_out
python tools/merge_rate.py 400
G1S_MERGE_POOL=8 python tools/merge_rate.py 400
G1S_MERGE_POOL=1 python tools/merge_rate.py 40
