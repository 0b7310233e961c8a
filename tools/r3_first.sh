#!/bin/bash
# first GPU contact of the stream kernel: records of 13 cases under fused and stream, compared; then the bench line of both
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
G1S_K3=fused timeout 600 python tools/mode_dump.py gpurun_out/md_fused.pkl > gpurun_out/md_fused.log 2>&1
G1S_K3=stream timeout 600 python tools/mode_dump.py gpurun_out/md_stream.pkl > gpurun_out/md_stream.log 2>&1
tail -3 gpurun_out/md_stream.log
python tools/mode_dump.py --cmp gpurun_out/md_fused.pkl gpurun_out/md_stream.pkl > gpurun_out/md_cmp.txt 2>&1
cat gpurun_out/md_cmp.txt
G1S_K3=fused timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_fused.json 2> gpurun_out/bench_fused.err
G1S_K3=stream timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_stream.json 2> gpurun_out/bench_stream.err
python - <<'PY'
import json
for m in ("fused","stream"):
    try:
        j=json.loads(open(f"gpurun_out/bench_{m}.json").read().strip().splitlines()[-1])
        print(m, j["value"], j["ms_per_step"], j["roofline"]["frac"], json.dumps(j["roofline"].get("kernels_us_per_launch")))
    except Exception as e:
        print(m, "failed", e); print(open(f"gpurun_out/bench_{m}.err").read()[-1500:])
PY
