#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
for m in fused stream; do
G1S_K3=$m timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_$m.json 2> gpurun_out/bench_$m.err
done
G1S_K3=stream timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --flat > gpurun_out/bench_stream_flat.json 2> gpurun_out/bench_stream_flat.err
G1S_K3=stream G1S_F_REUSE=0 timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_stream_noreuse.json 2> gpurun_out/bench_stream_noreuse.err
python - <<'PY'
import json
for m in ("fused","stream","stream_flat","stream_noreuse"):
    try:
        j=json.loads(open(f"gpurun_out/bench_{m}.json").read().strip().splitlines()[-1])
        print(m, round(j["value"]), round(j["ms_per_step"],2), round(j["roofline"]["frac"],4), json.dumps(j["roofline"].get("kernels_us_per_launch")))
    except Exception as e:
        print(m, "failed", e); print(open(f"gpurun_out/bench_{m}.err").read()[-1500:])
PY
