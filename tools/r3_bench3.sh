#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_$tag.json 2> gpurun_out/bench_$tag.err; }
run s2_0 G1S_K3=stream G1S_SIDE2=0
run s2_1 G1S_K3=stream G1S_SIDE2=1
run s2_1b G1S_K3=stream G1S_SIDE2=1
run s2_0b G1S_K3=stream G1S_SIDE2=0
run serial G1S_K3=stream G1S_SIDE2=1 G1S_F_SERIAL=1
python - <<'PY'
import json
for m in ("s2_0","s2_1","s2_1b","s2_0b","serial"):
    try:
        j=json.loads(open(f"gpurun_out/bench_{m}.json").read().strip().splitlines()[-1])
        print(m, round(j["value"]), round(j["ms_per_step"],2), j["step_ms"])
    except Exception as e:
        print(m, "failed", e); print(open(f"gpurun_out/bench_{m}.err").read()[-1500:])
PY
