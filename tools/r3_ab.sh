#!/bin/bash
# A/B of library builds on one box: tools/r3_ab.sh name1 name2 ... (libg1s_v_NAME.so; "main" = the product library), 3 rounds
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
: > gpurun_out/ab.txt
for round in 1 2 3; do
for n in "$@"; do
  lib=$PWD/grav1synth_amd/libg1s_v_$n.so; [ "$n" = main ] && lib=$PWD/grav1synth_amd/libg1s_diff.so
  G1S_LIB=$lib G1S_K3=stream TAG=$n timeout 120 python tools/ktime.py 3 2>/dev/null | tail -1 >> gpurun_out/ab.txt
done
done
python - <<'PY'
import json, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for line in open("gpurun_out/ab.txt"):
    j = json.loads(line)
    for k, v in j["kernels_us"].items():
        acc[j["tag"]][k].append(v)
    acc[j["tag"]]["sum"].append(j["sum_us"])
for tag, d in acc.items():
    print(tag, {k: min(v) for k, v in d.items() if k.startswith("k3s") or k == "sum"})
PY
