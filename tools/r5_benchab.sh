#!/bin/bash
# tools/r5_benchab.sh OUTNAME ROUNDS "ENV=.. G1S_LIB=v_x" ... -- the PIPELINED job (bench.py --steps 6) under several environments / variant
# libraries, the specs taken in turn ROUNDS times on one box (the box's own spread run to run is several per cent: read the rounds side by side)
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
O=gpurun_out/${1:-r05_benchab}.txt; R=${2:-2}; shift; shift; : > $O
run() {
  echo "## $*" >> $O
  local args=()
  for a in "$@"; do
    case "$a" in G1S_LIB=v_*) a="G1S_LIB=$PWD/grav1synth_amd/libg1s_${a#G1S_LIB=}.so";; esac
    args+=("$a")
  done
  env "${args[@]}" python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-all-flat ${BENCH_ARGS} 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(round(j['value']), round(j['ms_per_step'],1), round(j['roofline']['frac'],4))" >> $O
}
for i in $(seq $R); do for spec in "$@"; do run $spec; done; done
cat $O
