cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
O=gpurun_out/r05_mom.txt; : > $O
run() { echo "## $*" >> $O; env "$@" python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-all-flat 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(round(j['value']), j['ms_per_step'])" >> $O; }
for i in 1 2 3; do
  run X=default
  run G1S_MOM_STREAM=1
  run G1S_MOM_STREAM=2
  run G1S_SIDE2=1
done
cat $O
