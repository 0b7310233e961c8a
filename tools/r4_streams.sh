cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
python tools/ktime.py 2 > /dev/null 2>&1
timeout 900 python -m pytest tests/test_gpu_selfcheck.py -m gpu -x -q 2>&1 | tail -2
b() { local envs=() args=(); for a in "$@"; do case "$a" in --*) args+=("$a");; *=*) envs+=("$a");; *) args+=("$a");; esac; done
  env "${envs[@]}" python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-all-flat "${args[@]}" 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.readline()); print('$*', round(j['value']), round(j['ms_per_step'],1))"; }
for i in 1 2 3; do
b X=default
b G1S_W_ASIDE=1
b G1S_SIDE2=1
b G1S_ONE_STREAM=1
done
for w in 1080p8 1080p8_lag2_luma 8k10_444; do
b X=default --workload $w
b G1S_W_ASIDE=1 --workload $w
b X=default --workload $w
b G1S_W_ASIDE=1 --workload $w
done
