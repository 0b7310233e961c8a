cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
python tools/ktime.py 2 > /dev/null 2>&1
b() { local envs=() args=(); for a in "$@"; do case "$a" in --*) args+=("$a");; *=*) envs+=("$a");; *) args+=("$a");; esac; done
  env "${envs[@]}" python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-all-flat "${args[@]}" 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.readline()); print('$*', round(j['value']), round(j['ms_per_step'],1))"; }
for i in 1 2 3; do
b X=default8
b GPU_MAX_HW_QUEUES=4
done
for w in 1080p8 8k10_444; do
b X=default8 --workload $w
b GPU_MAX_HW_QUEUES=4 --workload $w
done
b X=default8 --flat
b GPU_MAX_HW_QUEUES=4 --flat
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
