cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
python tools/ktime.py 2 > /dev/null 2>&1
python -c "
import torch
print('priority range (least, greatest):', torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream,'priority_range') else 'n/a')"
b() { local envs=() args=(); for a in "$@"; do case "$a" in --*) args+=("$a");; *=*) envs+=("$a");; *) args+=("$a");; esac; done
  env "${envs[@]}" python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-all-flat "${args[@]}" 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.readline()); print('$*', round(j['value']), round(j['ms_per_step'],1))"; }
for i in 1 2 3; do
b X=default
b G1S_PRIO=1
b GPU_MAX_HW_QUEUES=8
b GPU_MAX_HW_QUEUES=8 G1S_PRIO=2
done
b X=default --workload 1080p8
b GPU_MAX_HW_QUEUES=8 --workload 1080p8
b X=default --workload 8k10_444
b GPU_MAX_HW_QUEUES=8 --workload 8k10_444
