cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
python tools/ktime.py 2 > /dev/null 2>&1
b() { local envs=() args=(); for a in "$@"; do case "$a" in --*) args+=("$a");; *=*) envs+=("$a");; *) args+=("$a");; esac; done
  env "${envs[@]}" python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-all-flat "${args[@]}" 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.readline()); print('$*', round(j['value']), round(j['ms_per_step'],1))"; }
for i in 1 2 3; do
b X=0
b G1S_F_SERIAL=1
b G1S_F_SERIAL=1 G1S_SIDE2=1
done
b X=0 --workload 1080p8
b G1S_F_SERIAL=1 --workload 1080p8
b X=0 --workload 8k10_444
b G1S_F_SERIAL=1 --workload 8k10_444
b X=0 --flat
b G1S_F_SERIAL=1 --flat
