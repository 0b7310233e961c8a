cd ${GRAFT_REPO_ROOT:-/root/repo}
python tools/ktime.py 2 > /dev/null 2>&1
b() { python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-all-flat "$@" 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.readline()); print('$*', round(j['value']), round(j['ms_per_step'],1), j['config']['batch_frames'])"; }
for i in 1 2; do
b --batch 64
b --batch 96
b --batch 128
done
b --workload 1080p8 --batch 128
b --workload 1080p8 --batch 256
b --workload 8k10_444 --batch 16
b --workload 8k10_444 --batch 32
