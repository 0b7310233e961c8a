cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
python tools/ktime.py 2 > /dev/null 2>&1
t() { env "$@" python tools/ktime.py 3 2>/dev/null | tail -1 | python -c "
import json,sys; j=json.loads(sys.stdin.readline()); k=j['kernels_us']; print('$*', 'sum', j['sum_us'], 'tail', k.get('k3w_tail'), 'luma', [v for n,v in k.items() if 'pass<0' in n], 'moments', [v for n,v in k.items() if 'moments' in n])"; }
for i in 1 2 3; do
t WL=8k10_444 BATCH=16 DISTINCT=16
t WL=8k10_444 BATCH=16 DISTINCT=16 GPU_MAX_HW_QUEUES=8
t WL=8k10_444 BATCH=16 DISTINCT=16 G1S_D2H_SYNC=1
t WL=4k10
t WL=4k10 GPU_MAX_HW_QUEUES=8
done
