run() { python bench.py --batch ${B:-64} --no-cpu-baseline --steps 5 --warmup 2 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.readline()); r=j['roofline']; k=r['kernels_us_per_launch']; b=${B:-64}
print('$1', round(j['value']), 'ms/step', round(j['ms_per_step'],2), 'frac', round(r['frac'],4), 'luma/frame', round(k['k3f_fused<16, 16, 2, 0, 0>']/b,2), 'chroma/frame', round(k['k3f_fused<16, 16, 2, 1, 0>']/b,2), 'mom/frame', round(k['k1_moments<2>']/b,2))
"; }
for d in 0 1 2 3 4 6; do G1S_F_DEPHASE=$d run dephase=$d; done > gpurun_out/dephase.txt 2>&1
