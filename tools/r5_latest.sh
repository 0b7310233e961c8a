#!/bin/bash
# tools/r5_latest.sh -- the device per-frame half (G1S_LATEST=device, k4_latest) against the host half on one box:
# parity tests, the kernel's phases (timers build), its launch time, and the bench line both ways.
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
O=gpurun_out/r05_latest.txt; : > $O
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "device_latest" 2>&1 | tail -2 >> $O
echo "## phases (timers build)" >> $O
G1S_LATEST=device G1S_LIB=$PWD/grav1synth_amd/libg1s_v_lt.so python tools/ktime.py 1 2>&1 | grep "k4_latest phases" | tail -3 >> $O
echo "## ktime, device half" >> $O
G1S_LATEST=device python tools/ktime.py 3 2>/dev/null | tail -1 >> $O
for i in 1 2; do
  for w in host device; do
    echo "## bench G1S_LATEST=$w" >> $O
    G1S_LATEST=$w python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-all-flat 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(round(j['value']), j['ms_per_step'], j['config'].get('per_frame_fold_half'))" >> $O
  done
done
cat $O
