cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
O=gpurun_out/r05_slots.txt; : > $O
run() { echo "## $*" >> $O; env "$@" python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-all-flat 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(round(j['value']), j['ms_per_step'], j['config'].get('per_frame_fold_half'), j['roofline']['kernels_us_per_launch'].get('k4_latest'))" >> $O; }
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "device_latest" 2>&1 | tail -1 >> $O
G1S_LATEST=device G1S_LIB=$PWD/grav1synth_amd/libg1s_v_lt.so python tools/ktime.py 1 2>&1 | grep "k4_latest phases" | tail -2 >> $O
for i in 1 2; do
  run G1S_LATEST=host
  run G1S_LATEST=device
  run G1S_LATEST=device G1S_LATEST_PRIO=1
done
cat $O
