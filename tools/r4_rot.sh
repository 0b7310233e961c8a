cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
V=$PWD/grav1synth_amd/libg1s_v_rot.so
python tools/ktime.py 2 > /dev/null 2>&1
G1S_LIB=$V timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "test_records_and_table_match_oracle or large_residuals or ragged or halo_dwords" 2>&1 | tail -2
for i in 1 2; do
python tools/ktime.py 4 2>/dev/null | tail -1
G1S_LIB=$V TAG=rot python tools/ktime.py 4 2>/dev/null | tail -1
done
WL=1080p8 BATCH=128 DISTINCT=128 python tools/ktime.py 3 2>/dev/null | tail -1
G1S_LIB=$V TAG=rot WL=1080p8 BATCH=128 DISTINCT=128 python tools/ktime.py 3 2>/dev/null | tail -1
WL=8k10_444 BATCH=16 DISTINCT=16 python tools/ktime.py 3 2>/dev/null | tail -1
G1S_LIB=$V TAG=rot WL=8k10_444 BATCH=16 DISTINCT=16 python tools/ktime.py 3 2>/dev/null | tail -1
