#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 1500 python -m pytest tests -m gpu -x -q -k "device_latest" 2>&1 | tail -5
G1S_LIB=$PWD/grav1synth_amd/libg1s_v_lt.so python tools/latest_time.py 3840x2160 10 64 2>&1 | grep -E "phases" | head -3
python tools/latest_time.py 3840x2160 10 64 2>&1 | tail -5
