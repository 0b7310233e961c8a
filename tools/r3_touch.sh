#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
for r in 1 2; do
for d in 0 128; do G1S_LIB=$PWD/grav1synth_amd/libg1s_v_dbg.so G1S_S_DBG=$d TAG=dbg$d timeout 200 python tools/ktime.py 3 2>/dev/null | tail -1 | cut -c1-200; done
TAG=main timeout 200 python tools/ktime.py 3 2>/dev/null | tail -1 | cut -c1-200
done
