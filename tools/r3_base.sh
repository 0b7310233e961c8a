#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
for d in 127 0; do
G1S_LIB=$PWD/grav1synth_amd/libg1s_v_dbg.so G1S_S_DBG=$d bash tools/prof.sh pmc_base$d --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY -- python $PWD/tools/diff_pmc.py 2 > /dev/null
echo dbg=$d; python tools/pmc_summary.py gpurun_out/pmc_base$d | grep -A 9 -E "k3s_fused"
done
find gpurun_out -name "*.csv" -size +1M -delete
