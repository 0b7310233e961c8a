#!/bin/bash
# parity of the stream kernel against fused on the 13 cases of tools/mode_dump.py, kernel times, instruction counters
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
G1S_K3=fused timeout 600 python tools/mode_dump.py gpurun_out/md_fused.pkl > gpurun_out/md_fused.log 2>&1
G1S_K3=stream timeout 600 python tools/mode_dump.py gpurun_out/md_stream.pkl > gpurun_out/md_stream.log 2>&1
tail -2 gpurun_out/md_stream.log
G1S_K3=stream G1S_F_REUSE=0 timeout 600 python tools/mode_dump.py gpurun_out/md_stream_nr.pkl > gpurun_out/md_stream_nr.log 2>&1
python tools/mode_dump.py --cmp gpurun_out/md_fused.pkl gpurun_out/md_stream.pkl > gpurun_out/md_cmp.txt 2>&1
python tools/mode_dump.py --cmp gpurun_out/md_fused.pkl gpurun_out/md_stream_nr.pkl > gpurun_out/md_cmp_nr.txt 2>&1
grep -E "DIFFERENT|differ|Error|error" -A3 gpurun_out/md_cmp.txt gpurun_out/md_cmp_nr.txt | head -40
tail -1 gpurun_out/md_cmp.txt; tail -1 gpurun_out/md_cmp_nr.txt
for m in fused stream; do G1S_K3=$m TAG=$m timeout 120 python tools/ktime.py 3 2>/dev/null | tail -1; done
G1S_K3=stream FLAT=1 TAG=stream_flat timeout 120 python tools/ktime.py 3 2>/dev/null | tail -1
G1S_K3=stream bash tools/prof.sh pmc_a_stream --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY -- python $PWD/tools/diff_pmc.py 2 > /dev/null
python tools/pmc_summary.py gpurun_out/pmc_a_stream | grep -A 9 -E "k3s_fused"
find gpurun_out -name "*.csv" -size +4M -delete
