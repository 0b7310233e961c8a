#!/bin/bash
# SQ counters of the accumulation kernels (stream and fused), lean driver, one stream, two 64-frame launches
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
(cd /tmp && rocprofv3 -L > ${GRAFT_REPO_ROOT:-/root/repo}/gpurun_out/counters_list.txt 2>&1)
grep -o "SQ_[A-Z_0-9]*" gpurun_out/counters_list.txt | sort -u | tr '\n' ' ' > gpurun_out/sq_names.txt
run() { # tag mode counters...
  tag=$1; mode=$2; shift 2
  G1S_K3=$mode bash tools/prof.sh pmc_${tag}_$mode --pmc "$@" -- python $PWD/tools/diff_pmc.py 2 > /dev/null
  python tools/pmc_summary.py gpurun_out/pmc_${tag}_$mode | grep -A 9 -E "k3s_fused|k3f_fused" > gpurun_out/pmc_${tag}_$mode.txt
  find gpurun_out/pmc_${tag}_$mode -name "*.csv" -size +4M -delete
}
for mode in stream fused; do
  run a $mode SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY
  run b $mode SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INSTS_MFMA SQ_WAIT_INST_LDS
done
run c stream SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_I8 SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_WAVES
cat gpurun_out/pmc_*_*.txt
