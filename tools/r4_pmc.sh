#!/bin/bash
# SQ counters of the wide chain's accumulation launches (k3w_pass), lean driver, one stream, two 64-frame launches
cd ${GRAFT_REPO_ROOT:-/root/repo}
OUTN=${1:-r04_sq_counters}
mkdir -p gpurun_out
run() { tag=$1; shift
  bash tools/prof.sh pmcw_$tag --pmc "$@" -- python $PWD/tools/diff_pmc.py 2 > /dev/null
  python tools/pmc_summary.py gpurun_out/pmcw_$tag | grep -A 9 -E "k3w_pass|k1_moments" > gpurun_out/pmcw_$tag.txt
  find gpurun_out/pmcw_$tag -name "*.csv" -size +4M -delete
}
run a SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY
run b SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INSTS_MFMA SQ_WAIT_INST_LDS
run c SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_I8 SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_WAVES
cat gpurun_out/pmcw_a.txt gpurun_out/pmcw_b.txt gpurun_out/pmcw_c.txt > gpurun_out/$OUTN.txt
cat gpurun_out/$OUTN.txt
