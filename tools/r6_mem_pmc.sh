#!/bin/bash
# memory-side counters of the chain's kernels (round 6): L2 hit rates, EA read latency (LEVEL / RDREQ), DRAM credit stalls, TLB misses.
# lean driver (tools/diff_pmc.py: 64 distinct frame addresses, one stream, two 64-frame launches); one rocprofv3 --pmc run a set
cd ${GRAFT_REPO_ROOT:-/root/repo}
OUTN=${1:-r06_mem_counters}
mkdir -p gpurun_out
run() { tag=$1; shift
  bash tools/prof.sh pmcm_$tag --pmc "$@" -- python $PWD/tools/diff_pmc.py 2 > /dev/null
  python tools/pmc_summary.py gpurun_out/pmcm_$tag | grep -A 9 -E "k3w_pass|k1_moments|k1_certify|k2w_select" > gpurun_out/pmcm_$tag.txt
  find gpurun_out/pmcm_$tag -name "*.csv" -size +4M -delete
}
run a TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_TAG_STALL_sum TCC_BUSY_sum TCC_CYCLE_sum
run b TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCP_LATENCY_sum
run c TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_TA_BUSY_sum TD_TC_STALL_sum TD_TD_BUSY_sum GRBM_GUI_ACTIVE TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum
cat gpurun_out/pmcm_a.txt gpurun_out/pmcm_b.txt gpurun_out/pmcm_c.txt > gpurun_out/$OUTN.txt
cat gpurun_out/$OUTN.txt
