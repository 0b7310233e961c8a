#!/bin/bash
# memory-side counters of the chain's kernels (round 6): L2 hit rates, EA read latency (LEVEL / RDREQ), DRAM credit stalls, TLB misses.
# lean driver (tools/diff_pmc.py: 64 distinct frame addresses, one stream, two 64-frame launches); one rocprofv3 --pmc run a set of
# at most FOUR counters of a block (gfx950: 4 TCC slots a pass; a set the hardware cannot take makes rocprofv3 abort -- and hang)
cd ${GRAFT_REPO_ROOT:-/root/repo}
ROOT=$PWD
OUTN=${1:-r06_mem_counters}
mkdir -p gpurun_out
run() { tag=$1; shift
  OUT=$ROOT/gpurun_out/pmcm_$tag; mkdir -p $OUT
  (cd /tmp && TMPDIR=/tmp timeout 150 rocprofv3 --output-format csv -d $OUT -o p --pmc "$@" -- python $ROOT/tools/diff_pmc.py 2 > $OUT/run.log 2>&1 < /dev/null)
  echo "set $tag rc=$?"
  python tools/pmc_summary.py gpurun_out/pmcm_$tag | grep -A 5 -E "k3w_pass|k1_moments|k1_certify" > gpurun_out/pmcm_$tag.txt
  find gpurun_out/pmcm_$tag -name "*.csv" -size +4M -delete
}
run a TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum
run b TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_TAG_STALL_sum TCC_BUSY_sum TCC_CYCLE_sum
run c TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_UTCL1_TRANSLATION_MISS_sum
run d TCP_UTCL1_TRANSLATION_HIT_sum TCP_TCR_TCP_STALL_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
cat gpurun_out/pmcm_a.txt gpurun_out/pmcm_b.txt gpurun_out/pmcm_c.txt gpurun_out/pmcm_d.txt > gpurun_out/$OUTN.txt
cat gpurun_out/$OUTN.txt
