cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
while read -r line; do
  i=$((i+1))
  timeout 200 rocprofv3 --output-format csv -d $R/gpurun_out/pm_$i -o p --pmc $line -- python $R/tools/diff_pmc.py 1 > $R/gpurun_out/pm_$i.log 2>&1 < /dev/null
  (cd $R; python tools/pmc_summary.py gpurun_out/pm_$i k3f_fused) >> $R/gpurun_out/pmc_mem.txt
  (cd $R; python tools/pmc_summary.py gpurun_out/pm_$i k1_moments) >> $R/gpurun_out/pmc_mem.txt
done <<'LIST'
TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum
TCP_UTCL1_SERIALIZATION_STALL_sum TCP_UTCL1_THRASHING_STALL_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_STALL_MULTI_MISS_sum
TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum
TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum
TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_GATE_EN1_sum
SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY
LIST
find $R/gpurun_out -name "*counter_collection.csv" -size +2M -delete
