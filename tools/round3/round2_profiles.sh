#!/bin/bash
# round-2 evidence: the bench line, the rocprofv3 kernel-trace summary of the same command, PMC counters of the fused InfoNCE kernels
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r2_final}
mkdir -p $O
cd $R
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err < /dev/null
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o st -- python $R/bench.py --no-secondary --no-cpu-baseline > $O/stats.json 2> $O/stats.err < /dev/null
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA SQ_ACTIVE_INST_VMEM SQ_LDS_IDX_ACTIVE SQ_WAVES" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 120 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/if_g$i -o p -- python $R/tools/bench_infonce_fused.py > $O/if_g$i.log 2>&1
done
python $R/tools/pmc_summary.py $O/pmc_infonce_fused.json "$O/if_g*/**/*counter_collection.csv"
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete
find $O -name "*.csv" | head; cut -c1-300 $O/bench.json
