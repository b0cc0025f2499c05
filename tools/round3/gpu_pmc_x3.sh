#!/bin/bash
# PMC counters of the split-bf16 GEMM on two shapes of the step (separate passes per counter group; kernel-trace only, no other trace domains)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-pmc_x3}
mkdir -p $O
cd /tmp
rocprofv3 -L > $O/counters.txt 2>&1
for shape in "16384 744 250 nt 1" "744 250 16384 tn 32"; do
  tag=$(echo $shape | tr ' ' '_')
  i=0
  for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT" \
             "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" \
             "SQ_INSTS_MFMA SQ_ACTIVE_INST_MISC SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_WAVES" \
             "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr" \
             "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    i=$((i+1))
    timeout 120 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/${tag}_g$i -o p -- python $R/tools/x3_one.py $shape 1 12 > $O/${tag}_g$i.log 2>&1
  done
done
python $R/tools/pmc_summary.py $O/summary_qkv.json "$O/16384_744_250_nt_1_g*/**/*counter_collection.csv" > /dev/null
python $R/tools/pmc_summary.py $O/summary_wqkv.json "$O/744_250_16384_tn_32_g*/**/*counter_collection.csv" > /dev/null
grep -c . $O/counters.txt
