#!/bin/bash
# round-6 evidence run on the GPU box:  bash tools/final_profiles.sh [tag]
#   1. PMC passes of the bench command (separate --pmc runs, kernel-trace only) -> per-kernel summary -> per-family HBM-traffic / MFMA-busy table
#      (written into profiles/ of this copy so that the bench runs below pick it up, and into gpurun_out/ to travel back)
#   2. rocprofv3 --kernel-trace --stats of the bench command
#   3. the full bench line (secondary workloads + CPU baseline)
#   4. PMC of the InfoNCE tile kernel at N = 2048, both arithmetic modes
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-r6_final}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_MFMA SQ_ACTIVE_INST_ANY SQ_WAVES" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/g$i -o p -- python $R/bench.py --steps 12 --warmup 3 --no-secondary --no-cpu-baseline > $O/g$i.log 2>&1
done
python $R/tools/pmc_summary.py $O/pmc_step.json "$O/g*/**/*counter_collection.csv"
python $R/tools/pmc_traffic_table.py $O/pmc_step.json 256 > $O/pmc_hbm_traffic.json
cp $O/pmc_hbm_traffic.json $R/profiles/r6_pmc_hbm_traffic.json
find $O -name "*.csv" -delete
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o st -- python $R/bench.py --steps 30 --warmup 8 --no-secondary --no-cpu-baseline > $O/stats.log 2>&1 < /dev/null
cp $O/stats/*kernel_stats.csv $O/kernel_stats.csv 2>/dev/null || find $O/stats -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
python $R/tools/step_timeline.py $(find $O/stats -name "*kernel_trace.csv" | head -1) 3 > $O/step_timeline.txt 2>/dev/null      # one steady-state step, dispatch by dispatch
find $O/stats -name "*.csv" -size +1M -delete
cd $R
if [ -z "$SKIP_BENCH" ]; then      # (tools/round6/gpu_final.sh takes the bench line FIRST, on the fresh box: after six minutes of tests and profiling passes the same
                                    #  command reads 5-7 % slower -- 0.695-0.70 against 0.653 ms per step on boxes that A/B calls measured at 0.65-0.66)
timeout 400 python bench.py --steps 50 --warmup 10 > $O/bench.json 2> $O/bench.err < /dev/null
cut -c1-400 $O/bench.json
fi
cd /tmp
j=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_MFMA SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVES" "FETCH_SIZE" "WRITE_SIZE"; do
  j=$((j+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/i$j -o p -- python $R/tools/bench_infonce_fused.py > $O/i$j.log 2>&1
done
python $R/tools/pmc_summary.py $O/pmc_infonce.json "$O/i*/**/*counter_collection.csv"
find $O -name "*.csv" -size +1M -delete
python - <<PY
import json
d = json.load(open("$O/pmc_infonce.json"))
for k, v in d.items():
    if "infonce" in k: print(k, {a: (round(b, 4) if isinstance(b, float) else b) for a, b in v.items()})
PY
