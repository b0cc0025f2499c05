#!/bin/bash
# PMC passes over the diffusion-prior training bench (tools/bench_prior_train.py): per-kernel counters of the plane GEMMs / weight gradients / stage tails
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-pmc_prior}
mkdir -p $O
cd /tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAVES" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/g$i -o p -- python $R/tools/bench_prior_train.py > $O/g$i.log 2>&1
done
python $R/tools/pmc_summary.py $O/pmc_prior.json "$O/g*/**/*counter_collection.csv"
find $O -name "*.csv" -delete
