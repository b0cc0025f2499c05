#!/bin/bash
# PMC counters (separate passes per counter group) of a stand-alone tool:  gpu_pmc_cmd.sh <out dir under gpurun_out> <kernel-name filter> <tool.py> [tool args...]
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$1; FILT=$2; shift 2
mkdir -p $O
cd /tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" \
           "SQ_INSTS_MFMA SQ_ACTIVE_INST_MISC SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_WAVES" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/g$i -o p -- python $R/tools/"$@" > $O/g$i.log 2>&1
done
python $R/tools/pmc_summary.py $O/summary.json "$O/g*/**/*counter_collection.csv"
find $O -name "*.csv" -size +2M -delete
python - <<PY
import json
d=json.load(open("$O/summary.json"))
for k,v in d.items():
    if "$FILT" in k: print(k, json.dumps(v, indent=0))
PY
