#!/bin/bash
out=gpurun_out/${1:-r3c}
mkdir -p $out
export TMPDIR=/tmp
(time python -m pytest tests/test_token_block.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -15) > $out/tests.log 2>&1
python tools/bench_token_block.py 50 > $out/time.json 2> $out/time.err
timeout 300 python bench.py --steps 40 --warmup 8 --no-secondary --no-cpu-baseline > $out/bench_fused.json 2> $out/bench_fused.err
cd /tmp
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT --output-format csv -d $GRAFT_REPO_ROOT/$out/g1 -o p -- python $GRAFT_REPO_ROOT/tools/bench_token_block.py 6 > $GRAFT_REPO_ROOT/$out/g1.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/pmc_summary.py $out/summary.json "$out/g*/**/*counter_collection.csv"
find $out -name "*.csv" -size +2M -delete
tail -4 $out/tests.log; cat $out/time.json; python -c "import json; d=json.load(open('$out/bench_fused.json')); print(d['ms_per_step'], d['value'])"
python - <<PY
import json
d=json.load(open("$out/summary.json"))
for k,v in d.items():
    if "token_block_fwd" in k: print(k, {c:int(x) for c,x in v.items()})
PY
