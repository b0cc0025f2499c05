#!/bin/bash
# round-3 evidence: the whole -m gpu suite, the default bench line, its rocprofv3 kernel-trace summary, PMC passes (FETCH / WRITE / SQ) of the same command
out=gpurun_out/${1:-r3_final}
mkdir -p $out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
(time python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -25) > $out/tests.log 2>&1
timeout 600 python bench.py > $out/bench.json 2> $out/bench.err
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/stats -o st -- python $R/bench.py --steps 30 --warmup 8 --no-secondary --no-cpu-baseline > $R/$out/stats.log 2>&1 < /dev/null
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $R/$out/pmc$i -o p -- python $R/bench.py --steps 8 --warmup 3 --no-secondary --no-cpu-baseline > $R/$out/pmc$i.log 2>&1 < /dev/null
done
cd $R
python tools/pmc_summary.py $out/pmc_summary.json "$out/pmc*/**/*counter_collection.csv"
find $out -name "*.csv" -size +2M -delete
grep -n "passed\|failed" $out/tests.log; head -c 600 $out/bench.json; echo; head -25 $out/stats/st_kernel_stats.csv | cut -c1-150
