#!/bin/bash
# kernel trace (per-dispatch timestamps) of the bench command -> timeline of one steady-state step
out=gpurun_out/r5tl
mkdir -p $out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --output-format csv -d $R/$out/prof -o trace -- python $R/bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline > $R/$out/bench.json 2> $R/$out/prof.err)
f=$(find $out/prof -name "*kernel_trace.csv" | head -1)
python tools/step_timeline.py $f 3 > $out/timeline.txt
rm -rf $out/prof
cat $out/timeline.txt
