#!/bin/bash
# round 6, call V: per-kernel trace of the diffusion prior's training step (batch 1024)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r6v}
mkdir -p $O
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o v -- python $R/tools/bench_prior_train.py > $O/prof.log 2>&1 < /dev/null
f=$(find $O/prof -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp $f $O/kernel_stats.csv && head -24 $O/kernel_stats.csv | cut -c1-150
find $O/prof -name "*.csv" -size +1M -delete
tail -3 $O/prof.log | cut -c1-300
