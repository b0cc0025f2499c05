#!/bin/bash
# round 6, call W: per-kernel trace of the SDXL-shaped sampling loop
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r6w}
mkdir -p $O
cd /tmp
timeout 250 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o v -- python $R/tools/round6/bench_sdxl_loop.py > $O/prof.log 2>&1 < /dev/null
f=$(find $O/prof -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp $f $O/kernel_stats.csv && head -16 $O/kernel_stats.csv | cut -c1-170
find $O/prof -name "*.csv" -size +1M -delete
tail -2 $O/prof.log | cut -c1-200
