#!/bin/bash
# round 6, call U: per-kernel trace of one VAE decode
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r6u}
mkdir -p $O
cd /tmp
timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o v -- python $R/tools/round6/bench_vae_decode.py > $O/prof.log 2>&1 < /dev/null
f=$(find $O/prof -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp $f $O/kernel_stats.csv && head -16 $O/kernel_stats.csv | cut -c1-160
find $O/prof -name "*.csv" -size +1M -delete
tail -2 $O/prof.log | cut -c1-300
