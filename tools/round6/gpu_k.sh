#!/bin/bash
# round 6, call K: stop-event forks vs recorded forks -- bench alternation and two profiled timelines of each
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
R=$GRAFT_REPO_ROOT
O=gpurun_out/${1:-r6k2}
mkdir -p $O
for i in 1 2 3 4; do
python bench.py --steps 300 --warmup 30 --no-secondary --no-cpu-baseline > $O/bench_stop_$i.json 2>> $O/bench.err
EEGCLIP_FORK_STOP_EVENT=0 python bench.py --steps 300 --warmup 30 --no-secondary --no-cpu-baseline > $O/bench_record_$i.json 2>> $O/bench.err
done
for m in 1 0; do for i in 1 2; do
(cd /tmp && EEGCLIP_FORK_STOP_EVENT=$m timeout 200 rocprofv3 --kernel-trace --output-format csv -d $R/$O/prof -o trace -- python $R/bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline > $R/$O/bench_prof.json 2> $R/$O/prof.err)
f=$(find $O/prof -name "*kernel_trace.csv" | head -1)
python tools/step_timeline.py $f 3 > $O/timeline_m${m}_$i.txt
python tools/step_timeline.py $f 9 > $O/timeline_m${m}_${i}b.txt
rm -rf $O/prof
done; done
for f in $O/bench_*_?.json; do python -c "import json,sys; d=json.load(open('$f')); print('$f', d['ms_per_step'], d['config']['host_enqueue_ms_per_step'])"; done
head -1 $O/timeline_m*.txt
