#!/bin/bash
# round 6, call I: fork events bound to the producing kernel's completion (no marker packet on the main queue) against recorded fork events
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
R=$GRAFT_REPO_ROOT
O=gpurun_out/r6i
mkdir -p $O
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_full_size_gpu.py tests/test_dp_gpu.py tests/test_kernels_head_gemm.py -x -q -m gpu 2>&1 | tail -5 > $O/tests.txt
cat $O/tests.txt
for i in 1 2 3; do
python bench.py --steps 200 --warmup 30 --no-secondary --no-cpu-baseline > $O/bench_stop_$i.json 2>> $O/bench.err
EEGCLIP_FORK_STOP_EVENT=0 python bench.py --steps 200 --warmup 30 --no-secondary --no-cpu-baseline > $O/bench_record_$i.json 2>> $O/bench.err
done
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --output-format csv -d $R/$O/prof -o trace -- python $R/bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline > $R/$O/bench_prof.json 2> $R/$O/prof.err)
f=$(find $O/prof -name "*kernel_trace.csv" | head -1)
python tools/step_timeline.py $f 3 > $O/timeline.txt
rm -rf $O/prof
for f in bench_stop_1 bench_record_1 bench_stop_2 bench_record_2 bench_stop_3 bench_record_3; do python -c "import json,sys; d=json.load(open('$O/$f.json')); print('$f', d['ms_per_step'], d['config']['host_enqueue_ms_per_step'], d['config']['launches_per_step'])"; done
cat $O/timeline.txt
tail -5 $O/bench.err
