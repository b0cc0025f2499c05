#!/bin/bash
# round 6, call J: one bench + timeline of the current build (kernel durations of the step, dispatch by dispatch), kernel tests of what changed
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
R=$GRAFT_REPO_ROOT
O=gpurun_out/${1:-r6j}
mkdir -p $O
timeout 900 python -m pytest tests/test_model_gpu.py -x -q -m gpu -k "step_plan or train_step or p0" 2>&1 | grep -v "amdgpu.ids\|RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -4 > $O/tests.txt
cat $O/tests.txt
for i in 1 2 3; do
python bench.py --steps 200 --warmup 30 --no-secondary --no-cpu-baseline > $O/bench_$i.json 2>> $O/bench.err
EEGCLIP_HEAD_FUSED_EPI=1 python bench.py --steps 200 --warmup 30 --no-secondary --no-cpu-baseline > $O/bench_off_$i.json 2>> $O/bench.err
done
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --output-format csv -d $R/$O/prof -o trace -- python $R/bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline > $R/$O/bench_prof.json 2> $R/$O/prof.err)
f=$(find $O/prof -name "*kernel_trace.csv" | head -1)
python tools/step_timeline.py $f 3 > $O/timeline.txt
rm -rf $O/prof
for f in bench_1 bench_off_1 bench_2 bench_off_2 bench_3 bench_off_3; do python -c "import json,sys; d=json.load(open('$O/$f.json')); print('$f', d['ms_per_step'], d['config']['host_enqueue_ms_per_step'], d['config']['launches_per_step'])"; done
cat $O/timeline.txt
