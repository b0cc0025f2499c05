#!/bin/bash
# round 6, call T: the session's final build against the build of the session's first commit, the evidence run's own bench command, alternated on one box
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-r6t}
mkdir -p $O
for i in 1 2 3; do
python bench.py --steps 50 --warmup 10 --no-secondary --no-cpu-baseline > $O/bench_new_$i.json 2>> $O/bench.err
python tools/with_lib.py eeg_image_decode_amd/csrc/libeegclip_hip_prev.so bench.py --steps 50 --warmup 10 --no-secondary --no-cpu-baseline > $O/bench_prev_$i.json 2>> $O/bench.err
done
python bench.py --steps 300 --warmup 30 --no-secondary --no-cpu-baseline > $O/bench_new_300.json 2>> $O/bench.err
python tools/with_lib.py eeg_image_decode_amd/csrc/libeegclip_hip_prev.so bench.py --steps 300 --warmup 30 --no-secondary --no-cpu-baseline > $O/bench_prev_300.json 2>> $O/bench.err
for f in $O/bench_*.json; do python -c "import json,sys; d=json.load(open('$f')); print('$f', d['ms_per_step'], d['config']['host_enqueue_ms_per_step'], d['roofline']['frac'])"; done
