#!/bin/bash
# round 6, call D: riders (no second-stream work at step start), lean BN2 backward, one fork for the head's side launches, row-per-workgroup head LayerNorm
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6d
timeout 900 python -m pytest tests/test_kernels_head_gemm.py tests/test_model_gpu.py tests/test_full_size_gpu.py -x -q -m gpu -k "infonce_small or step_plan or train_step" 2>&1 | tail -25 > gpurun_out/r6d/tests.txt
for i in 1 2; do
python bench.py --steps 200 --warmup 30 --no-secondary --no-cpu-baseline > gpurun_out/r6d/bench$i.json 2>> gpurun_out/r6d/bench.err
EEGCLIP_INFONCE_SMALL=0 python bench.py --steps 200 --warmup 30 --no-secondary --no-cpu-baseline > gpurun_out/r6d/bench_ifsmall0_$i.json 2>> gpurun_out/r6d/bench.err
EEGCLIP_HEAD_GEMM=0 python bench.py --steps 200 --warmup 30 --no-secondary --no-cpu-baseline > gpurun_out/r6d/bench_head0_$i.json 2>> gpurun_out/r6d/bench.err
done
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r6d/prof -o trace -- python $R/bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline > $R/gpurun_out/r6d/bench_prof.json 2> $R/gpurun_out/r6d/prof.err)
f=$(find gpurun_out/r6d/prof -name "*kernel_trace.csv" | head -1)
python tools/step_timeline.py $f 3 > gpurun_out/r6d/timeline.txt
rm -rf gpurun_out/r6d/prof
cat gpurun_out/r6d/tests.txt
for f in bench1 bench_ifsmall0_1 bench_head0_1 bench2 bench_ifsmall0_2 bench_head0_2; do python -c "import json,sys; d=json.load(open('gpurun_out/r6d/$f.json')); print('$f', d['ms_per_step'], d['config']['host_enqueue_ms_per_step'], d['config']['launches_per_step'])"; done
cat gpurun_out/r6d/timeline.txt
