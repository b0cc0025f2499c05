#!/bin/bash
# round 6, call B: the head on the K-parallel plane GEMM -- parity on the hardware, the step, its dispatch timeline
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6b
timeout 900 python -m pytest tests/test_kernels_head_gemm.py tests/test_model_gpu.py tests/test_full_size_gpu.py tests/test_kernels_infonce_fused.py tests/test_kernels_ops.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r6b/tests.txt
python bench.py --steps 200 --warmup 30 --no-secondary --no-cpu-baseline > gpurun_out/r6b/bench.json 2> gpurun_out/r6b/bench.err
EEGCLIP_HEAD_GEMM=0 python bench.py --steps 200 --warmup 30 --no-secondary --no-cpu-baseline > gpurun_out/r6b/bench_head0.json 2> gpurun_out/r6b/bench_head0.err
python bench.py --steps 200 --warmup 30 --no-secondary --no-cpu-baseline > gpurun_out/r6b/bench2.json 2>> gpurun_out/r6b/bench.err
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r6b/prof -o trace -- python $R/bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline > $R/gpurun_out/r6b/bench_prof.json 2> $R/gpurun_out/r6b/prof.err)
f=$(find gpurun_out/r6b/prof -name "*kernel_trace.csv" | head -1)
python tools/step_timeline.py $f 3 > gpurun_out/r6b/timeline.txt
rm -rf gpurun_out/r6b/prof
cat gpurun_out/r6b/tests.txt
for f in bench bench_head0 bench2; do python -c "import json,sys; d=json.load(open('gpurun_out/r6b/$f.json')); print('$f', d['ms_per_step'], d['config']['host_enqueue_ms_per_step'], d['config']['launches_per_step'])"; done
cat gpurun_out/r6b/timeline.txt
