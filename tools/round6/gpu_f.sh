#!/bin/bash
# round 6, call F: the step plan for the reconstruction objective and the joint-subject model
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r6f
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_full_size_gpu.py tests/test_kernels_ops.py -x -q -m gpu 2>&1 | tail -12 > gpurun_out/r6f/tests.txt
cat gpurun_out/r6f/tests.txt
timeout 300 python -c "
import json, bench
print(json.dumps(bench._sec_joint()), flush=True)
import os
os.environ['EEGCLIP_STEP_PLAN'] = '0'
print('no plan', json.dumps(bench._sec_joint()), flush=True)
" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6f/joint.txt
python bench.py --steps 200 --warmup 30 --no-secondary --no-cpu-baseline > gpurun_out/r6f/bench.json 2> gpurun_out/r6f/bench.err
python -c "import json; d=json.load(open('gpurun_out/r6f/bench.json')); print('bench', d['ms_per_step'], d['config']['host_enqueue_ms_per_step'], d['config']['launches_per_step'])"
