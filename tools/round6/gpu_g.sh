#!/bin/bash
# round 6, call G: the data-parallel step plan -- two gloo ranks sharing the GPU (RCCL refuses duplicate devices): host enqueue per step with / without the plan
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r6g
timeout 900 python -m pytest tests/test_dp_gpu.py -x -q -m gpu 2>&1 | tail -6 > gpurun_out/r6g/tests.txt
cat gpurun_out/r6g/tests.txt
for mode in 1 0; do
EEGCLIP_STEP_PLAN_DP=$mode EEGCLIP_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 2951$mode bench.py --gpus 2 --steps 100 --warmup 20 --no-secondary --no-cpu-baseline > gpurun_out/r6g/bench_gloo2_plan$mode.json 2> gpurun_out/r6g/bench_gloo2_plan$mode.err
python -c "import json; d=json.load(open('gpurun_out/r6g/bench_gloo2_plan$mode.json')); print('plan=$mode', d['ms_per_step'], d['config']['host_enqueue_ms_per_step'], d['config']['submission'][:60])"
done
