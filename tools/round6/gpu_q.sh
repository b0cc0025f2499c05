#!/bin/bash
# round 6, call Q: conflict-free conv-stack LDS reads (k-slot permutation, E planes) against the previous build on one box; the 256 x 256 InfoNCE tile
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
R=$GRAFT_REPO_ROOT
O=gpurun_out/${1:-r6q}
mkdir -p $O
F='amdgpu.ids\|RCCL\|HIP version\|ROCm version\|Hostname\|Librccl'
timeout 900 python -m pytest tests/test_kernels_cstack.py tests/test_kernels_infonce_fused.py tests/test_token_block.py -x -q -m gpu 2>&1 | grep -v "$F" | tail -3 > $O/tests.txt
cat $O/tests.txt
timeout 300 python tools/round6/bench_infonce_tiles.py $O/infonce_tiles.json 2>&1 | grep -v "$F" | tail -5
timeout 200 python tools/bench_cstack.py > $O/cstack_new.txt 2>&1; tail -12 $O/cstack_new.txt
timeout 200 python tools/with_lib.py eeg_image_decode_amd/csrc/libeegclip_hip_prev.so tools/bench_cstack.py > $O/cstack_prev.txt 2>&1; tail -12 $O/cstack_prev.txt
for i in 1 2 3; do
python bench.py --steps 300 --warmup 30 --no-secondary --no-cpu-baseline > $O/bench_new_$i.json 2>> $O/bench.err
python tools/with_lib.py eeg_image_decode_amd/csrc/libeegclip_hip_prev.so bench.py --steps 300 --warmup 30 --no-secondary --no-cpu-baseline > $O/bench_prev_$i.json 2>> $O/bench.err
done
for f in $O/bench_*_?.json; do python -c "import json,sys; d=json.load(open('$f')); print('$f', d['ms_per_step'], d['config']['host_enqueue_ms_per_step'])"; done
bash tools/gpu_pmc_cmd.sh ${1:-r6q}/pmc cstack bench_cstack.py 2>&1 | tail -12
