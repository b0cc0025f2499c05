#!/bin/bash
# round 6, call L: A/B of two builds of the library on one box (tools/with_lib.py): the in-tree build against eeg_image_decode_amd/csrc/libeegclip_hip_prev.so
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
R=$GRAFT_REPO_ROOT
O=gpurun_out/${1:-r6l2}
mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_ops.py tests/test_kernels_head_gemm.py tests/test_model_gpu.py -x -q -m gpu 2>&1 | grep -v "amdgpu.ids\|RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -3 > $O/tests.txt
cat $O/tests.txt
for i in 1 2 3; do
python bench.py --steps 300 --warmup 30 --no-secondary --no-cpu-baseline > $O/bench_new_$i.json 2>> $O/bench.err
python tools/with_lib.py eeg_image_decode_amd/csrc/libeegclip_hip_prev.so bench.py --steps 300 --warmup 30 --no-secondary --no-cpu-baseline > $O/bench_prev_$i.json 2>> $O/bench.err
done
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --output-format csv -d $R/$O/prof -o trace -- python $R/bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline > $R/$O/bench_prof.json 2> $R/$O/prof.err)
f=$(find $O/prof -name "*kernel_trace.csv" | head -1)
python tools/step_timeline.py $f 3 > $O/timeline.txt
rm -rf $O/prof
for f in $O/bench_*_?.json; do python -c "import json,sys; d=json.load(open('$f')); print('$f', d['ms_per_step'], d['config']['host_enqueue_ms_per_step'])"; done
sed -n 1,26p $O/timeline.txt
