#!/bin/bash
# round 5, call C: the recomputing conv-stack backward on the hardware: parity, step time, stand-alone timings, kernel trace + PMC of the cstack kernels
out=gpurun_out/r5c
mkdir -p $out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
F='RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids'
(timeout 300 python -m pytest tests/test_kernels_cstack.py -m gpu -q -p no:cacheprovider 2>&1 | grep -v "$F" | tail -15) > $out/tests_cstack.log 2>&1
tail -3 $out/tests_cstack.log
python tools/bench_cstack.py $out/cstack_bench.json
B="--steps 40 --warmup 8 --no-secondary --no-cpu-baseline"
run() { name=$1; shift; env "$@" timeout 200 python bench.py $B > $out/$name.json 2> $out/$name.err; python -c "import json; d=json.load(open('$out/$name.json')); print('$name', d['ms_per_step'], d['config'].get('host_enqueue_ms_per_step'))" || tail -5 $out/$name.err; }
run new X=1
run fwdonly EEGCLIP_CSTACK_BWD=0
run old EEGCLIP_CSTACK=0
run new2 X=1
(timeout 600 python -m pytest tests/test_model_gpu.py tests/test_full_size_gpu.py tests/test_dp_gpu.py tests/test_token_block.py -m gpu -x -q -p no:cacheprovider -k "not sdxl and not prior" 2>&1 | grep -v "$F" | tail -15) > $out/tests_model.log 2>&1
tail -4 $out/tests_model.log
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/prof -o trace -- python $R/bench.py --steps 30 --warmup 5 --no-secondary --no-cpu-baseline > $R/$out/bench_prof.json 2> $R/$out/prof.err)
f=$(find $out/prof -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp $f $out/step_kernel_stats.csv && head -40 $out/step_kernel_stats.csv | cut -c1-150
rm -rf $out/prof
bash tools/gpu_pmc_cmd.sh r5c/pmc cstack bench_cstack.py 2>&1 | tail -80 > $out/pmc_tail.log
