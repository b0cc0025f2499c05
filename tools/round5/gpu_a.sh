#!/bin/bash
# round 5, call A: the recomputing conv-stack forward (csrc/cstack.hip) on the hardware: parity, step time with / without it, kernel trace
out=gpurun_out/r5a
mkdir -p $out
export TMPDIR=/tmp
(timeout 300 python -m pytest tests/test_kernels_cstack.py -m gpu -q -p no:cacheprovider 2>&1 | tail -15) > $out/tests_cstack.log 2>&1
tail -3 $out/tests_cstack.log
B="--steps 40 --warmup 8 --no-secondary --no-cpu-baseline"
run() { name=$1; shift; env "$@" timeout 200 python bench.py $B > $out/$name.json 2> $out/$name.err; python -c "import json; d=json.load(open('$out/$name.json')); print('$name', d['ms_per_step'], d['config'].get('host_enqueue_ms_per_step'))" || tail -5 $out/$name.err; }
run cstack X=1
run old EEGCLIP_CSTACK=0
run cstack2 X=1
run old2 EEGCLIP_CSTACK=0
(timeout 400 python -m pytest tests/test_model_gpu.py tests/test_full_size_gpu.py -m gpu -x -q -p no:cacheprovider -k "not sdxl and not prior" 2>&1 | tail -5) > $out/tests_model.log 2>&1
tail -3 $out/tests_model.log
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$out/prof -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 5 --no-secondary --no-cpu-baseline > $GRAFT_REPO_ROOT/$out/bench_prof.json 2> $GRAFT_REPO_ROOT/$out/bench_prof.err
cd $GRAFT_REPO_ROOT
f=$(find $out/prof -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp $f $out/kernel_stats.csv && head -30 $out/kernel_stats.csv | cut -c1-200
find $out/prof -name "*.db" -delete; find $out/prof -name "*kernel_trace.csv" -size +20M -delete
