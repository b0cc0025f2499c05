#!/bin/bash
# round 5, call B: stand-alone timings + kernel trace of the cstack forward; the model parity tests with it
out=gpurun_out/r5b
mkdir -p $out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python tools/bench_cstack.py $out/cstack_bench.json
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats -d $R/$out/prof -o trace -- python $R/tools/bench_cstack.py > /dev/null 2> $R/$out/prof.err)
f=$(find $out/prof -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp $f $out/cstack_kernel_stats.csv && head -12 $out/cstack_kernel_stats.csv | cut -c1-160
rm -rf $out/prof
(timeout 500 python -m pytest tests/test_model_gpu.py tests/test_full_size_gpu.py -m gpu -x -q -p no:cacheprovider -k "not sdxl and not prior" 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -15) > $out/tests_model.log 2>&1
tail -4 $out/tests_model.log
