#!/bin/bash
out=gpurun_out/${1:-r3g}
mkdir -p $out
export TMPDIR=/tmp
(time python -m pytest tests/test_kernels_wgrad.py tests/test_token_block.py tests/test_dp_gpu.py tests/test_full_size_gpu.py -m gpu -q -p no:cacheprovider -x -k "not prior and not cross_attention and not pipe_train" 2>&1 | tail -40) > $out/tests.log 2>&1
for i in 1 2; do
EEGCLIP_WGRAD_PLANES=0 timeout 300 python bench.py --steps 40 --warmup 8 --no-secondary --no-cpu-baseline > $out/bench_oldwgrad_$i.json 2> $out/bench_oldwgrad_$i.err
timeout 300 python bench.py --steps 40 --warmup 8 --no-secondary --no-cpu-baseline > $out/bench_planes_$i.json 2> $out/bench_planes_$i.err
done
timeout 300 python bench.py --breakdown --steps 20 --no-secondary --no-cpu-baseline > $out/breakdown.json 2> $out/breakdown.txt
grep -n "passed\|failed" $out/tests.log
for f in $out/bench_*.json; do echo $f; python -c "import json,sys; d=json.load(open('$f')); print(d['ms_per_step'], d['value'])"; done
head -45 $out/breakdown.txt
