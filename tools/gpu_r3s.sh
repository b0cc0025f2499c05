#!/bin/bash
out=gpurun_out/${1:-r3s}
mkdir -p $out
export TMPDIR=/tmp
python tools/bench_sconv_fwd.py > $out/scf.txt 2>&1
(time python -m pytest tests/test_kernels_ops.py tests/test_model_gpu.py -m gpu -q -p no:cacheprovider -x -n 4 -k "spatial or train or eval or fixture or accuracy" 2>&1 | tail -6) > $out/tests.log 2>&1
B="--steps 40 --warmup 8 --no-secondary --no-cpu-baseline"
for i in 1 2 3; do
timeout 200 python bench.py $B > $out/bench_$i.json 2> $out/bench_$i.err
done
grep -n "passed\|failed" $out/tests.log
for f in $out/bench_*.json; do echo -n "$f  "; python -c "import json,sys; d=json.load(open('$f')); print(d['ms_per_step'], d['value'], d['config']['host_enqueue_ms_per_step'])"; done
tail -12 $out/scf.txt
