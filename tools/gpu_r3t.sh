#!/bin/bash
out=gpurun_out/${1:-r3t}
mkdir -p $out
export TMPDIR=/tmp
python tools/bench_sconv_fwd.py > $out/scf.txt 2>&1
(time python -m pytest tests/test_kernels_ops.py -m gpu -q -p no:cacheprovider -x -k "spatial" 2>&1 | tail -4) > $out/tests.log 2>&1
B="--steps 40 --warmup 8 --no-secondary --no-cpu-baseline"
for i in 1 2 3; do
timeout 200 python bench.py $B > $out/bench_$i.json 2> $out/bench_$i.err
done
timeout 300 python bench.py --breakdown --steps 20 --no-secondary --no-cpu-baseline > $out/breakdown.json 2> $out/breakdown.txt
grep -n "passed\|failed" $out/tests.log
for f in $out/bench_*.json; do echo -n "$f  "; python -c "import json,sys; d=json.load(open('$f')); print(d['ms_per_step'], d['value'], d['config']['host_enqueue_ms_per_step'])"; done
grep "full" $out/scf.txt
grep "conv" $out/breakdown.txt
