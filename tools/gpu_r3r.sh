#!/bin/bash
out=gpurun_out/${1:-r3r}
mkdir -p $out
export TMPDIR=/tmp
(time python -m pytest tests/test_dp_gpu.py tests/test_model_gpu.py tests/test_token_block.py -m gpu -q -p no:cacheprovider -x -n 4 2>&1 | tail -8) > $out/tests.log 2>&1
python tools/host_phase_probe.py > $out/host_phases.txt 2>&1
B="--steps 40 --warmup 8 --no-secondary --no-cpu-baseline"
for i in 1 2 3; do
timeout 200 python bench.py $B > $out/bench_$i.json 2> $out/bench_$i.err
done
timeout 200 python bench.py --batch 64 $B > $out/bench_b64.json 2> $out/bench_b64.err
grep -n "passed\|failed" $out/tests.log
for f in $out/bench_*.json; do echo -n "$f  "; python -c "import json,sys; d=json.load(open('$f')); print(d['ms_per_step'], d['value'], d['config']['host_enqueue_ms_per_step'])"; done
tail -12 $out/host_phases.txt
