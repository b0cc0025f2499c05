#!/bin/bash
out=gpurun_out/${1:-r3h}
mkdir -p $out
export TMPDIR=/tmp
(time python -m pytest tests/test_token_block.py tests/test_dp_gpu.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -30) > $out/tests.log 2>&1
B="--steps 40 --warmup 8 --no-secondary --no-cpu-baseline"
for f in none wgrad convbwd attnbwd tb_fwd tb_bwd convfwd wgrad,convbwd; do
timeout 200 python tools/exp_skip.py $f $B > $out/skip_$f.json 2> $out/skip_$f.err
done
grep -n "passed\|failed" $out/tests.log
for f in $out/skip_*.json; do echo -n "$f  "; python -c "import json,sys; d=json.load(open('$f')); print(d['ms_per_step'], d['value'])"; done
