#!/bin/bash
out=gpurun_out/${1:-r3l}
mkdir -p $out
export TMPDIR=/tmp
(time python -m pytest tests/test_token_block.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -8) > $out/tests.log 2>&1
python tools/bench_token_block.py 30 256 > $out/time_b256.json 2> $out/time_b256.err
EEGCLIP_TB_DEBUG=2 python tools/bench_token_block.py 20 256 > $out/phases.json 2> $out/phases.err
B="--steps 40 --warmup 8 --no-secondary --no-cpu-baseline"
for i in 1 2 3; do
timeout 200 python bench.py $B > $out/bench_$i.json 2> $out/bench_$i.err
done
timeout 300 python bench.py --breakdown --steps 20 --no-secondary --no-cpu-baseline > $out/breakdown.json 2> $out/breakdown.txt
grep -n "passed\|failed" $out/tests.log
cat $out/time_b256.json
tail -4 $out/phases.err
for f in $out/bench_*.json; do echo -n "$f  "; python -c "import json,sys; d=json.load(open('$f')); print(d['ms_per_step'], d['value'])"; done
head -20 $out/breakdown.txt
