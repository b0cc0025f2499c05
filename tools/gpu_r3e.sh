#!/bin/bash
out=gpurun_out/${1:-r3e}
mkdir -p $out
export TMPDIR=/tmp
(time python -m pytest tests/test_token_block.py tests/test_dp_gpu.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -12) > $out/tests.log 2>&1
python tools/bench_token_block.py 50 > $out/time.json 2> $out/time.err
EEGCLIP_TB_DEBUG=1 python tools/bench_token_block.py 50 > $out/time_nostores.json 2> $out/time_nostores.err
for i in 1 2; do
EEGCLIP_SIDE2=0 timeout 300 python bench.py --steps 40 --warmup 8 --no-secondary --no-cpu-baseline > $out/bench_oneside_$i.json 2> $out/bench_oneside_$i.err
timeout 300 python bench.py --steps 40 --warmup 8 --no-secondary --no-cpu-baseline > $out/bench_twoside_$i.json 2> $out/bench_twoside_$i.err
done
EEGCLIP_TOKEN_BLOCK_BWD=0 timeout 300 python bench.py --steps 40 --warmup 8 --no-secondary --no-cpu-baseline > $out/bench_fwdonly.json 2> $out/bench_fwdonly.err
timeout 300 python bench.py --breakdown --steps 20 --no-secondary --no-cpu-baseline > $out/breakdown.json 2> $out/breakdown.txt
tail -5 $out/tests.log; cat $out/time.json $out/time_nostores.json
for f in $out/bench_*.json; do echo $f; python -c "import json,sys; d=json.load(open('$f')); print(d['ms_per_step'], d['value'])"; done
head -14 $out/breakdown.txt
