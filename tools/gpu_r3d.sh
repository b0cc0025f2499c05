#!/bin/bash
out=gpurun_out/${1:-r3d}
mkdir -p $out
export TMPDIR=/tmp
(time python -m pytest tests/test_token_block.py tests/test_model_gpu.py tests/test_full_size_gpu.py tests/test_dp_gpu.py -m gpu -q -p no:cacheprovider -x -k "not prior and not cross_attention and not pipe_train" 2>&1 | tail -30) > $out/tests.log 2>&1
for i in 1 2; do
EEGCLIP_TOKEN_BLOCK_BWD=0 timeout 300 python bench.py --steps 40 --warmup 8 --no-secondary --no-cpu-baseline > $out/bench_fwdonly_$i.json 2> $out/bench_fwdonly_$i.err
timeout 300 python bench.py --steps 40 --warmup 8 --no-secondary --no-cpu-baseline > $out/bench_fused_$i.json 2> $out/bench_fused_$i.err
done
timeout 300 python bench.py --breakdown --steps 20 --no-secondary --no-cpu-baseline > $out/breakdown.json 2> $out/breakdown.txt
tail -8 $out/tests.log
for f in $out/bench_*.json; do echo $f; python -c "import json,sys; d=json.load(open('$f')); print(d['ms_per_step'], d['value'])"; done
head -40 $out/breakdown.txt
