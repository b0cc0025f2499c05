#!/bin/bash
# round 3, GPU call B: fused token block -- parity vs the unfused plan on the GPU, A/B bench, per-kernel breakdown
out=gpurun_out/r3b
mkdir -p $out
export TMPDIR=/tmp
(time python -m pytest tests/test_token_block.py tests/test_model_gpu.py tests/test_full_size_gpu.py -m gpu -q -p no:cacheprovider -x -k "not prior and not cross_attention and not pipe_train" 2>&1 | tail -30) > $out/tests.log 2>&1
for i in 1 2; do
EEGCLIP_TOKEN_BLOCK=0 timeout 300 python bench.py --steps 40 --warmup 8 --no-secondary --no-cpu-baseline > $out/bench_unfused_$i.json 2> $out/bench_unfused_$i.err
timeout 300 python bench.py --steps 40 --warmup 8 --no-secondary --no-cpu-baseline > $out/bench_fused_$i.json 2> $out/bench_fused_$i.err
done
timeout 300 python bench.py --breakdown --steps 20 --no-secondary --no-cpu-baseline > $out/breakdown.json 2> $out/breakdown.txt
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o st -- python bench.py --steps 30 --warmup 8 --no-secondary --no-cpu-baseline > $out/stats.log 2>&1 < /dev/null
tail -8 $out/tests.log
for f in $out/bench_*.json; do echo $f; python -c "import json,sys; d=json.load(open('$f')); print(d['ms_per_step'], d['value'])"; done
head -30 $out/breakdown.txt
