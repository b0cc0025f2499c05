#!/bin/bash
# one gpurun call of round 2: everything that needs the GPU in one go (tests, GEMM sweep, bench, breakdown)
tag=${1:-r2a}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
(time python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -80) > $out/tests.log 2>&1
python tools/bench_gemm_x3.py --out $out/x3_sweep.json > $out/x3_sweep.log 2>&1
python bench.py --steps 30 --warmup 5 > $out/bench.json 2> $out/bench.err
EEGCLIP_GEMM_PRECISION=f32 python bench.py --steps 30 --warmup 5 --no-secondary --no-cpu-baseline > $out/bench_f32.json 2> $out/bench_f32.err
python bench.py --breakdown --steps 20 --no-secondary --no-cpu-baseline > $out/breakdown.json 2> $out/breakdown.txt
tail -5 $out/tests.log; head -c 2500 $out/bench.json
