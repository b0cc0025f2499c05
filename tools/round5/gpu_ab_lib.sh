#!/bin/bash
# A/B of the in-tree library against another build on one box: bash tools/round5/gpu_ab_lib.sh <alt .so> [rounds]
out=gpurun_out/r5abl
mkdir -p $out
B="--steps 100 --warmup 10 --no-secondary --no-cpu-baseline"
show() { python -c "import json; d=json.load(open('$out/$1.json')); c=d['config']; print('$1', d['ms_per_step'], c.get('host_enqueue_ms_per_step'), c.get('launches_per_step'))" || tail -5 $out/$1.err; }
for i in $(seq 1 ${2:-3}); do
  timeout 200 python bench.py $B > $out/new$i.json 2> $out/new$i.err; show new$i
  timeout 200 python tools/with_lib.py $1 bench.py $B > $out/alt$i.json 2> $out/alt$i.err; show alt$i
done
echo new; timeout 100 python tools/bench_token_block.py 60 2>&1 | tail -1
echo alt; timeout 100 python tools/with_lib.py $1 tools/bench_token_block.py 60 2>&1 | tail -1
echo new; timeout 100 python tools/bench_token_block.py 60 2>&1 | tail -1
echo alt; timeout 100 python tools/with_lib.py $1 tools/bench_token_block.py 60 2>&1 | tail -1
