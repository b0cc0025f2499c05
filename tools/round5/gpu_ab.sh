#!/bin/bash
# A/B of environment switches on one box: bash tools/round5/gpu_ab.sh "<env A>" "<env B>" [rounds]
out=gpurun_out/r5ab
mkdir -p $out
B="--steps 100 --warmup 10 --no-secondary --no-cpu-baseline"
run() { name=$1; shift; env $@ timeout 200 python bench.py $B > $out/$name.json 2> $out/$name.err; python -c "import json; d=json.load(open('$out/$name.json')); c=d['config']; print('$name', '$*', d['ms_per_step'], c.get('host_enqueue_ms_per_step'), c.get('launches_per_step'))" || tail -5 $out/$name.err; }
for i in $(seq 1 ${3:-3}); do
  run a$i ${1:-X=1}
  run b$i ${2:-X=1}
done
