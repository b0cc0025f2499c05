#!/bin/bash
# A/B/C of environment switches on one box: bash tools/round5/gpu_ab3.sh rounds "<env A>" "<env B>" ["<env C>" ...]
out=gpurun_out/r5ab3
mkdir -p $out
B="--steps 100 --warmup 10 --no-secondary --no-cpu-baseline"
n=$1; shift
for i in $(seq 1 $n); do
  j=0
  for e in "$@"; do
    j=$((j+1))
    env $e timeout 200 python bench.py $B > $out/v${j}_$i.json 2> $out/v${j}_$i.err
    python -c "import json; d=json.load(open('$out/v${j}_$i.json')); c=d['config']; print('v$j', '$e', d['ms_per_step'], c.get('host_enqueue_ms_per_step'), c.get('launches_per_step'))" || tail -5 $out/v${j}_$i.err
  done
done
