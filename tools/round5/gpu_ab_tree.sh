#!/bin/bash
# A/B of the working tree against a built checkout of another commit on one box: bash tools/round5/gpu_ab_tree.sh <other tree> [rounds]
# (tools/round5/alt/ is git-ignored: `git worktree add tools/round5/alt/head <commit>` + `python -m eeg_image_decode_amd.build` in it)
out=gpurun_out/r5abt
mkdir -p $out
B="--steps 100 --warmup 10 --no-secondary --no-cpu-baseline"
show() { python -c "import json; d=json.load(open('$out/$1.json')); c=d['config']; print('$1', d['ms_per_step'], c.get('host_enqueue_ms_per_step'), c.get('launches_per_step'))" || tail -5 $out/$1.err; }
for i in $(seq 1 ${2:-3}); do
  timeout 200 python bench.py $B > $out/new$i.json 2> $out/new$i.err; show new$i
  (cd $1 && timeout 200 python bench.py $B) > $out/old$i.json 2> $out/old$i.err; show old$i
done
