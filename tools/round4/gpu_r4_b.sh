#!/bin/bash
out=gpurun_out/r4b
mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels_wgrad.py tests/test_token_block.py -m gpu -x -q > $out/pytest.log 2>&1
echo "pytest rc=$?" | tee -a $out/pytest.log
tail -3 $out/pytest.log
B="--steps 40 --warmup 8 --no-secondary --no-cpu-baseline"
run() { name=$1; shift; env "$@" timeout 150 python bench.py $B > $out/$name.json 2> $out/$name.err; python -c "import json; d=json.load(open('$out/$name.json')); r=d['roofline']; print('$name', d['ms_per_step'], r['frac'], r['single_stream']['frac'], {k: v['ms_per_step_single_stream'] for k, v in list(r['families'].items())[:3]})"; }
run base X=1
run var1 EEGCLIP_WGRAD_VARIANT=1
run sk24 EEGCLIP_WGRAD_SK=24
run sk32 EEGCLIP_WGRAD_SK=32
timeout 200 python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline --breakdown > $out/breakdown.json 2> $out/breakdown.txt
grep "^#" $out/breakdown.txt | grep "wgrad\|token_block\|sum of"
