#!/bin/bash
out=gpurun_out/r4d
mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels_ops.py tests/test_token_block.py tests/test_model_gpu.py tests/test_full_size_gpu.py -m gpu -x -q -k "not sdxl and not prior" > $out/pytest.log 2>&1
echo "pytest rc=$?" | tee -a $out/pytest.log
tail -3 $out/pytest.log
B="--steps 40 --warmup 8 --no-secondary --no-cpu-baseline"
run() { name=$1; shift; env "$@" timeout 150 python bench.py $B > $out/$name.json 2> $out/$name.err; python -c "import json; d=json.load(open('$out/$name.json')); r=d['roofline']; print('$name', d['ms_per_step'], r['frac'], r['single_stream']['frac'], {k: v['ms_per_step_single_stream'] for k, v in list(r['families'].items())[:8]})"; }
run base X=1
run unfused EEGCLIP_CONV_BWD_FUSED=0
run base2 X=1
timeout 200 python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline --breakdown > $out/breakdown.json 2> $out/breakdown.txt
grep "^#" $out/breakdown.txt | head -24
