#!/bin/bash
out=gpurun_out/r4f
mkdir -p $out
export TMPDIR=/tmp
python tools/bench_conv_bwd.py $out/conv_bwd_bench.json
python -m pytest tests/test_kernels_ops.py -m gpu -q -k conv_backward_fused 2>&1 | tail -2
B="--steps 40 --warmup 8 --no-secondary --no-cpu-baseline"
run() { name=$1; shift; env "$@" timeout 150 python bench.py $B > $out/$name.json 2> $out/$name.err; python -c "import json; d=json.load(open('$out/$name.json')); r=d['roofline']; print('$name', d['ms_per_step'], r['frac'], r['single_stream']['frac'], {k: v['ms_per_step_single_stream'] for k, v in list(r['families'].items())[:6]})"; }
run base X=1
run fused EEGCLIP_CONV_BWD_FUSED=1
run base2 X=1
run fused2 EEGCLIP_CONV_BWD_FUSED=1
EEGCLIP_CONV_BWD_FUSED=1 timeout 300 python -m pytest tests/test_model_gpu.py tests/test_full_size_gpu.py -m gpu -x -q -k "not sdxl and not prior" 2>&1 | tail -2
bash tools/gpu_pmc_cmd.sh r4f_pmc conv_bwd_fused bench_conv_bwd.py /tmp/x.json fused 2>&1 | tail -36
