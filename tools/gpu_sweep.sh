#!/bin/bash
out=gpurun_out/${1:-sweep}
mkdir -p $out
export TMPDIR=/tmp
B="--steps 40 --warmup 8 --no-secondary --no-cpu-baseline"
run() { name=$1; shift; env "$@" timeout 120 python bench.py $B > $out/$name.json 2> $out/$name.err; python -c "import json; d=json.load(open('$out/$name.json')); print('$name', d['ms_per_step'])"; }
run base_1 X=1
run wsk16 EEGCLIP_WGRAD_SK=16
run wsk24 EEGCLIP_WGRAD_SK=24
run wsk48 EEGCLIP_WGRAD_SK=48
run wsk64 EEGCLIP_WGRAD_SK=64
run base_2 X=1
run headsk4 EEGCLIP_HEAD_SK=4
run headsk16 EEGCLIP_HEAD_SK=16
run scwns256 EEGCLIP_SCW_NS=256
run scwg32 EEGCLIP_SCW_G=32
run scwg80 EEGCLIP_SCW_G=80
run tswr6 EEGCLIP_TSW_R=6
run scfx3 EEGCLIP_SCONV_FWD_X3=1
run lnside0 EEGCLIP_LN_SIDE=0
run base_3 X=1
