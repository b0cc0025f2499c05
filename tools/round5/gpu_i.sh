#!/bin/bash
# round 5, call I: stacked query gradient (one dA launch), per-phase kernel-argument reload in the fused block, AdamW split around the embedding gradient
out=gpurun_out/r5i
mkdir -p $out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
F='RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids'
(timeout 400 python -m pytest tests/test_token_block.py tests/test_kernels_gemm_x3.py tests/test_model_gpu.py -m gpu -q -x -p no:cacheprovider -k "token_block or split_rows or single_submission or reconstruction_train or train_model_matches or clip or loss" 2>&1 | grep -v "$F" | tail -30) > $out/tests_a.log 2>&1
tail -4 $out/tests_a.log
B="--steps 60 --warmup 8 --no-secondary --no-cpu-baseline"
run() { name=$1; shift; env "$@" timeout 200 python bench.py $B > $out/$name.json 2> $out/$name.err; python -c "import json; d=json.load(open('$out/$name.json')); c=d['config']; print('$name', d['ms_per_step'], c.get('host_enqueue_ms_per_step'), c.get('launches_per_step'), d['roofline']['kernel'][:40], d['roofline']['frac'])" || tail -5 $out/$name.err; }
run plan X=1
run nosplit EEGCLIP_ADAM_SPLIT=0
run plan2 X=1
run noplan EEGCLIP_STEP_PLAN=0
EEGCLIP_TB_DEBUG=2 timeout 200 python tools/bench_token_block.py 40 2>&1 | grep -v "^\[" | tail -4
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/prof -o trace -- python $R/bench.py --steps 30 --warmup 5 --no-secondary --no-cpu-baseline > $R/$out/bench_prof.json 2> $R/$out/prof.err)
f=$(find $out/prof -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp $f $out/step_kernel_stats.csv
rm -rf $out/prof
python - <<PY
import csv
rows=list(csv.DictReader(open("$out/step_kernel_stats.csv")))
calls=[int(r['Calls']) for r in rows if 'cstack_fwd' in r['Name']]
steps=calls[0] if calls else 1
tot=sum(float(r['TotalDurationNs']) for r in rows)
print('total kernel us/step', round(tot/steps/1e3,1), 'launches/step', round(sum(int(r['Calls']) for r in rows)/steps,1))
for r in rows:
    n=r['Name'].split('(')[0].replace('void ','')[:56]
    if int(r['Calls'])/steps > 0.3: print(f"{n:56s} {int(r['Calls'])/steps:5.2f} avg {float(r['AverageNs'])/1e3:7.1f} per-step {float(r['TotalDurationNs'])/steps/1e3:7.1f}")
PY
(timeout 900 python -m pytest tests/test_kernels_cstack.py tests/test_kernels_ops.py tests/test_model_gpu.py tests/test_full_size_gpu.py tests/test_dp_gpu.py tests/test_dataset_gpu.py tests/test_token_block.py tests/test_host_api.py -m gpu -q -p no:cacheprovider -k "not sdxl and not prior" 2>&1 | grep -v "$F" | tail -12) > $out/tests_model.log 2>&1
tail -4 $out/tests_model.log
