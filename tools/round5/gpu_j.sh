#!/bin/bash
# round 5, call J: kernel trace of the step after merging the weight-gradient launches + the model-level GPU tests
out=gpurun_out/r5j
mkdir -p $out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
F='RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids'
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
(timeout 900 python -m pytest tests/test_kernels_cstack.py tests/test_kernels_ops.py tests/test_kernels_wgrad.py tests/test_model_gpu.py tests/test_full_size_gpu.py tests/test_dp_gpu.py tests/test_dataset_gpu.py tests/test_token_block.py tests/test_host_api.py -m gpu -q -p no:cacheprovider -k "not sdxl and not prior" 2>&1 | grep -v "$F" | tail -12) > $out/tests_model.log 2>&1
tail -4 $out/tests_model.log
