#!/bin/bash
# round 5, call D: the leaner cstack kernels (ones-slot affine, packed epilogues, product-major MFMAs, per-wave staging): parity, timings, PMC
out=gpurun_out/r5f
mkdir -p $out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
F='RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids'
(timeout 300 python -m pytest tests/test_kernels_cstack.py -m gpu -q -p no:cacheprovider 2>&1 | grep -v "$F" | tail -15) > $out/tests_cstack.log 2>&1
tail -3 $out/tests_cstack.log
python tools/bench_cstack.py $out/cstack_bench.json
B="--steps 40 --warmup 8 --no-secondary --no-cpu-baseline"
run() { name=$1; shift; env "$@" timeout 200 python bench.py $B > $out/$name.json 2> $out/$name.err; python -c "import json; d=json.load(open('$out/$name.json')); print('$name', d['ms_per_step'], d['config'].get('host_enqueue_ms_per_step'))" || tail -5 $out/$name.err; }
run new X=1
run old EEGCLIP_CSTACK=0
run new2 X=1
(timeout 600 python -m pytest tests/test_model_gpu.py tests/test_full_size_gpu.py tests/test_dp_gpu.py -m gpu -q -p no:cacheprovider -k "not sdxl and not prior" 2>&1 | grep -v "$F" | tail -25) > $out/tests_model.log 2>&1
tail -12 $out/tests_model.log
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
print('total kernel us/step', round(tot/steps/1e3,1))
for r in rows[:24]:
    n=r['Name'].split('(')[0].replace('void ','')[:56]
    print(f"{n:56s} {int(r['Calls'])/steps:5.2f} avg {float(r['AverageNs'])/1e3:7.1f} per-step {float(r['TotalDurationNs'])/steps/1e3:7.1f}")
PY
true

