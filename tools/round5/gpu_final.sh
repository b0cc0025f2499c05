#!/bin/bash
# round 5: the evidence run -- full -m gpu suite, then tools/final_profiles.sh (PMC passes + kernel trace + full bench line + InfoNCE PMC), PMC of the cstack kernels
out=gpurun_out/r5_final
mkdir -p $out
export TMPDIR=/tmp
F='RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids'
(timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | grep -v "$F" | tail -15) > $out/tests_gpu.log 2>&1
tail -3 $out/tests_gpu.log
bash tools/final_profiles.sh r5_final > $out/final_profiles.log 2>&1
tail -5 $out/final_profiles.log | cut -c1-300
python tools/bench_cstack.py $out/cstack_bench.json
bash tools/gpu_pmc_cmd.sh r5_final/pmc_cstack cstack bench_cstack.py > $out/pmc_cstack.log 2>&1
find $out/pmc_cstack -name "*.csv" -delete
ls $out | head -40
