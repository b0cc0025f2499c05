#!/bin/bash
# round 6: the evidence run -- the full bench line, the full -m gpu suite, smoke(), then tools/final_profiles.sh (PMC passes + kernel trace + timeline +
# InfoNCE PMC), the head-GEMM shapes alone, the data-parallel step plan on two gloo ranks of this one GPU
out=gpurun_out/r6_final
mkdir -p $out
export TMPDIR=/tmp
F='RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids'
# the bench line first, on the fresh box (its `traffic` fields then come from the committed PMC table of the previous evidence run of the same kernels)
timeout 400 python bench.py --steps 50 --warmup 10 > $out/bench.json 2> $out/bench.err < /dev/null
cut -c1-300 $out/bench.json
export SKIP_BENCH=1
(timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | grep -v "$F" | tail -15) > $out/tests_gpu.log 2>&1
tail -3 $out/tests_gpu.log
(timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | grep -v "$F" | tail -5) > $out/smoke.log 2>&1
tail -2 $out/smoke.log
bash tools/final_profiles.sh r6_final > $out/final_profiles.log 2>&1
tail -5 $out/final_profiles.log | cut -c1-300
timeout 300 python tools/round6/bench_infonce_tiles.py $out/infonce_tiles.json > $out/infonce_tiles.txt 2>&1
tail -3 $out/infonce_tiles.txt | cut -c1-600
timeout 300 python tools/round6/bench_head_gemm.py > $out/head_gemm_bench.txt 2>&1
tail -12 $out/head_gemm_bench.txt
ls $out | head -40
