#!/bin/bash
# round 6, call A: the new head GEMM on the hardware (parity + microbench over slice counts) and the baseline step on the same box
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
python -m pytest tests/test_kernels_head_gemm.py tests/test_kernels_gemm_planes.py -x -q -m gpu 2>&1 | tail -5 > gpurun_out/r6_a_tests.txt
python tools/round6/bench_head_gemm.py > gpurun_out/r6_a_head_bench.txt 2>&1
python bench.py --steps 200 --warmup 30 > gpurun_out/r6_a_bench_baseline.json 2> gpurun_out/r6_a_bench_baseline.err
cat gpurun_out/r6_a_tests.txt gpurun_out/r6_a_head_bench.txt
head -c 600 gpurun_out/r6_a_bench_baseline.json
