#!/bin/bash
# after `gpurun -- 'bash tools/round6/gpu_final.sh'`: copy the evidence of gpurun_out/<tag>/ into profiles/ under the round's names and regenerate the
# numbers table.   bash tools/collect_profiles.sh [r6]
set -e
cd "$(dirname "$0")/.."
R=${1:-r6}
O=gpurun_out/${R}_final
cp $O/bench.json profiles/${R}_final_bench.json
cp $O/kernel_stats.csv profiles/${R}_final_kernel_stats.csv
cp $O/step_timeline.txt profiles/${R}_step_timeline.txt
cp $O/pmc_step.json profiles/${R}_final_pmc_step.json
cp $O/pmc_hbm_traffic.json profiles/${R}_pmc_hbm_traffic.json
cp $O/pmc_infonce.json profiles/${R}_pmc_infonce.json
[ -f $O/infonce_tiles.json ] && cp $O/infonce_tiles.json profiles/${R}_infonce_tiles.json
[ -f gpurun_out/${R}_head_gemm_bench.json ] && cp gpurun_out/${R}_head_gemm_bench.json profiles/${R}_head_gemm_bench.json
(tail -3 $O/tests_gpu.log; tail -2 $O/smoke.log) > profiles/${R}_final_tests_gpu.txt
python tools/numbers_table.py profiles/${R}_final_bench.json > profiles/${R}_numbers.md
ls -la profiles/${R}_*
