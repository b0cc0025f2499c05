#!/bin/bash
# round 6, call S: cross-barrier fragment prefetch in every 128-tile form (producer waves included), both arithmetic modes
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-r6s}
mkdir -p $O
F='amdgpu.ids\|RCCL\|HIP version\|ROCm version\|Hostname\|Librccl'
timeout 900 python -m pytest tests/test_kernels_infonce_fused.py tests/test_infonce_sharded.py -x -q -m gpu 2>&1 | grep -v "$F" | tail -3 > $O/tests.txt
cat $O/tests.txt
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_full_size_gpu.py -x -q -m gpu -k "infonce or loss or clip or sharded" 2>&1 | grep -v "$F" | tail -3
timeout 400 python tools/round6/bench_infonce_tiles.py $O/infonce_tiles.json > $O/infonce_tiles.txt 2>&1
python - <<PY
import json
d = json.load(open("$O/infonce_tiles.json"))
for n, r in d.items():
    for k, v in r.items(): print(n, k, v["logits_block_us"], v["frac_of_bf16_peak"], v["mfma_work_frac_of_peak"], v["grad_us"], v["lse_maxdiff_to_first"])
PY
