#!/bin/bash
# round 6, call H: the row-sharded loss on planes
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r6h
timeout 900 python -m pytest tests/test_infonce_sharded.py tests/test_kernels_infonce_fused.py tests/test_dp_gpu.py -x -q -m gpu 2>&1 | tail -6 > gpurun_out/r6h/tests.txt
cat gpurun_out/r6h/tests.txt
timeout 300 python -c "
import json, os, bench
print(json.dumps(bench._sec_infonce_per_rank()), flush=True)
os.environ['EEGCLIP_SHARDED_PLANES'] = '0'
print('old', json.dumps(bench._sec_infonce_per_rank()), flush=True)
" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6h/per_rank.txt
