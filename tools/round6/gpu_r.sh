#!/bin/bash
# round 6, call R: InfoNCE tile kernel -- fragment reads behind a step's first MFMA (all tile forms), 256-tile with / without the cross-barrier prefetch
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-r6r}
mkdir -p $O
F='amdgpu.ids\|RCCL\|HIP version\|ROCm version\|Hostname\|Librccl'
timeout 900 python -m pytest tests/test_kernels_infonce_fused.py tests/test_infonce_sharded.py -x -q -m gpu 2>&1 | grep -v "$F" | tail -3 > $O/tests.txt
cat $O/tests.txt
timeout 300 python tools/round6/bench_infonce_tiles.py $O/infonce_tiles.json 2>&1 | grep -v "$F" | tail -5
timeout 300 python tools/with_lib.py eeg_image_decode_amd/csrc/libeegclip_hip_prev.so tools/round6/bench_infonce_tiles.py $O/infonce_tiles_prev.json 2>&1 | grep -v "$F" | tail -5
timeout 300 python tools/bench_infonce_fused.py --out $O/infonce_fused.json 2>&1 | grep -v "$F" | cut -c1-1500 | tail -4
