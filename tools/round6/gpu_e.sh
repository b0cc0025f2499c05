#!/bin/bash
# round 6, call E: the VAE (csrc/vae.hip) on the hardware -- layer tests, end-to-end vs the fp32 restatement, the 1024-px decode time
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r6e
timeout 900 python -m pytest tests/test_vae_gpu.py tests/test_sdxl_gpu.py -x -q -m gpu 2>&1 | tail -40 > gpurun_out/r6e/tests.txt
cat gpurun_out/r6e/tests.txt
timeout 300 python -c "
import json, torch
from eeg_image_decode_amd import vae
for lat in (64, 128):
    print(json.dumps(vae.bench_decode(images=1, latent=lat)), flush=True)
" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6e/vae_decode.txt
