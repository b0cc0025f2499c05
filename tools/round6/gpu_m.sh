#!/bin/bash
# round 6, call M: the diffusion prior's training step as one submission (prior._TrainStepPlan): GPU tests, then bench._sec_prior_train with the plan on / off
# and the weight gradients beside the chain (EEGCLIP_PRIOR_WGRAD_MERGE=0) / as two launches at its end, alternated
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-r6m}
mkdir -p $O
timeout 900 python -m pytest tests/test_prior_gpu.py tests/test_full_size_gpu.py -x -q -m gpu -k "prior or pipe" 2>&1 | grep -v "amdgpu.ids\|RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -3 > $O/tests.txt
cat $O/tests.txt
for i in 1 2 3; do for cfg in "1 1" "1 2" "1 0"; do set -- $cfg
EEGCLIP_PRIOR_STEP_PLAN=$1 EEGCLIP_PRIOR_WGRAD_MERGE=$2 python -c "
import bench, json
r = bench._sec_prior_train(); print('plan=$1 merge=$2', r['ms_per_step'])
" 2>/dev/null | tail -1 | tee -a $O/prior.txt
done; done
