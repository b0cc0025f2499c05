#!/bin/bash
out=gpurun_out/${1:-r3f}
mkdir -p $out
export TMPDIR=/tmp
(time python -m pytest tests/test_kernels_infonce_fused.py tests/test_infonce_sharded.py tests/test_token_block.py tests/test_model_gpu.py tests/test_dp_gpu.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -15) > $out/tests.log 2>&1
python tools/bench_token_block.py 40 64 > $out/time_b64.json 2> $out/time_b64.err
python tools/bench_token_block.py 40 256 > $out/time_b256.json 2> $out/time_b256.err
python tools/bench_token_block.py 40 512 > $out/time_b512.json 2> $out/time_b512.err
timeout 600 python bench.py --steps 40 --warmup 8 --no-cpu-baseline > $out/bench.json 2> $out/bench.err
tail -6 $out/tests.log; cat $out/time_b64.json $out/time_b256.json $out/time_b512.json
python - <<PY
import json
d=json.load(open("$out/bench.json"))
print(d["ms_per_step"], d["value"])
print(json.dumps(d["roofline"]["families"], indent=0))
s=d["secondary"]
for k in ("infonce_global_batch_2048","infonce_per_rank_block","exact_fp32_products","prior_train_batch_1024"): print(k, json.dumps(s.get(k)))
PY
