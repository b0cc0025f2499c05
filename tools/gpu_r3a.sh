#!/bin/bash
# round 3, GPU call A: the whole -m gpu suite (new: fused row-sharded InfoNCE at configs[2] sizes, K/V cache, F1 full width), the retrieval-accuracy
# parity run in the default split-bf16 arithmetic with 1000 held-out classes, the bench line of the round's starting point
out=gpurun_out/r3a
mkdir -p $out
export TMPDIR=/tmp
(time python -m pytest tests -m gpu -q -p no:cacheprovider -x 2>&1 | tail -40) > $out/tests.log 2>&1
timeout 900 python tools/accuracy_parity.py 150 128 1000 > $out/accuracy_parity.json 2> $out/accuracy_parity.err
timeout 600 python bench.py --steps 30 --warmup 5 > $out/bench.json 2> $out/bench.err
tail -6 $out/tests.log; head -c 1500 $out/accuracy_parity.json; echo; head -c 600 $out/bench.json
