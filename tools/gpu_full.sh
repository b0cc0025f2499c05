#!/bin/bash
# full -m gpu suite + one bench line
out=gpurun_out/${1:-full}
mkdir -p $out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/ -m gpu -x -q > $out/pytest.log 2>&1
echo "pytest rc=$?" | tee -a $out/pytest.log
tail -8 $out/pytest.log
timeout 200 python bench.py --steps 40 --warmup 8 --no-secondary --no-cpu-baseline > $out/bench.json 2> $out/bench.err
python -c "import json; d=json.load(open('$out/bench.json')); r=d['roofline']; print(d['ms_per_step'], r['frac'], {k: v['ms_per_step_single_stream'] for k, v in list(r['families'].items())[:8]})"
