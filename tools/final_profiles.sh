set -x
export TMPDIR=/tmp
O=gpurun_out/final
mkdir -p $O
timeout 300 python bench.py --steps 50 --warmup 10 > $O/bench.json 2> $O/bench.err < /dev/null
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o st -- python bench.py --steps 30 --warmup 8 --no-cpu-baseline > $O/stats.log 2>&1 < /dev/null
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o f -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline > $O/pmc_f.log 2>&1 < /dev/null
timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o w -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline > $O/pmc_w.log 2>&1 < /dev/null
timeout 200 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT --output-format csv -d $O/pmc_sq -o s -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline > $O/pmc_s.log 2>&1 < /dev/null
find $O -name "*.csv" | head -20
cut -c1-300 $O/bench.json
