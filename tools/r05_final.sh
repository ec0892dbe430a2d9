#!/bin/bash
# round 5: the whole -m gpu suite on the final source, smoke, the default bench (all legs, PMC kept), rocprofv3 kernel stats of a bench run
set -u
cd "$(dirname "$0")/.."
ROOT=$PWD
OUT=$ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
( timeout 3000 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -15 ) > $OUT/r05_final_pytest.txt
tail -3 $OUT/r05_final_pytest.txt
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/r05_final_smoke.txt 2>&1; tail -1 $OUT/r05_final_smoke.txt
rm -rf $OUT/r05_pmc_final
MFX_BENCH_KEEP_PMC=$OUT/r05_pmc_final python bench.py --steps 20 --warmup 5 > $OUT/r05_bench_final.json 2> $OUT/r05_bench_final.log
tail -c 300 $OUT/r05_bench_final.log
D=/tmp/kt_bench
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $D -o kt -- python $ROOT/bench.py --steps 10 --warmup 3 --no-pmc --no-cpu-baseline --no-e2e --no-full-index --no-streamed ) > $OUT/r05_final_trace.log 2>&1
F=$(find $D -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && cp "$F" $OUT/r05_final_bench_kernel_stats.csv
rm -rf $D
head -5 $OUT/r05_final_bench_kernel_stats.csv | cut -c1-200
