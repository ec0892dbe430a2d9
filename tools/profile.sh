#!/bin/bash
# rocprofv3 evidence for bench.py (run on the GPU box via gpurun):
#   1. the default bench command, un-profiled; its own two PMC passes (FETCH_SIZE, WRITE_SIZE: separate runs, TCC slots)
#      leave their raw rows in $OUT/pmc (MFX_BENCH_KEEP_PMC)
#   2. --kernel-trace --stats of the same command (--no-pmc: the PMC children must not run under an outer rocprofv3; --no-streamed: every hist launch but the cpu-baseline sample is then a full-size one, so the stats row averages cleanly)
# Outputs: gpurun_out/prof_$TAG/ ; copy the summaries you want judged into profiles/.
TAG=${1:-r02}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B0=$SECONDS; MFX_BENCH_KEEP_PMC=$OUT/pmc python $REPO/bench.py --gpus 1 --steps 10 --warmup 3 > $OUT/bench.json 2> $OUT/bench.log; echo "default bench run: $((SECONDS - B0)) s wall" | tee $OUT/bench_wall.txt
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python $REPO/bench.py --gpus 1 --steps 10 --warmup 3 --no-pmc --no-streamed --no-e2e --no-full-index > $OUT/bench_under_rocprof.json 2> $OUT/bench_under_rocprof.log
find $OUT -name "*.csv" | head -50
# keep the merged-back payload small: drop per-launch traces
find $OUT -name "*kernel_trace.csv" -size +20M -delete
du -sh $OUT
