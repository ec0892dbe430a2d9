#!/bin/bash
# rocprofv3 evidence for bench.py (run on the GPU box via gpurun):
#   1. --kernel-trace --stats of the default bench command
#   2. PMC passes (FETCH_SIZE, WRITE_SIZE separately -- TCC slots) restricted to the hist kernel
#   3. FETCH_SIZE calibration on a known count of random 16-byte loads (tools/ubench_gather)
# Outputs: gpurun_out/prof_$TAG/ ; copy the summaries you want judged into profiles/.
TAG=${1:-r01}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $REPO/bench.py --gpus 1 --steps 5 --warmup 2"

timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- $BENCH > $OUT/bench_trace.json 2> $OUT/bench_trace.log
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --pmc $C --kernel-include-regex "mfx_hist" --output-format csv -d $OUT/pmc_$C -o bench -- $BENCH --no-cpu-baseline > $OUT/bench_pmc_$C.json 2> $OUT/bench_pmc_$C.log
done
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "gather_kernel|stream_kernel" --output-format csv -d $OUT/calib -o ubench -- $REPO/tools/_build/ubench_gather 32 > $OUT/calib_ubench.txt 2>&1
find $OUT -name "*.csv" | head -50
# keep the merged-back payload small: drop per-launch traces of the torch kernels
find $OUT -name "*kernel_trace.csv" -size +20M -delete
du -sh $OUT
