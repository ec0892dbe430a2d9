#!/bin/bash
set -u
cd "$(dirname "$0")/.."
OUT=$PWD/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
( timeout 2400 python -m pytest tests/test_gpu_placed_db.py tests/test_gpu_seqonly.py tests/test_golden.py tests/test_gpu_db.py -x -q 2>&1 | tail -6 ) > $OUT/r05_seventeenth_tests.txt
tail -3 $OUT/r05_seventeenth_tests.txt
bash tools/r05_sixteenth.sh
