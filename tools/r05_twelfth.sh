#!/bin/bash
set -u
cd "$(dirname "$0")/.."
ROOT=$PWD
OUT=$ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
( timeout 2400 python -m pytest tests/test_gpu_variants.py tests/test_gpu_cfg4_fullsize.py -x -q 2>&1 | tail -12 ) > $OUT/r05_twelfth_tests.txt
tail -4 $OUT/r05_twelfth_tests.txt
bash tools/r05_eleventh.sh
