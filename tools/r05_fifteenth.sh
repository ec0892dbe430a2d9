#!/bin/bash
set -u
cd "$(dirname "$0")/.."
OUT=$PWD/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
( timeout 2400 python -m pytest tests/test_gpu_variants.py tests/test_gpu_knobs.py tests/test_cli.py -x -q 2>&1 | tail -12 ) > $OUT/r05_fifteenth_tests.txt
tail -6 $OUT/r05_fifteenth_tests.txt
