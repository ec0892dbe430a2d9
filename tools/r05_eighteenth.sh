#!/bin/bash
set -u
cd "$(dirname "$0")/.."
OUT=$PWD/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
( timeout 2400 python -m pytest tests/test_gpu_variants.py tests/test_gpu_cfg4_fullsize.py tests/test_cli.py -x -q 2>&1 | tail -6 ) > $OUT/r05_eighteenth_tests.txt
tail -3 $OUT/r05_eighteenth_tests.txt
( MFX_VAR_TIMING=1 MFX_TMP=/dev/shm/mfx_cfg4 MFX_CFG4_SLEEP=6 MFX_CFG4_SLOTS=1,1,1,1 timeout 1500 python tools/cfg4_polish_timing.py 3e9 3.9e6 cli 2>&1 | grep -v "^$" | cut -c1-260 ) > $OUT/r05_cfg4_cli_pool.txt
rm -rf /dev/shm/mfx_cfg4
grep "SLOTS=1\|mfx_variants\]\|timing:\|clusters in\|8 slots ==" $OUT/r05_cfg4_cli_pool.txt | grep -v "load:" | tail -18
