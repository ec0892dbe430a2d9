#!/bin/bash
# round 5: traverse on the device -- the variant tests, config 4 at its size through API and CLI (device traverse on / off)
set -u
cd "$(dirname "$0")/.."
ROOT=$PWD
OUT=$ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
( timeout 2400 python -m pytest tests/test_gpu_variants.py tests/test_cli.py tests/test_gpu_cfg4_fullsize.py tests/test_gpu_knobs.py -x -q 2>&1 | tail -12 ) > $OUT/r05_tenth_tests.txt
tail -4 $OUT/r05_tenth_tests.txt
( MFX_TMP=/dev/shm/mfx_cfg4 MFX_CFG4_SLEEP=6 MFX_CFG4_SLOTS=1,1,1 timeout 1500 python tools/cfg4_polish_timing.py 3e9 3.9e6 cli 2>&1 | grep -v "^$" | cut -c1-260 ) > $OUT/r05_cfg4_cli_trv.txt
grep "SLOTS=1\|mfx_variants\]\|timing:\|clusters in\|8 slots ==" $OUT/r05_cfg4_cli_trv.txt | tail -16
( MFX_VAR_DEVICE_TRAVERSE=0 MFX_TMP=/dev/shm/mfx_cfg4 MFX_CFG4_SLEEP=6 MFX_CFG4_SLOTS=1,1 timeout 1500 python tools/cfg4_polish_timing.py 3e9 3.9e6 cli 2>&1 | grep -v "^$" | cut -c1-260 ) > $OUT/r05_cfg4_cli_notrv.txt
rm -rf /dev/shm/mfx_cfg4
grep "SLOTS=1\|mfx_variants\]\|timing:\|clusters in" $OUT/r05_cfg4_cli_notrv.txt | tail -10
