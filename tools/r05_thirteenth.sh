#!/bin/bash
set -u
cd "$(dirname "$0")/.."
ROOT=$PWD
OUT=$ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
( MFX_VAR_TIMING=1 MFX_TMP=/dev/shm/mfx_cfg4 MFX_CFG4_SLEEP=6 MFX_CFG4_SLOTS=1,1,1 timeout 1500 python tools/cfg4_polish_timing.py 3e9 3.9e6 cli 2>&1 | grep -v "^$" | cut -c1-260 ) > $OUT/r05_cfg4_cli_trv.txt
grep "SLOTS=1\|mfx_variants\]\|timing:\|clusters in\|8 slots ==" $OUT/r05_cfg4_cli_trv.txt | grep -v "load:" | tail -18
( MFX_VAR_DEVICE_TRAVERSE=0 MFX_VAR_TIMING=1 MFX_TMP=/dev/shm/mfx_cfg4 MFX_CFG4_SLEEP=6 MFX_CFG4_SLOTS=1,1,1 timeout 1500 python tools/cfg4_polish_timing.py 3e9 3.9e6 cli 2>&1 | grep -v "^$" | cut -c1-260 ) > $OUT/r05_cfg4_cli_notrv.txt
rm -rf /dev/shm/mfx_cfg4
grep "SLOTS=1\|mfx_variants\]\|timing:\|clusters in" $OUT/r05_cfg4_cli_notrv.txt | grep -v "load:" | tail -14
