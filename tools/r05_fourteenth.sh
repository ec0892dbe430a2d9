#!/bin/bash
# round 5: stage A ahead (mfx_vcf_prepare) once the clusters are enumerated on the device: config 4 through the CLI, MFX_CLI_VCF_AHEAD=1 (load ahead) against 2 (load + stage A ahead)
set -u
cd "$(dirname "$0")/.."
ROOT=$PWD
OUT=$ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
( MFX_TMP=/dev/shm/mfx_cfg4 MFX_CFG4_SLEEP=6 MFX_CFG4_SLOTS=1 MFX_CFG4_AHEAD_AB=1 timeout 1500 python tools/cfg4_polish_timing.py 3e9 3.9e6 cli 2>&1 | grep -v "^$" | cut -c1-260 ) > $OUT/r05_cfg4_cli_ahead_trv.txt
rm -rf /dev/shm/mfx_cfg4
grep "wall=\|mfx_variants\]\|timing:" $OUT/r05_cfg4_cli_ahead_trv.txt | grep -v "load:" | tail -22
