#!/bin/bash
# round 5: kernel trace of config 4 through the API with the clusters enumerated on the device
set -u
cd "$(dirname "$0")/.."
ROOT=$PWD
OUT=$ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
D=/tmp/kt_cfg4
( cd /tmp && MFX_TMP=/dev/shm/mfx_cfg4 MFX_VAR_TIMING=1 timeout 1200 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $D -o kt -- python $ROOT/tools/cfg4_polish_timing.py 3e9 3.9e6 ) > $OUT/r05_cfg4_trv_trace.log 2>&1
rm -rf /dev/shm/mfx_cfg4
F=$(find $D -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && cp "$F" $OUT/r05_cfg4_trv_kernel_stats.csv
F=$(find $D -name "*memory_copy_stats.csv" | head -1); [ -n "$F" ] && cp "$F" $OUT/r05_cfg4_trv_memcpy_stats.csv
rm -rf $D
grep "mfx_variants\]\|clusters in" $OUT/r05_cfg4_trv_trace.log | head
grep -i "traverse\|var_score\|dump_kernel" $OUT/r05_cfg4_trv_kernel_stats.csv | cut -c1-200
cat $OUT/r05_cfg4_trv_memcpy_stats.csv | head -8 | cut -c1-200
