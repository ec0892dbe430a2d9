#!/bin/bash
# kernel trace of BASELINE config 5's shape on one GPU (tests/test_gpu_cfg5_shape.py: 3 Gb, k = 31, 8 shards on device 0): the rows of
# the route -> owner loop (mfx_route_*_kernel, mfx_hist_keys_kernel, copies) for profiles/.   tools/cfg5_profile.sh <outprefix>
set -u
OUT=$(realpath -m $1)
cd "$(dirname "$0")/.."
ROOT=$PWD
export TMPDIR=/tmp
D=/tmp/cfg5kt_$$
( cd /tmp && timeout 1500 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $D -o kt -- python -m pytest $ROOT/tests/test_gpu_cfg5_shape.py -x -q -s -m gpu ) > ${OUT}_trace.log 2>&1
for f in kernel_stats memory_copy_stats; do F=$(find $D -name "*${f}.csv" | head -1); [ -n "$F" ] && cp "$F" ${OUT}_${f}.csv; done
rm -rf $D
grep -E "config-5 shape|passed|failed" ${OUT}_trace.log | tail -5
