#!/bin/bash
# round 5: page cache -> HBM without the CPU copy?  (tools/native/ubench_hostreg.cpp)
set -u
cd "$(dirname "$0")/.."
OUT=$PWD/gpurun_out
mkdir -p $OUT
F=/dev/shm/hostreg.bin
python - <<'PY'
import numpy as np
b = np.random.default_rng(1).integers(0, 256, 64 << 20, dtype=np.uint8).tobytes()
with open("/dev/shm/hostreg.bin", "wb") as f:
    for i in range(128):
        f.write(b)
PY
{
nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null
for t in 8 16 32; do timeout 300 tools/_build/ubench_hostreg $F 256 $t; done
timeout 300 tools/_build/ubench_hostreg $F 32 16
timeout 300 tools/_build/ubench_hostreg $F 1024 16
} > $OUT/r05_hostreg.txt 2>&1
rm -f $F
cat $OUT/r05_hostreg.txt
