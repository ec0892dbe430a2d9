#!/bin/bash
set -u
cd "$(dirname "$0")/.."
OUT=$PWD/gpurun_out
mkdir -p $OUT
F=/dev/shm/fill.bin
python - <<'PY'
import numpy as np
b = np.random.default_rng(1).integers(0, 256, 64 << 20, dtype=np.uint8).tobytes()
with open("/dev/shm/fill.bin", "wb") as f:
    for i in range(64):
        f.write(b)
PY
timeout 600 tools/_build/ubench_fill $F > $OUT/r05_fill.txt 2>&1
rm -f $F
cat $OUT/r05_fill.txt
