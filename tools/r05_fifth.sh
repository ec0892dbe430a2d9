#!/bin/bash
# round 5, fifth GPU call: the whole -m gpu suite on the candidate source, then the default bench (all legs), kernel stats of the bench
set -u
cd "$(dirname "$0")/.."
ROOT=$PWD
OUT=$ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
( timeout 3000 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -15 ) > $OUT/r05_fifth_pytest.txt
MFX_BENCH_KEEP_PMC=$OUT/r05_pmc_final python bench.py --steps 20 --warmup 5 > $OUT/r05_bench_candidate.json 2> $OUT/r05_bench_candidate.log
tail -c 400 $OUT/r05_bench_candidate.log
# k = 31 knobs
k31() {
  local label=$1 lib=$2
  if [ "$lib" = "default" ]; then unset MFX_LIB; else export MFX_LIB=$ROOT/$lib; fi
  python - "$label" <<'PY' 2>>$OUT/r05_fifth_err.txt
import sys, os, json
sys.path.insert(0, os.getcwd())
import torch
import merfin_amd as m
from tools import synth_torch as st
import bench
kp = m.KParams.from_file(26.0, os.path.join("tests", "golden", "example_lookup_table.txt"))
r = bench.k31_leg(m, st, torch, 3_000_000_000, 26.0, kp, 0, False, 10)
print("%-34s k = 31: %.2f G k-mers/s  %.3f ms  kmissing %d" % (sys.argv[1], r["value"] / 1e9, r["ms_per_step"], r["kmissing"]))
PY
  unset MFX_LIB
}
{
k31 "default (6 x 2, flush 32)" default
for v in k31_7x2 k31_5x2 k31_6x4 k31_f48 k31_f16; do k31 $v tools/_build/ab/lib_$v.so; done
k31 "default (6 x 2, flush 32)" default
} > $OUT/r05_k31_knobs.txt 2>&1
cat $OUT/r05_k31_knobs.txt
