#!/bin/bash
# round 5: layout 9 with the direct form's offset mini-bucket (placed records scanned for it) against the layout-8 build; when the
# k = 21 headline is back (>= 148.5 G k-mers/s), the final run (tools/r05_final.sh) follows in the same call
set -u
cd "$(dirname "$0")/.."
ROOT=$PWD
OUT=$ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
( timeout 2400 python -m pytest tests/test_gpu_placed_db.py tests/test_gpu_seqonly.py tests/test_gpu_parity.py tests/test_gpu_bench_path.py tests/test_golden.py -x -q 2>&1 | tail -6 ) > $OUT/r05_place_ab3_tests.txt
grep -h "passed\|failed" $OUT/r05_place_ab3_tests.txt
one() {
  local label=$1 lib=$2; shift 2
  if [ "$lib" = "default" ]; then unset MFX_LIB; else export MFX_LIB=$ROOT/$lib; fi
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-e2e --no-k31 --no-full-index --no-streamed "$@" 2>>$OUT/r05_place_ab_err.txt | python -c "
import json,sys
d=json.load(sys.stdin); r=d['roofline']
print('%-56s' % '$label', '%.2f G k-mers/s' % (d['value']/1e9), '%.3f ms' % d['ms_per_step'], 'lines/k-mer %.4f' % (r.get('lines_per_kmer') or 0), 'VALU/k-mer %.1f' % ((r.get('issue') or {}).get('valu_insts_per_kmer') or 0), 'kmissing', d['config']['kmissing'], 'koverCpy %.7f' % d['config']['koverCpy'])
"
  unset MFX_LIB
}
{
one "layout 9: mix line, offset bucket (k <= 21)" default
one "layout-8 line, offset bucket (= layout 8)" tools/_build/ab/lib_oldboth.so
one "layout 9: mix line, offset bucket (k <= 21)" default --no-pmc
one "layout-8 line, offset bucket (= layout 8)" tools/_build/ab/lib_oldboth.so --no-pmc
} > $OUT/r05_place_ab3.txt 2>&1
cat $OUT/r05_place_ab3.txt
python - <<'PY' 2>/dev/null | tail -4
import os, sys
sys.path.insert(0, os.getcwd())
import torch, numpy as np
import merfin_amd as m
from tools import synth_torch as st
import bench
kp = m.KParams.from_file(26.0, os.path.join("tests", "golden", "example_lookup_table.txt"))
r = bench.k31_leg(m, st, torch, 3_000_000_000, 26.0, kp, 0, False, 10)
print("k = 31: %.2f G k-mers/s  %.3f ms  kmissing %d" % (r["value"] / 1e9, r["ms_per_step"], r["kmissing"]))
PY
V=$(awk 'NR==3 {print int($0 ~ /G k-mers/ ? 1 : 0)}' $OUT/r05_place_ab3.txt)
G=$(sed -n 3p $OUT/r05_place_ab3.txt | grep -o "[0-9.]* G k-mers/s" | cut -d' ' -f1)
echo "k21 (no pmc) = $G"
if python -c "import sys; sys.exit(0 if float('${G:-0}') >= 148.5 else 1)"; then
  bash tools/r05_final.sh
else
  echo "final run skipped"
fi
