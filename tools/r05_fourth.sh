#!/bin/bash
# round 5, fourth GPU call: kernel A/B (tabulated over-copy term, two-level sliding minimum, deferred tail), k = 31, cfg4 through the CLI with
# delta-coded databases, PMC rows of the route -> owner kernels
set -u
cd "$(dirname "$0")/.."
ROOT=$PWD
OUT=$ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
( timeout 2400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_seqonly.py tests/test_gpu_bench_path.py tests/test_gpu_knobs.py tests/test_gpu_streamed_multi.py tests/test_gpu_fullsize.py tests/test_gpu_cfg1.py tests/test_golden.py tests/test_gpu_variants.py -x -q 2>&1 | tail -25 ) > $OUT/r05_fourth_tests.txt
one() {   # label, lib ("default" or path), env spec, extra bench flags
  local label=$1 lib=$2 spec=$3; shift 3
  if [ "$lib" = "default" ]; then unset MFX_LIB; else export MFX_LIB=$ROOT/$lib; fi
  env $spec python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-e2e --no-k31 --no-full-index --no-streamed "$@" 2>>$OUT/r05_fourth_err.txt | python -c "
import json,sys
d=json.load(sys.stdin); r=d['roofline']
print('%-34s' % '$label', '%-12s' % '$spec', '%.2f G k-mers/s' % (d['value']/1e9), '%.3f ms' % d['ms_per_step'], 'table %.1f GB' % d['config']['index_gb'], 'lines/k-mer %.4f' % (r.get('lines_per_kmer') or 0), 'HBM frac %.3f' % (r.get('frac') or 0), 'VALU/k-mer %.1f' % ((r.get('issue') or {}).get('valu_insts_per_kmer') or 0), 'VALU frac %.3f' % ((r.get('issue') or {}).get('valu_issue_frac') or 0), 'kmissing', d['config']['kmissing'], 'koverCpy %.7f' % d['config']['koverCpy'])
"
  unset MFX_LIB
}
{
echo "# -hist k = 21 / 3 Gb, sequence-only compact index at load factor 0.18; every build: the over-copy term of the tabulated pairs from a table in HBM (mfx_kfix), koverCpy as an integer sum"
one "defer + 2-level min (default)" default MFX_X=1
one "not deferred, 2-level min" tools/_build/ab/lib_nodefer.so MFX_X=1
one "defer, 1-level min" tools/_build/ab/lib_min1.so MFX_X=1
one "not deferred, 1-level min" tools/_build/ab/lib_nodefer_min1.so MFX_X=1
one "defer + 2-level min (default)" default MFX_MZ_W=5
one "not deferred, 2-level min" tools/_build/ab/lib_nodefer.so MFX_MZ_W=5
one "defer + 2-level min (default)" default MFX_X=2 --no-pmc
one "not deferred, 2-level min" tools/_build/ab/lib_nodefer.so MFX_X=2 --no-pmc
one "defer + 2-level min, lf 0.4" default MFX_LOAD_FACTOR=0.4 --no-pmc
one "not deferred, lf 0.4" tools/_build/ab/lib_nodefer.so MFX_LOAD_FACTOR=0.4 --no-pmc
} > $OUT/r05_kernel_ab2.txt 2>&1
k31() {
  local label=$1 lib=$2
  if [ "$lib" = "default" ]; then unset MFX_LIB; else export MFX_LIB=$ROOT/$lib; fi
  python - "$label" <<'PY' 2>>$OUT/r05_fourth_err.txt
import sys, os, json
sys.path.insert(0, os.getcwd())
import torch
import merfin_amd as m
from tools import synth_torch as st
import bench
kp = m.KParams.from_file(26.0, os.path.join("tests", "golden", "example_lookup_table.txt"))
r = bench.k31_leg(m, st, torch, 3_000_000_000, 26.0, kp, 0, True, 10)
ro = r["roofline"]
print("%-34s k = 31: %.2f G k-mers/s  %.3f ms  table %.1f GB  lines/k-mer %.4f  HBM frac %.3f  VALU/k-mer %.1f  VALU frac %.3f  kmissing %d" % (
    sys.argv[1], r["value"] / 1e9, r["ms_per_step"], r["index_gb"], ro.get("lines_per_kmer") or 0, ro.get("frac") or 0,
    (ro.get("issue") or {}).get("valu_insts_per_kmer") or 0, (ro.get("issue") or {}).get("valu_issue_frac") or 0, r["kmissing"]))
PY
  unset MFX_LIB
}
{
k31 "defer + 2-level min (default)" default
k31 "not deferred, 2-level min" tools/_build/ab/lib_nodefer.so
k31 "defer, 1-level min" tools/_build/ab/lib_min1.so
k31 "not deferred, 1-level min" tools/_build/ab/lib_nodefer_min1.so
} >> $OUT/r05_kernel_ab2.txt 2>&1
cat $OUT/r05_kernel_ab2.txt
# config 4 through the CLI, databases sorted + delta-coded
( MFX_TMP=/dev/shm/mfx_cfg4 MFX_CFG4_SLEEP=6 MFX_CFG4_SLOTS=1,1,1 timeout 1500 python tools/cfg4_polish_timing.py 3e9 3.9e6 cli 2>&1 | grep -v "^$" | cut -c1-260 ) > $OUT/r05_cfg4_cli.txt
rm -rf /dev/shm/mfx_cfg4
# PMC rows of the route -> owner loop (config 5's shape on one GPU)
for ctr in FETCH_SIZE WRITE_SIZE; do
  D=/tmp/pmc_cfg5_$ctr
  ( cd /tmp && timeout 1200 rocprofv3 --pmc $ctr --kernel-include-regex "mfx_route_fused_kernel|mfx_hist_keys_kernel" --output-format csv -d $D -o pmc -- python -m pytest $ROOT/tests/test_gpu_cfg5_shape.py -x -q -s -m gpu ) > $OUT/r05_pmc_cfg5_$ctr.log 2>&1
  F=$(find $D -name "*counter_collection.csv" | head -1)
  [ -n "$F" ] && python - "$F" $ctr > $OUT/r05_cfg5_pmc_$ctr.txt <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(sys.argv[1])):
    if r.get("Counter_Name") != sys.argv[2]:
        continue
    n = r["Kernel_Name"].split("(")[0][:60]
    agg[n][0] += 1
    agg[n][1] += float(r["Counter_Value"])
for n, (c, v) in agg.items():
    print("%-62s launches %5d  %s sum %.6e KiB" % (n, c, sys.argv[2], v))
PY
  rm -rf $D
done
grep -h "config-5 shape\|passed\|failed" $OUT/r05_pmc_cfg5_FETCH_SIZE.log | tail -5
