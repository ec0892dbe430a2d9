#!/bin/bash
# round 5, third GPU call: the deferred tail of the probe (correctness, then A/B at 3 Gb: k = 21 with w = 4 / 5, k = 31), the staged load again
set -u
cd "$(dirname "$0")/.."
ROOT=$PWD
OUT=$ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
( timeout 2400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_seqonly.py tests/test_gpu_bench_path.py tests/test_gpu_knobs.py tests/test_gpu_streamed_multi.py tests/test_gpu_fullsize.py tests/test_gpu_cfg1.py tests/test_gpu_null_stream.py -x -q 2>&1 | tail -25 ) > $OUT/r05_third_tests.txt
one() {   # label, lib ("default" or path), env spec, extra bench flags
  local label=$1 lib=$2 spec=$3; shift 3
  if [ "$lib" = "default" ]; then unset MFX_LIB; else export MFX_LIB=$ROOT/$lib; fi
  env $spec python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-e2e --no-k31 --no-full-index --no-streamed "$@" 2>>$OUT/r05_third_err.txt | python -c "
import json,sys
d=json.load(sys.stdin); r=d['roofline']
print('%-22s' % '$label', '%-12s' % '$spec', '%.2f G k-mers/s' % (d['value']/1e9), '%.3f ms' % d['ms_per_step'], 'table %.1f GB' % d['config']['index_gb'], 'lines/k-mer %.4f' % (r.get('lines_per_kmer') or 0), 'HBM frac %.3f' % (r.get('frac') or 0), 'VALU/k-mer %.1f' % ((r.get('issue') or {}).get('valu_insts_per_kmer') or 0), 'VALU frac %.3f' % ((r.get('issue') or {}).get('valu_issue_frac') or 0), 'kmissing', d['config']['kmissing'], 'koverCpy %.7f' % d['config']['koverCpy'])
"
  unset MFX_LIB
}
{
echo "# -hist k = 21 / 3 Gb, sequence-only compact index at load factor 0.18: the probe's tail deferred (flush at 32 parked queries; 16; 48) against not deferred; windows w = 4 (t = 6) and w = 5 (t = 7)"
one "defer (flush 32)" default MFX_X=1
one "not deferred" tools/_build/ab/lib_nodefer.so MFX_X=1
one "defer (flush 32)" default MFX_MZ_W=5
one "not deferred" tools/_build/ab/lib_nodefer.so MFX_MZ_W=5
one "defer (flush 16)" tools/_build/ab/lib_defer16.so MFX_X=1 --no-pmc
one "defer (flush 48)" tools/_build/ab/lib_defer48.so MFX_X=1 --no-pmc
one "defer (flush 32)" default MFX_X=2 --no-pmc
one "not deferred" tools/_build/ab/lib_nodefer.so MFX_X=2 --no-pmc
one "defer (flush 32)" default MFX_MZ_W=5 --no-pmc
for lf in 0.25 0.4; do
  one "defer lf $lf" default "MFX_LOAD_FACTOR=$lf" --no-pmc
  one "not deferred lf $lf" tools/_build/ab/lib_nodefer.so "MFX_LOAD_FACTOR=$lf" --no-pmc
  one "defer lf $lf w5" default "MFX_LOAD_FACTOR=$lf MFX_MZ_W=5" --no-pmc
done
} > $OUT/r05_defer_ab.txt 2>&1
k31() {
  local label=$1 lib=$2
  if [ "$lib" = "default" ]; then unset MFX_LIB; else export MFX_LIB=$ROOT/$lib; fi
  python - "$label" <<'PY' 2>>$OUT/r05_third_err.txt
import sys, os, json
sys.path.insert(0, os.getcwd())
import torch
import merfin_amd as m
from tools import synth_torch as st
import bench
kp = m.KParams.from_file(26.0, os.path.join("tests", "golden", "example_lookup_table.txt"))
r = bench.k31_leg(m, st, torch, 3_000_000_000, 26.0, kp, 0, True, 10)
ro = r["roofline"]
print("%-22s k = 31: %.2f G k-mers/s  %.3f ms  table %.1f GB  lines/k-mer %.4f  HBM frac %.3f  VALU/k-mer %.1f  VALU frac %.3f  kmissing %d" % (
    sys.argv[1], r["value"] / 1e9, r["ms_per_step"], r["index_gb"], ro.get("lines_per_kmer") or 0, ro.get("frac") or 0,
    (ro.get("issue") or {}).get("valu_insts_per_kmer") or 0, (ro.get("issue") or {}).get("valu_issue_frac") or 0, r["kmissing"]))
PY
  unset MFX_LIB
}
{
k31 "defer (flush 32)" default
k31 "not deferred" tools/_build/ab/lib_nodefer.so
} >> $OUT/r05_defer_ab.txt 2>&1
DIR=/dev/shm/mfx_r05_$$
python - "$DIR" <<'PY' > $OUT/r05_inputs3.log 2>&1
import sys, os
sys.path.insert(0, os.getcwd())
import torch
import merfin_amd as m
from tools import synth_torch as st, e2e_inputs
inp = e2e_inputs.write_inputs(m, st, torch, 3_000_000_000, sys.argv[1], ncontigs=24, k=21, lam=26.0)
print("inputs written", inp["db_bytes"], inp["read_kmers"], inp["write_s"])
PY
CMD="$ROOT/merfin_amd/bin/merfin -hist -sequence $DIR/asm.fasta -readmers $DIR/read.mfxk -peak 26 -prob $ROOT/tests/golden/example_lookup_table.txt -output $DIR/o.hist"
sleep 10
{
echo "# merfin -hist at 3 Gb (5.9 G-k-mer delta-coded read database, 14.15 GB), table load factor 0.4: staged database load (escapes staged, 4 reader threads until the sequence is uploaded) against MFX_DB_STAGE=0; 4 runs BACK TO BACK then 2 spaced by 5 s"
for spec in "MFX_X=1" "MFX_DB_STAGE=0" "MFX_DB_STAGE_THREADS=2" "MFX_DB_STAGE_THREADS=8" "MFX_X=2" "MFX_DB_STAGE=0"; do
  sleep 8
  for rep in 1 2 3 4 5 6; do
    [ $rep -ge 5 ] && sleep 5
    s=$(date +%s.%N)
    env $spec MFX_CLI_TIMING=2 MFX_INGEST_TIMING=1 $CMD 2> $DIR/err.txt
    e=$(date +%s.%N)
    echo "$spec $([ $rep -ge 5 ] && echo spaced || echo b2b) rep $rep wall $(python3 -c "print(round($e - $s, 3))") s  $(grep -h 'timing' $DIR/err.txt | tr '\n' ' ' | cut -c1-420)  md5 $(md5sum < $DIR/o.hist | cut -c1-8)"
    grep -h 'staged build' $DIR/err.txt | head -1 | cut -c1-400 | sed 's/^/      /'
  done
done
} > $OUT/r05_e2e_staged2.txt 2>&1
rm -rf $DIR
