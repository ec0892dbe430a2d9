#!/bin/bash
# round 5, first GPU call: new tests, e2e load-factor A/B back to back at 3 Gb, PMC rows of the build kernels, then the bench.
set -u
cd "$(dirname "$0")/.."
ROOT=$PWD
OUT=$ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests/test_gpu_bench_launch.py tests/test_gpu_streamed_multi.py -x -q 2>&1 | tail -15 ) > $OUT/r05_first_tests.txt
DIR=/dev/shm/mfx_r05_$$
python - "$DIR" <<'PY' > $OUT/r05_inputs.log 2>&1
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
echo "# merfin -hist at 3 Gb (5.9 G-k-mer delta-coded read database), load factor of the sequence-only table, 4 runs BACK TO BACK (no pause) then 2 spaced by 5 s"
for lf in 0.18 0.225 0.3 0.4 0.5; do
  sleep 8
  for rep in 1 2 3 4; do
    s=$(date +%s.%N)
    MFX_LOAD_FACTOR=$lf MFX_CLI_TIMING=2 $CMD 2> $DIR/err.txt
    e=$(date +%s.%N)
    echo "lf $lf b2b rep $rep wall $(python3 -c "print(round($e - $s, 3))") s  $(grep -h 'timing' $DIR/err.txt | tr '\n' ' ' | cut -c1-420)  md5 $(md5sum < $DIR/o.hist | cut -c1-8)"
  done
  for rep in 1 2; do
    sleep 5
    s=$(date +%s.%N)
    MFX_LOAD_FACTOR=$lf MFX_CLI_TIMING=2 $CMD 2> $DIR/err.txt
    e=$(date +%s.%N)
    echo "lf $lf spaced rep $rep wall $(python3 -c "print(round($e - $s, 3))") s  $(grep -h 'timing' $DIR/err.txt | tr '\n' ' ' | cut -c1-420)"
  done
done
} > $OUT/r05_e2e_lf_ab.txt 2>&1
# PMC rows of the build kernels (separate passes)
for ctr in FETCH_SIZE WRITE_SIZE; do
  sleep 5
  D=/tmp/pmc_build_$ctr
  ( cd /tmp && timeout 900 rocprofv3 --pmc $ctr --kernel-include-regex "mfx_table_add_delta_kernel|mfx_count_kernel" --output-format csv -d $D -o pmc -- $CMD ) > $OUT/r05_pmc_build_$ctr.log 2>&1
  F=$(find $D -name "*counter_collection.csv" | head -1)
  [ -n "$F" ] && python - "$F" $ctr > $OUT/r05_build_pmc_$ctr.txt <<'PY'
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
rm -rf $DIR
sleep 5
MFX_BENCH_KEEP_PMC=$OUT/r05_pmc python bench.py --steps 20 --warmup 5 > $OUT/r05_bench_first.json 2> $OUT/r05_bench_first.log
tail -c 600 $OUT/r05_bench_first.log
