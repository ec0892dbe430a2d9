#!/bin/bash
# round 5, second GPU call: the streamed multi-slot tests, the staged database load (tests + 3 Gb end-to-end A/B)
set -u
cd "$(dirname "$0")/.."
ROOT=$PWD
OUT=$ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
( timeout 2400 python -m pytest tests/test_gpu_streamed_multi.py tests/test_gpu_bench_launch.py "tests/test_gpu_seqonly.py" tests/test_cli.py tests/test_golden.py -x -q 2>&1 | tail -25 ) > $OUT/r05_second_tests.txt
DIR=/dev/shm/mfx_r05_$$
python - "$DIR" <<'PY' > $OUT/r05_inputs2.log 2>&1
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
echo "# merfin -hist at 3 Gb (5.9 G-k-mer delta-coded read database, 14.15 GB): staged database load (default) against MFX_DB_STAGE=0; table load factor 0.4 (CLI default) / 0.18; 4 runs BACK TO BACK then 2 spaced by 5 s"
for spec in "MFX_X=1" "MFX_DB_STAGE=0" "MFX_LOAD_FACTOR=0.3" "MFX_LOAD_FACTOR=0.18" "MFX_X=2"; do
  sleep 8
  for rep in 1 2 3 4 5 6; do
    [ $rep -ge 5 ] && sleep 5
    s=$(date +%s.%N)
    env $spec MFX_CLI_TIMING=2 MFX_INGEST_TIMING=1 $CMD 2> $DIR/err.txt
    e=$(date +%s.%N)
    echo "$spec $([ $rep -ge 5 ] && echo spaced || echo b2b) rep $rep wall $(python3 -c "print(round($e - $s, 3))") s  $(grep -h 'timing' $DIR/err.txt | tr '\n' ' ' | cut -c1-420)  md5 $(md5sum < $DIR/o.hist | cut -c1-8)"
    grep -h 'staged build\|-- ingest' $DIR/err.txt | head -2 | cut -c1-400 | sed 's/^/      /'
  done
done
} > $OUT/r05_e2e_staged.txt 2>&1
rm -rf $DIR
