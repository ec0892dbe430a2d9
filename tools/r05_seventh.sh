#!/bin/bash
# round 5, seventh GPU call: the placed database with its kernel in contiguous rounds
set -u
cd "$(dirname "$0")/.."
ROOT=$PWD
OUT=$ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
( timeout 2400 python -m pytest tests/test_gpu_placed_db.py tests/test_gpu_db.py tests/test_golden.py tests/test_gpu_seqonly.py -x -q 2>&1 | tail -25 ) > $OUT/r05_seventh_tests.txt
tail -3 $OUT/r05_seventh_tests.txt
DIR=/dev/shm/mfx_r05_$$
python - "$DIR" <<'PY' > $OUT/r05_inputs7.log 2>&1
import sys, os
sys.path.insert(0, os.getcwd())
import torch
import merfin_amd as m
from tools import synth_torch as st, e2e_inputs
inp = e2e_inputs.write_inputs(m, st, torch, 3_000_000_000, sys.argv[1], ncontigs=24, k=21, lam=26.0, placed=True)
print("inputs written", inp["db_bytes"], inp["placed_db_bytes"], inp["read_kmers"], inp["write_s"])
PY
CMDA="$ROOT/merfin_amd/bin/merfin -hist -sequence $DIR/asm.fasta -peak 26 -prob $ROOT/tests/golden/example_lookup_table.txt -output $DIR/o.hist"
sleep 10
{
echo "# merfin -hist at 3 Gb, table load factor 0.4 (CLI default): the read database k-mer-SORTED (read.mfxk) against PLACED (read.placed.mfxk, update kernel in contiguous rounds); 4 runs back to back then 2 spaced by 5 s"
tail -1 $OUT/r05_inputs7.log
for db in read.mfxk read.placed.mfxk; do
for spec in "MFX_X=1" "MFX_DB_STAGE=0"; do
  sleep 8
  for rep in 1 2 3 4 5 6; do
    [ $rep -ge 5 ] && sleep 5
    s=$(date +%s.%N)
    env $spec MFX_CLI_TIMING=2 MFX_INGEST_TIMING=1 $CMDA -readmers $DIR/$db 2> $DIR/err.txt
    e=$(date +%s.%N)
    echo "$db $spec $([ $rep -ge 5 ] && echo spaced || echo b2b) rep $rep wall $(python3 -c "print(round($e - $s, 3))") s  $(grep -h 'timing' $DIR/err.txt | tr '\n' ' ' | cut -c1-420)  md5 $(md5sum < $DIR/o.hist | cut -c1-8)"
    grep -h 'staged build\|-- ingest' $DIR/err.txt | head -1 | cut -c1-400 | sed 's/^/      /'
  done
done
done
} > $OUT/r05_e2e_placed2.txt 2>&1
for db in read.mfxk read.placed.mfxk; do
  D=/tmp/kt_$db
  ( cd /tmp && MFX_DB_STAGE=0 MFX_INGEST_STREAMS_ONE=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $D -o kt -- $CMDA -readmers $DIR/$db ) > $OUT/r05_placed_trace_$db.log 2>&1
  F=$(find $D -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && cp "$F" $OUT/r05_placed2_kernel_stats_$db.csv
  F=$(find $D -name "*kernel_trace.csv" | head -1)
  [ -n "$F" ] && python - "$F" $db >> $OUT/r05_placed2_span.txt <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "add_placed" in r["Kernel_Name"] or "add_delta" in r["Kernel_Name"]]
s = [int(r["Start_Timestamp"]) for r in rows]; e = [int(r["End_Timestamp"]) for r in rows]
ev = sorted([(a, 1) for a in s] + [(b, -1) for b in e]); busy = 0; depth = 0; last = 0
for t, dlt in ev:
    if depth > 0: busy += t - last
    depth += dlt; last = t
print("%-20s update launches %d: first start -> last end %.3f s, busy (union) %.3f s" % (sys.argv[2], len(rows), (max(e) - min(s)) / 1e9, busy / 1e9))
PY
  rm -rf $D
  for ctr in FETCH_SIZE WRITE_SIZE; do
    D=/tmp/pmc_$db$ctr
    ( cd /tmp && MFX_DB_STAGE=0 timeout 900 rocprofv3 --pmc $ctr --kernel-include-regex "mfx_table_add_placed_kernel|mfx_table_add_delta_kernel" --output-format csv -d $D -o pmc -- $CMDA -readmers $DIR/$db ) > /dev/null 2>&1
    F=$(find $D -name "*counter_collection.csv" | head -1)
    [ -n "$F" ] && python - "$F" $ctr $db >> $OUT/r05_placed2_pmc.txt <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(sys.argv[1])):
    if r.get("Counter_Name") != sys.argv[2]:
        continue
    n = r["Kernel_Name"].split("(")[0][:60]
    agg[n][0] += 1
    agg[n][1] += float(r["Counter_Value"])
for n, (c, v) in agg.items():
    print("%-20s %-40s launches %5d  %s sum %.6e KiB" % (sys.argv[3], n, c, sys.argv[2], v))
PY
    rm -rf $D
  done
done
rm -rf $DIR
cat $OUT/r05_placed2_span.txt $OUT/r05_placed2_pmc.txt
