#!/bin/bash
# round 5: kernel trace of one `merfin -hist` process at 3 Gb from the placed database (final source: the placed update scans its k-mers for the offset bucket)
set -u
cd "$(dirname "$0")/.."
ROOT=$PWD
OUT=$ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
DIR=/dev/shm/mfx_r05_$$
python - "$DIR" <<'PY' > $OUT/r05_inputs12.log 2>&1
import sys, os
sys.path.insert(0, os.getcwd())
import torch
import merfin_amd as m
from tools import synth_torch as st, e2e_inputs
inp = e2e_inputs.write_inputs(m, st, torch, 3_000_000_000, sys.argv[1], ncontigs=24, k=21, lam=26.0, placed=True)
print("inputs written", inp["db_bytes"], inp["placed_db_bytes"], inp["read_kmers"], inp["write_s"])
PY
sleep 6
for db in read.placed.mfxk read.mfxk; do
D=/tmp/kt_e2e_$db
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $D -o kt -- $ROOT/merfin_amd/bin/merfin -hist -sequence $DIR/asm.fasta -peak 26 -prob $ROOT/tests/golden/example_lookup_table.txt -output $DIR/o.hist -readmers $DIR/$db ) > $OUT/r05_e2e_trace_$db.log 2>&1
F=$(find $D -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && cp "$F" $OUT/r05_e2e_kernel_stats_$db.csv
rm -rf $D
echo "== $db"; grep -i "add_placed\|add_delta\|count_kernel\|hist_kernel\|table_init\|Name" $OUT/r05_e2e_kernel_stats_$db.csv | cut -c1-60,100-220
sleep 5
done
rm -rf $DIR
