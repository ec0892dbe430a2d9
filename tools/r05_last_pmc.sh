#!/bin/bash
# round 5, last call: HBM bytes (rocprofv3 --pmc, separate passes) of the build's kernels in one `merfin -hist` process at 3 Gb from the placed database, final source
set -u
cd "$(dirname "$0")/.."
ROOT=$PWD
OUT=$ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
DIR=/dev/shm/mfx_r05_$$
python - "$DIR" <<'PY' > $OUT/r05_inputs13.log 2>&1
import sys, os
sys.path.insert(0, os.getcwd())
import torch
import merfin_amd as m
from tools import synth_torch as st, e2e_inputs
inp = e2e_inputs.write_inputs(m, st, torch, 3_000_000_000, sys.argv[1], ncontigs=24, k=21, lam=26.0, placed=True)
print("inputs written", inp["db_bytes"], inp["placed_db_bytes"], inp["read_kmers"], inp["write_s"])
PY
sleep 5
: > $OUT/r05_build_kernels_pmc_final.txt
for C in FETCH_SIZE WRITE_SIZE; do
D=/tmp/pmc_$C
( cd /tmp && MFX_DB_STAGE=0 timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $D -o p -- $ROOT/merfin_amd/bin/merfin -hist -sequence $DIR/asm.fasta -peak 26 -prob $ROOT/tests/golden/example_lookup_table.txt -output $DIR/o.hist -readmers $DIR/read.placed.mfxk ) > $OUT/r05_last_pmc_$C.log 2>&1
F=$(find $D -name "*counter_collection.csv" | head -1)
python3 - "$F" "$C" >> $OUT/r05_build_kernels_pmc_final.txt <<'PY'
import csv, sys, collections
f, c = sys.argv[1], sys.argv[2]
s = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(f)):
    if r.get("Counter_Name") != c: continue
    k = r["Kernel_Name"].split("(")[0][:60]
    s[k][0] += 1; s[k][1] += float(r["Counter_Value"])
for k, (n, v) in sorted(s.items(), key=lambda kv: -kv[1][1])[:6]:
    print("%-62s launches %5d  %s sum %.6e KiB%s" % (k, n, c, v, "  (x2 on gfx950: %.1f GB)" % (v * 2 * 1024 / 1e9) if c == "FETCH_SIZE" else "  (%.1f GB)" % (v * 1024 / 1e9)))
PY
rm -rf $D
sleep 4
done
rm -rf $DIR
cat $OUT/r05_build_kernels_pmc_final.txt
