#!/bin/bash
# kernel + memory-copy trace of one `merfin -hist` process on bench-shaped inputs (tools/e2e_inputs.py):
#   tools/e2e_trace.sh <bases> <outprefix>      ->  <outprefix>_kernel_stats.csv, _memory_copy_stats.csv, _run.log
set -u
BASES=$1; OUT=$(realpath -m $2)
cd "$(dirname "$0")/.."
ROOT=$PWD
export TMPDIR=/tmp
DIR=/dev/shm/mfx_e2etrace_$$
python - "$BASES" "$DIR" <<'PY'
import sys, os
sys.path.insert(0, os.getcwd())
import torch
import merfin_amd as m
from tools import synth_torch as st, e2e_inputs
inp = e2e_inputs.write_inputs(m, st, torch, int(float(sys.argv[1])), sys.argv[2], ncontigs=24, k=21, lam=26.0)
print("inputs written", inp["db_bytes"])
PY
sleep 5
CMD="$ROOT/merfin_amd/bin/merfin -hist -sequence $DIR/asm.fasta -readmers $DIR/read.mfxk -peak 26 -prob $ROOT/tests/golden/example_lookup_table.txt -output $DIR/o.hist"
MFX_CLI_TIMING=2 MFX_INGEST_TIMING=1 $CMD > ${OUT}_run.log 2>&1            # un-traced, for the phase times
sleep 8
D=/tmp/e2ekt_$$
( cd /tmp && MFX_CLI_TIMING=2 MFX_INGEST_TIMING=1 timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $D -o kt -- $CMD ) > ${OUT}_trace.log 2>&1
for f in kernel_stats memory_copy_stats; do F=$(find $D -name "*${f}.csv" | head -1); [ -n "$F" ] && cp "$F" ${OUT}_${f}.csv; done
# when did the copies and the insert kernels run?  (first start, last end, busy time)
python - "$D" <<'PY'
import csv, glob, sys
d = sys.argv[1]
def spans(pattern, namecol, startcol, endcol, pick):
    for f in glob.glob(d + "/**/*" + pattern, recursive=True):
        rows = [r for r in csv.DictReader(open(f)) if pick(r)]
        if not rows:
            continue
        s = [int(r[startcol]) for r in rows]; e = [int(r[endcol]) for r in rows]
        busy = sum(b - a for a, b in zip(s, e))
        print("%s: %d rows, first start -> last end %.3f s, busy %.3f s" % (pattern, len(rows), (max(e) - min(s)) / 1e9, busy / 1e9))
        return min(s), max(e)
k = spans("kernel_trace.csv", "Kernel_Name", "Start_Timestamp", "End_Timestamp", lambda r: "add_delta" in r.get("Kernel_Name", ""))
c = spans("memory_copy_trace.csv", "Name", "Start_Timestamp", "End_Timestamp", lambda r: "HOST_TO_DEVICE" in (r.get("Direction", "") + r.get("Name", "")).upper())
n = spans("kernel_trace.csv", "Kernel_Name", "Start_Timestamp", "End_Timestamp", lambda r: "mfx_count_kernel" in r.get("Kernel_Name", ""))
if k and c and n:
    print("count kernel %.3f s; first H2D starts %.3f s after the count kernel's start; first insert %.3f s after it; copies end %.3f s, inserts end %.3f s after it" %
          ((n[1] - n[0]) / 1e9, (c[0] - n[0]) / 1e9, (k[0] - n[0]) / 1e9, (c[1] - n[0]) / 1e9, (k[1] - n[0]) / 1e9))
PY
grep -h "timing\|ingest" ${OUT}_run.log ${OUT}_trace.log | cut -c1-300
rm -rf $D $DIR
