#!/bin/bash
# HBM traffic + instruction counters of the -hist kernel at a given k on the sequence-only index of one random sequence
# (tools/hist_rates_by_k.py): separate rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE cannot share one), kernel trace in its own.
#   tools/pmc_hist_k.sh <bases> <k> <outprefix>        e.g. tools/pmc_hist_k.sh 3e9 31 gpurun_out/r04_k31
set -u
BASES=$1; K=$2; OUT=$(realpath -m $3)
cd "$(dirname "$0")/.."
ROOT=$PWD
export TMPDIR=/tmp
if [ "$K" -le 31 ]; then export MFX_RATES_KINDS=seq; fi
RE=${PMC_KERNEL_RE:-hist_kernel}                 # k > 31: the 128-bit kernels are mfx_w_hist_kernel
mkdir -p "$(dirname "$OUT")"
for CTR in FETCH_SIZE WRITE_SIZE SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES; do
  D=/tmp/pmc_$$_$CTR
  ( cd /tmp && timeout 900 rocprofv3 --pmc $CTR --kernel-include-regex $RE --output-format csv -d $D -o pmc -- python $ROOT/tools/hist_rates_by_k.py $BASES $K ) > ${OUT}_pmc_$CTR.log 2>&1 || true
  F=$(find $D -name '*counter_collection.csv' | head -1)
  if [ -n "$F" ]; then cp "$F" ${OUT}_pmc_$CTR.csv; fi
  rm -rf $D
done
D=/tmp/kt_$$
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $D -o kt -- python $ROOT/tools/hist_rates_by_k.py $BASES $K ) > ${OUT}_kernel_trace.log 2>&1 || true
F=$(find $D -name '*kernel_stats.csv' | head -1)
if [ -n "$F" ]; then cp "$F" ${OUT}_kernel_stats.csv; fi
rm -rf $D
python - "$OUT" "$BASES" "$K" <<'PY'
import csv, glob, sys
out, bases, k = sys.argv[1], float(sys.argv[2]), int(sys.argv[3])
vals = {}
for f in glob.glob(out + "_pmc_*.csv"):
    for row in csv.DictReader(open(f)):
        if "hist_kernel" in row.get("Kernel_Name", ""):
            vals.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
n = bases - k + 1
for c, v in sorted(vals.items()):
    a = sum(v) / len(v)
    extra = ""
    if c == "FETCH_SIZE":
        extra = "  -> x2 (gfx950 counts a 128 B fabric read as 64 B) = %.2f GB per launch = %.3f lines per k-mer" % (2 * a * 1024 / 1e9, 2 * a * 1024 / 128 / n)
    if c == "WRITE_SIZE":
        extra = "  = %.3f GB per launch" % (a * 1024 / 1e9)
    if c == "SQ_INSTS_VALU":
        extra = "  = %.1f wave-VALU instructions per k-mer and lane" % (a * 64 / n)
    print("%-18s mean over %d launches: %.4g%s" % (c, len(v), a, extra))
PY
