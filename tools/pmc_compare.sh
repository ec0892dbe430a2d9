#!/bin/bash
# tools/pmc_compare.sh <out file> <k> <lib label>=<lib path|default> ...  -- per-launch PMC counters of mfx_hist_kernel on the 1 Gb bench world for
# several builds of the library (MFX_LIB), one rocprofv3 pass per counter group (counters only: no trace domains)
OUT=$1; shift; K=$1; shift
cd /tmp; export TMPDIR=/tmp
for spec in "$@"; do
  label=${spec%%=*}; lib=${spec#*=}
  if [ "$lib" = "default" ]; then unset MFX_LIB; else export MFX_LIB=$GRAFT_REPO_ROOT/$lib; fi
  for ctrs in "SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "FETCH_SIZE" "WRITE_SIZE"; do
    d=/tmp/pmc_$label; rm -rf $d
    rocprofv3 --pmc $ctrs --kernel-include-regex mfx_hist_kernel --output-format csv -d $d -o pmc -- python $GRAFT_REPO_ROOT/bench.py --pmc-child --bases 1e9 --index seq --pmc-k $K > /dev/null 2>&1
    python3 - "$label" $d <<'PY' >> $GRAFT_REPO_ROOT/$OUT
import sys, glob, csv, collections
label, d = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list)
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "mfx_hist_kernel" in row.get("Kernel_Name", ""):
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, v in acc.items():
    print("%s %s %.6g (mean of %d launches)" % (label, k, sum(v) / len(v), len(v)))
PY
  done
done
