#!/bin/bash
# tools/pmc_valu.sh <out> <k> <label>=<lib|default> ...: wave-level VALU / SALU / LDS instructions per launch of mfx_hist_kernel (1 Gb bench world)
OUT=$1; shift; K=$1; shift
cd /tmp; export TMPDIR=/tmp
for spec in "$@"; do
  label=${spec%%=*}; lib=${spec#*=}
  if [ "$lib" = "default" ]; then unset MFX_LIB; else export MFX_LIB=$GRAFT_REPO_ROOT/$lib; fi
  for ctrs in "SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_INSTS_VMEM_WR"; do
    d=/tmp/pmcv_$label; rm -rf $d
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
    print("%s %s %.5g" % (label, k, sum(v) / len(v)))
PY
  done
done
