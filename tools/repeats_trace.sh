#!/bin/bash
# tools/repeats_trace.sh <out file> <k> <levels>: rocprofv3 kernel-trace stats (main kernel / rest kernel split) of tools/repeats_rates.py
OUT=$1; K=$2; LEVELS=$3
cd /tmp; export TMPDIR=/tmp
for L in ${LEVELS//,/ }; do
  d=/tmp/rt_$L; rm -rf $d
  rocprofv3 --kernel-trace --stats --output-format csv -d $d -o rt -- python $GRAFT_REPO_ROOT/tools/repeats_rates.py --bases 3e9 --levels $L --k $K > /tmp/rt_$L.log 2>&1
  echo "== level $L k=$K" >> $GRAFT_REPO_ROOT/$OUT
  grep "^{" /tmp/rt_$L.log | python3 -c "import sys,json; r=json.loads(sys.stdin.readline()); print('rate %.1f G  dbg %r  novf %d' % (r['kmers_per_s']/1e9, r['dbg'], r['novf']))" >> $GRAFT_REPO_ROOT/$OUT
  f=$(find $d -name "*kernel_stats.csv" | head -1)
  python3 - "$f" >> $GRAFT_REPO_ROOT/$OUT <<'PY'
import sys, csv
for row in csv.DictReader(open(sys.argv[1])):
    n = row["Name"]
    if "mfx_hist" in n or "mfx_sum" in n:
        print("  %-60s calls %s avg %.3f ms min %.3f max %.3f" % (n[:60], row["Calls"], float(row["AverageNs"])/1e6, float(row["MinNs"])/1e6, float(row["MaxNs"])/1e6))
PY
done
