#!/bin/bash
# A/B of the compact layout's window count: w = 4 (t = 6, the default) against w = 5 (t = 7) at k = 21 / 3 Gb, PMC included
set -u
cd "$(dirname "$0")/.."
OUT=$PWD/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests/test_gpu_seqonly.py -x -q -k "mod_minimizer_placement_every_k" 2>&1 | tail -5 ) > $OUT/r05_w5_tests.txt
for spec in "MFX_X=1" "MFX_MZ_W=5" "MFX_X=1" "MFX_MZ_W=5"; do
  env $spec python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-e2e --no-k31 --no-full-index --no-streamed 2>$OUT/r05_w5_err.txt | python -c "
import json,sys
d=json.load(sys.stdin); r=d['roofline']
print('$spec', '%.2f G k-mers/s' % (d['value']/1e9), '%.3f ms' % d['ms_per_step'], 'table %.1f GB' % d['config']['index_gb'], 'lines/k-mer %.4f' % (r.get('lines_per_kmer') or 0), 'frac %.3f' % (r.get('frac') or 0), 'VALU/k-mer %.1f' % ((r.get('issue') or {}).get('valu_insts_per_kmer') or 0), 'kmissing', d['config']['kmissing'], 'koverCpy', d['config']['koverCpy'])
"
done > $OUT/r05_w5_ab.txt 2>&1
for lf in 0.25 0.3 0.4; do
for spec in "MFX_X=1" "MFX_MZ_W=5"; do
  env $spec MFX_LOAD_FACTOR=$lf python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-e2e --no-k31 --no-full-index --no-streamed --no-pmc 2>>$OUT/r05_w5_err.txt | python -c "
import json,sys
d=json.load(sys.stdin)
print('$spec lf $lf', '%.2f G k-mers/s' % (d['value']/1e9), '%.3f ms' % d['ms_per_step'], 'table %.1f GB' % d['config']['index_gb'], 'kmissing', d['config']['kmissing'])
"
done; done >> $OUT/r05_w5_ab.txt 2>&1
cat $OUT/r05_w5_ab.txt
