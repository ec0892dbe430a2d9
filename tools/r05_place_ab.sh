#!/bin/bash
# A/B of layout version 9's placement pieces (line from the bijective mix, mini-bucket from the window) against layout 8's
set -u
cd "$(dirname "$0")/.."
ROOT=$PWD
OUT=$ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
one() {
  local label=$1 lib=$2; shift 2
  if [ "$lib" = "default" ]; then unset MFX_LIB; else export MFX_LIB=$ROOT/$lib; fi
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-e2e --no-k31 --no-full-index --no-streamed "$@" 2>>$OUT/r05_place_ab_err.txt | python -c "
import json,sys
d=json.load(sys.stdin); r=d['roofline']
print('%-44s' % '$label', '%.2f G k-mers/s' % (d['value']/1e9), '%.3f ms' % d['ms_per_step'], 'lines/k-mer %.4f' % (r.get('lines_per_kmer') or 0), 'VALU/k-mer %.1f' % ((r.get('issue') or {}).get('valu_insts_per_kmer') or 0), 'kmissing', d['config']['kmissing'], 'koverCpy %.7f' % d['config']['koverCpy'])
"
  unset MFX_LIB
}
{
one "layout 9 (mix line, window bucket)" default
one "layout-8 line, window bucket" tools/_build/ab/lib_oldline.so
one "mix line, offset bucket" tools/_build/ab/lib_xbucket.so
one "layout-8 line, offset bucket (= layout 8)" tools/_build/ab/lib_oldboth.so
one "layout 9 (mix line, window bucket)" default --no-pmc
one "layout-8 line, offset bucket (= layout 8)" tools/_build/ab/lib_oldboth.so --no-pmc
} > $OUT/r05_place_ab.txt 2>&1
cat $OUT/r05_place_ab.txt
python - <<'PY' 2>/dev/null | tail -8
import os, sys
sys.path.insert(0, os.getcwd())
import torch, numpy as np
import merfin_amd as m
from tools import synth_torch as st
for lib in (None,):
    ix, seqs, asm, info = st.build_world(m, 1_000_000_000, k=21, lam=26.0, ncontigs=24, seq_only=True)
    ev = m.Evaluator(ix, m.KParams.from_file(26.0, "tests/golden/example_lookup_table.txt"))
    ev.debug(True); r = ev.hist(seqs); c = ev.debug_counters(); ev.debug(False)
    print("1 Gb debug counters (layout 9):", c, "kasm", r.kasm)
PY
