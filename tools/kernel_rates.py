#!/usr/bin/env python3
"""Runs every kernel family once on a 1 Gb world so that `rocprofv3 --kernel-trace --stats` can time them:
   -hist, -dump values, -completeness, the assembly counter, table inserts.   python tools/kernel_rates.py [bases]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import merfin_amd as m
from tools import synth_torch as st

bases = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
ix, seqs, asm, info = st.build_world(m, bases, k=21, lam=26.0, ncontigs=4)
ev = m.Evaluator(ix, m.KParams.from_file(26.0, os.path.join(ROOT, "tests", "golden", "example_lookup_table.txt")))
r = ev.hist(seqs)
n0 = int(asm[0].numel())
CH = 1 << 24
for o in range(0, min(n0, 8 * CH), CH):
    ev.dump_values(seqs, 0, o, min(n0, o + CH))
ev.completeness_pieces()
print("world: %d bases, %d distinct k-mers, table %.1f GB; kasm %d; contig 0: %d positions dumped in chunks of %d"
      % (bases, info["distinct"], info["bytes"] / 1e9, r.kasm, min(n0, 8 * CH), CH))
