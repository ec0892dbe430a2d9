#!/usr/bin/env python3
"""Repeats the streamed -hist (packed transport, two kernel streams, parked packer threads) on one world and checks that
every run returns the resident launch's result bit for bit.   python tools/stream_soak.py [bases] [runs]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import merfin_amd as m
from tools import synth_torch as st

bases = int(float(sys.argv[1])) if len(sys.argv) > 1 else 512_000_000
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 200
ix, seqs, asm, info = st.build_world(m, bases, k=21, lam=26.0, ncontigs=24)
ev = m.Evaluator(ix, m.KParams.from_file(26.0, os.path.join(ROOT, "tests", "golden", "example_lookup_table.txt")))
ref = ev.hist(seqs)
host = [a.cpu().numpy() for a in asm]
lens = [len(h) for h in host]
s2 = m.Sequences.create(lens)
bad = 0
t0 = time.time()
for i in range(runs):
    r = ev.hist_streamed(s2, host)
    ok = (r.kasm == ref.kasm and r.kmissing == ref.kmissing and r.koverCpy == ref.koverCpy and (r.undr() == ref.undr()).all()
          and (r.over() == ref.over()).all() and (r.contig_kasm() == ref.contig_kasm()).all())
    bad += not ok
    if not ok:
        print("run %d differs: kasm %d/%d kmissing %d/%d kover %r/%r" % (i, r.kasm, ref.kasm, r.kmissing, ref.kmissing, r.koverCpy, ref.koverCpy), flush=True)
dt = time.time() - t0
print("%d streamed runs of %d bases: %d differ; %.1f ms per run = %.1f G k-mers/s" % (runs, bases, bad, dt / runs * 1e3, ref.kasm * runs / dt / 1e9))
sys.exit(1 if bad else 0)
