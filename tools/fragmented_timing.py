#!/usr/bin/env python3
"""-hist kernel time on the SAME bases cut into contigs of different lengths (a fragmented assembly changes contig on
every tile): python tools/fragmented_timing.py [bases]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import merfin_amd as m
from tools import synth_torch as st

bases = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
ix, seqs, asm, info = st.build_world(m, bases, k=21, lam=26.0, ncontigs=24, device=0)
kp = m.KParams.from_file(26.0, os.path.join(ROOT, "tests", "golden", "example_lookup_table.txt"))
ev = m.Evaluator(ix, kp)
stream = torch.cuda.current_stream().cuda_stream
for clen in (None, 50000, 5000, 1000, 200):
    if clen is None:
        sq, label = seqs, "24 contigs"
    else:
        ptrs, lens = [], []
        for a in asm:
            n = a.numel()
            for o in range(0, n, clen):
                ptrs.append(a.data_ptr() + o)
                lens.append(min(clen, n - o))
        sq = m.Sequences.from_device(ptrs, lens, device=0)
        label = "%d contigs of %d bp" % (len(lens), clen)
    counts = torch.zeros(m.hist_words(ev.nbins, sq.ncontigs), dtype=torch.int64, device="cuda")
    kover = torch.zeros(1, dtype=torch.float64, device="cuda")
    best = 1e9
    for it in range(4):
        counts.zero_(); kover.zero_()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); ev.hist_launch(sq, 0, sq.ntiles, counts, kover, stream=stream); e1.record()
        torch.cuda.synchronize()
        if it:
            best = min(best, e0.elapsed_time(e1))
    kasm = int(counts[2 * ev.nbins].item())
    print("%-32s %8d tiles  %7.2f ms  %6.1f G k-mers/s (%d k-mers)" % (label, sq.ntiles, best, kasm / best / 1e6, kasm), flush=True)
