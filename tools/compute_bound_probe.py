#!/usr/bin/env python3
"""How fast is the -hist kernel when HBM is NOT the limit?  A small genome (its table fits the 256 MB Infinity Cache)
evaluated many times over in one launch (the same contigs listed again and again).  python tools/compute_bound_probe.py [bases] [repeats]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import merfin_amd as m
from tools import synth_torch as st

bases = int(float(sys.argv[1])) if len(sys.argv) > 1 else 3_000_000
rep = int(sys.argv[2]) if len(sys.argv) > 2 else 300
ix, seqs, asm, info = st.build_world(m, bases, k=21, lam=26.0, ncontigs=24, device=0)
kp = m.KParams.from_file(26.0, os.path.join(ROOT, "tests", "golden", "example_lookup_table.txt"))
ev = m.Evaluator(ix, kp)
ptrs = [a.data_ptr() for a in asm] * rep
lens = [a.numel() for a in asm] * rep
sq = m.Sequences.from_device(ptrs, lens, device=0)
counts = torch.zeros(m.hist_words(ev.nbins, sq.ncontigs), dtype=torch.int64, device="cuda")
kover = torch.zeros(1, dtype=torch.float64, device="cuda")
stream = torch.cuda.current_stream().cuda_stream
best = 1e9
for it in range(4):
    counts.zero_(); kover.zero_()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); ev.hist_launch(sq, 0, sq.ntiles, counts, kover, stream=stream); e1.record()
    torch.cuda.synchronize()
    if it:
        best = min(best, e0.elapsed_time(e1))
kasm = int(counts[2 * ev.nbins].item())
print("table %.2f GB, %d k-mers per launch (genome of %d bp x %d): %.2f ms = %.1f G k-mers/s" %
      (info["bytes"] / 1e9, kasm, bases, rep, best, kasm / best / 1e6), flush=True)
