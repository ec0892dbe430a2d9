#!/usr/bin/env python3
"""Fixed cost of one mfx_hist_launch (kernel + the two koverCpy sum kernels) vs the number of tiles: python tools/launch_overhead.py [bases]"""
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
counts = torch.zeros(m.hist_words(ev.nbins, seqs.ncontigs), dtype=torch.int64, device="cuda")
kover = torch.zeros(1, dtype=torch.float64, device="cuda")
stream = torch.cuda.current_stream().cuda_stream
T = seqs.ntiles
for nt in (1, 256, 1024, 4096, 16384, 65536, T):
    nt = min(nt, T)
    best = 1e9
    for it in range(6):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); ev.hist_launch(seqs, 0, nt, counts, kover, stream=stream); e1.record()
        torch.cuda.synchronize()
        if it:
            best = min(best, e0.elapsed_time(e1))
    print("%8d tiles  %8.3f ms   %6.1f us/1k tiles" % (nt, best, best * 1e3 / (nt / 1000.0)), flush=True)
