#!/usr/bin/env python3
"""What one rank of an N-GPU strong-scaling run does, timed on one GPU: the 3 Gb index is replicated, the
rank evaluates tiles [T*r/N, T*(r+1)/N).  Prints kernel ms per N and MFX_BLOCKS_PER_CU, next to the ideal t1/N (per rank: best of 5 launches; reported: the slowest rank)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import merfin_amd as m
from merfin_amd import distributed as D
from tools import synth_torch as st

bases = int(float(sys.argv[1])) if len(sys.argv) > 1 else 3_000_000_000
ix, seqs, asm, info = st.build_world(m, bases, k=21, lam=26.0, ncontigs=24, device=0)
kp = m.KParams.from_file(26.0, os.path.join(ROOT, "tests", "golden", "example_lookup_table.txt"))
T = seqs.ntiles
stream = torch.cuda.current_stream().cuda_stream
base = None
CYCLIC = bool(int(os.environ.get("CYCLIC", "0")))      # 1: block-cyclic shares (what bench.py / mgpu.py use for N > 1)
for bpc in (os.environ.get("BPCS", "8,16,32,64").split(",")):
    os.environ["MFX_BLOCKS_PER_CU"] = bpc
    ev = m.Evaluator(ix, kp)
    counts = torch.zeros(m.hist_words(ev.nbins, seqs.ncontigs), dtype=torch.int64, device="cuda")
    kover = torch.zeros(1, dtype=torch.float64, device="cuda")
    row = []
    for N in (1, 2, 4, 8):
        worst = 0.0                                      # slowest RANK; each rank's time = best of 5 launches
        for r in range(N):
            lo, hi = D.shard(T, r, N)
            best = 1e9
            for it in range(6):
                e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
                e0.record()
                if CYCLIC:
                    ev.hist_launch_cyclic(seqs, r, N, counts, kover, stream=stream)
                else:
                    ev.hist_launch(seqs, lo, hi, counts, kover, stream=stream)
                e1.record()
                torch.cuda.synchronize()
                if it:
                    best = min(best, e0.elapsed_time(e1))
            worst = max(worst, best)
        row.append(worst)
    if base is None:
        base = row[0]
    print(("cyclic   " if CYCLIC else "contiguous ") + "blocks/CU %3s: " % bpc + "  ".join("N=%d %.2f ms (x%.2f of ideal)" % (N, t, t / (row[0] / N)) for N, t in zip((1, 2, 4, 8), row)), flush=True)
