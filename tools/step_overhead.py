#!/usr/bin/env python3
"""Fixed cost of ONE bench.py step at a 1/N block-cyclic share, measured on one GPU (the 8-GPU run's per-rank step is this
plus whatever RCCL adds over xGMI): the step exactly as bench.py --gpus N runs it -- clear the counts image, launch the
rank's share, all-reduce through the library's RCCL communicator (a world of ONE here: the collective calls run and are the
identity), copy the image to the host -- against the kernel's own time (HIP events).
   python tools/step_overhead.py [bases=3e9] [N=8] [steps=200]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import merfin_amd as m
from tools import synth_torch as st

bases = int(float(sys.argv[1])) if len(sys.argv) > 1 else 3_000_000_000
N = int(sys.argv[2]) if len(sys.argv) > 2 else 8
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 200
ix, seqs, asm, info = st.build_world(m, bases, k=21, lam=26.0, ncontigs=24, seq_only=True)
del asm
ev = m.Evaluator(ix, m.KParams.from_file(26.0, os.path.join(ROOT, "tests", "golden", "example_lookup_table.txt")))
words = m.hist_words(ev.nbins, seqs.ncontigs)
counts = torch.zeros(words, dtype=torch.int64, device="cuda")
kover = torch.zeros(1, dtype=torch.float64, device="cuda")
h_counts = torch.zeros(words, dtype=torch.int64).pin_memory()
h_kover = torch.zeros(1, dtype=torch.float64).pin_memory()
stream = torch.cuda.current_stream().cuda_stream
comm = m.Comm(m.Comm.unique_id(), 0, 1, device=0)

for rank in sorted({0, N - 1}):
    pairs = []

    def step(timed, reduce=True, d2h=True, clear=True):
        if clear:
            counts.zero_(); kover.zero_()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        if N > 1:
            ev.hist_launch_cyclic(seqs, rank, N, counts, kover, block_tiles=256, stream=stream)
        else:
            ev.hist_launch(seqs, 0, seqs.ntiles, counts, kover, stream=stream)
        e1.record()
        if reduce:
            comm.hist_allreduce(ev, counts, kover, seqs.ncontigs, stream=stream)
        if d2h:
            h_counts.copy_(counts, non_blocking=True); h_kover.copy_(kover, non_blocking=True)
        if timed:
            pairs.append((e0, e1))

    for what, kw in (("full step (clear + launch + all-reduce + D2H)", {}), ("without the all-reduce", {"reduce": False}),
                     ("launch + D2H only", {"reduce": False, "clear": False}), ("launch only", {"reduce": False, "clear": False, "d2h": False})):
        pairs.clear()
        for _ in range(10):
            step(False, **kw)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step(True, **kw)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        km = sum(a.elapsed_time(b) for a, b in pairs) / len(pairs)
        print("rank %d of %d, %-48s: %.3f ms per step, kernels (hist + the two koverCpy sums) %.3f ms, fixed cost %.3f ms" % (rank, N, what, dt * 1e3, km, dt * 1e3 - km), flush=True)
    # one step at a time (synchronised): the latency view
    lat = []
    for _ in range(50):
        torch.cuda.synchronize(); t0 = time.perf_counter(); step(False); torch.cuda.synchronize(); lat.append(time.perf_counter() - t0)
    print("rank %d of %d, one synchronised step: median %.3f ms" % (rank, N, sorted(lat)[len(lat) // 2] * 1e3), flush=True)
comm.close()
