#!/usr/bin/env python3
"""Streamed -hist (SURVEY 8(d)'s evaluate phase) under every placement of the encoder threads and a few thread counts, one world.
   python tools/stream_place.py [bases]"""
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
ix, seqs, asm, info = st.build_world(m, bases, k=21, lam=26.0, ncontigs=24, seq_only=True)
ev = m.Evaluator(ix, m.KParams.from_file(26.0, os.path.join(ROOT, "tests", "golden", "example_lookup_table.txt")))
ref = ev.hist(seqs)
t = []
for _ in range(5):
    t0 = time.perf_counter(); ev.hist(seqs); t.append(time.perf_counter() - t0)
print("resident: %.2f ms = %.1f G k-mers/s" % (min(t) * 1e3, ref.kasm / min(t) / 1e9), flush=True)
lens = [int(a.numel()) for a in asm]
pins = [m.PinnedBuffer(n) for n in lens]
for pb, a in zip(pins, asm):
    torch.from_numpy(pb.array).copy_(a)
torch.cuda.synchronize()
pinned = [pb.array for pb in pins]
pageable = [np.array(x) for x in pinned]
s2 = m.Sequences.create(lens)
configs = sys.argv[2:] or ["node:16", "os:16", "spread:16", "all:16", "spread:15", "spread:24", "os:24", "spread:32"]
for cfg in configs:
    place, thr = cfg.split(":")
    os.environ["MFX_PACK_PLACE"] = place
    os.environ["MFX_HOST_THREADS"] = thr
    for name, bufs in (("pageable", pageable), ("pinned", pinned)):
        for _ in range(2):
            ev.hist_streamed(s2, bufs)
        tt = []
        for _ in range(7):
            t0 = time.perf_counter(); r = ev.hist_streamed(s2, bufs); tt.append(time.perf_counter() - t0)
        ok = r.kasm == ref.kasm and r.kmissing == ref.kmissing and r.koverCpy == ref.koverCpy
        tt.sort()
        print("%-7s %2s threads %-8s source: best %.2f median %.2f worst %.2f ms = %.1f G k-mers/s (%.2f of resident) %s" %
              (place, thr, name, tt[0] * 1e3, tt[len(tt) // 2] * 1e3, tt[-1] * 1e3, ref.kasm / tt[0] / 1e9, min(t) / tt[0], "ok" if ok else "DIFFERS"), flush=True)
os.environ["MFX_STREAM_TIMING"] = "1"
for cfg in configs[:4]:
    place, thr = cfg.split(":")
    os.environ["MFX_PACK_PLACE"] = place
    os.environ["MFX_HOST_THREADS"] = thr
    ev.hist_streamed(s2, pageable)
