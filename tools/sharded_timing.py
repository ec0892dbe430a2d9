#!/usr/bin/env python3
"""Per-GPU compute time of the SHARDED-index -hist (BASELINE config 5 shape) measured on one GPU.
World of N virtual ranks; this process plays rank 0: it holds shard 0 of the index (the k-mers whose minimizer
hashes to rank 0) and the whole assembly.  Timed:
  route   : extract + label + group by owner (mfx_route_kernel + radix sort + gather) of rank 0's 1/N of the tiles
  evaluate: probe + K* + bin (mfx_hist_keys_kernel) of the k-mers rank 0 OWNS -- here collected by routing every
            tile (what the other ranks would send it), i.e. ~1/N of all k-mers
The exchange itself (12 B per k-mer over xGMI) needs N GPUs and is only estimated.
  python tools/sharded_timing.py [bases] [N]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import merfin_amd as m
from merfin_amd import distributed as D
from tools import synth_torch as st

bases = int(float(sys.argv[1])) if len(sys.argv) > 1 else 3_000_000_000
N = int(sys.argv[2]) if len(sys.argv) > 2 else 8
k, lam = 21, 26.0
dev = "cuda:0"
torch.cuda.set_device(0)
sizes = st.contig_sizes(bases, 24)
truth, layout = st.make_truth(sizes, st.SEED, dev)
asm = st.make_assembly(truth, layout, st.SEED)
cap = int((bases * 2.03 + 1024) / N * 1.15) + 1024
ix = m.Index(k, cap)
ix.set_shard(0, N)
t0 = time.time()
st.add_reads_from_truth(ix, truth, k, lam, st.SEED)
del truth
st.add_error_kmers(ix, bases, k, st.SEED)
seqs = m.Sequences.from_device([a.data_ptr() for a in asm], [a.numel() for a in asm], device=0)
ix.count_asm(seqs)
torch.cuda.synchronize()
info = ix.info()
print("shard 0 of %d: %d k-mers, %.1f GB table, built in %.1fs" % (N, info["distinct"], info["bytes"] / 1e9, time.time() - t0), flush=True)
kp = m.KParams.from_file(lam, os.path.join(ROOT, "tests", "golden", "example_lookup_table.txt"))
ev = m.Evaluator(ix, kp)
T = seqs.ntiles
per = 16384
router = m.Router(ix, N, per)
counts = torch.zeros(m.hist_words(ev.nbins, seqs.ncontigs), dtype=torch.int64, device="cuda")
kover = torch.zeros(1, dtype=torch.float64, device="cuda")
keys = torch.empty(per * m.TILE, dtype=torch.int64, device="cuda")
ctg = torch.empty(per * m.TILE, dtype=torch.int32, device="cuda")
stream = torch.cuda.current_stream().cuda_stream
mine = []                     # k-mers owned by rank 0, as they would arrive
t_route_all = 0.0
n_routed = 0
for tb in range(0, T, per):
    te = min(T, tb + per)
    torch.cuda.synchronize()
    t = time.perf_counter()
    send = router.route(seqs, tb, te, ev.nbins, counts, keys, ctg, stream=stream)
    torch.cuda.synchronize()
    t_route_all += time.perf_counter() - t
    n_routed += int(send.sum())
    n0 = int(send[0])
    mine.append((keys[:n0].clone(), ctg[:n0].clone()))
lo, hi = D.shard(T, 0, N)
n_mine = sum(int(a.numel()) for a, _ in mine)
print("routed %d k-mers of %d tiles in %.1f ms => a rank's 1/%d share: %.1f ms (%.1f G k-mers/s)" %
      (n_routed, T, t_route_all * 1e3, N, t_route_all * 1e3 / N, n_routed / t_route_all / 1e9), flush=True)
best = None
for rep in range(3):
    counts.zero_(); kover.zero_()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for a, c in mine:
        if a.numel():
            ev.hist_keys_launch(a, c, a.numel(), seqs.ncontigs, counts, kover, stream=stream)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    best = dt if best is None else min(best, dt)
print("rank 0 owns %d of them (%.1f %%): evaluated in %.1f ms (%.1f G k-mers/s)" % (n_mine, 100.0 * n_mine / n_routed, best * 1e3, n_mine / best / 1e9), flush=True)
out_bytes = (n_routed / N) * (N - 1) / N * 12
print("exchange (estimate): %.2f GB leave each GPU; at 7 x 153 GB/s peak xGMI egress >= %.1f ms" % (out_bytes / 1e9, out_bytes / (7 * 153e9) * 1e3))
per_rank = t_route_all / N + best
print("compute per rank and whole-genome pass: %.1f ms => %.1f G k-mers/s aggregate on %d GPUs before the exchange" %
      (per_rank * 1e3, n_routed / per_rank / 1e9, N))
