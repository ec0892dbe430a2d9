"""Per-call host overhead of the one-process N-slot -hist (mfx_hist_run_multi): wall time of a call with N slots sharing
ONE GPU against the sum of its N kernel launches timed alone (HIP events around mfx_hist_launch_cyclic on each slot's share).
On 8 real GPUs the kernel of a slot is ~4 ms at 3 Gb, so whatever the call adds per slot (streams, allocations, pinned
mirrors -- all owned by the evaluators now) has to stay far below that.
  python tools/multi_overhead.py [bases=256e6] [slots=8]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch

import merfin_amd as m
from tools import synth_torch as st

bases = int(float(sys.argv[1])) if len(sys.argv) > 1 else 256_000_000
N = int(sys.argv[2]) if len(sys.argv) > 2 else 8
k, lam = 21, 26.0
ix, seqs, asm, info = st.build_world(m, bases, k=k, lam=lam, ncontigs=24, seq_only=True)
kp = m.KParams.from_file(lam, os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "example_lookup_table.txt"))
evs = [m.Evaluator(ix, kp) for _ in range(N)]
words = m.hist_words(evs[0].nbins, seqs.ncontigs)
counts = torch.zeros(words, dtype=torch.int64, device="cuda")
kover = torch.zeros(1, dtype=torch.float64, device="cuda")
s = torch.cuda.current_stream().cuda_stream


def kernels_alone():
    tot = 0.0
    for d in range(N):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        evs[d].hist_launch_cyclic(seqs, d, N, counts, kover, stream=s)
        e1.record()
        torch.cuda.synchronize()
        tot += e0.elapsed_time(e1)
    return tot


for _ in range(3):
    m.hist_multi(evs, [seqs] * N)
    kernels_alone()
reps = 20
t0 = time.perf_counter()
for _ in range(reps):
    res = m.hist_multi(evs, [seqs] * N)
t_multi = (time.perf_counter() - t0) / reps * 1e3
t_k = sum(kernels_alone() for _ in range(reps)) / reps
one = m.Evaluator(ix, kp)
one.hist(seqs)
t0 = time.perf_counter()
for _ in range(reps):
    r1 = one.hist(seqs)
t_one = (time.perf_counter() - t0) / reps * 1e3
assert r1.kmissing == res.kmissing and r1.kasm == res.kasm
print("%d bases, %d slots on one GPU: mfx_hist_run_multi %.3f ms per call; its %d launches timed alone (events) sum to %.3f ms; "
      "difference %.3f ms = %.3f ms per slot.  mfx_hist_run (one slot): %.3f ms per call."
      % (bases, N, t_multi, N, t_k, t_multi - t_k, (t_multi - t_k) / N, t_one))
