#!/usr/bin/env python3
"""Phase breakdown of SURVEY 8(d)'s evaluate phase (mfx_hist_run_streamed: first tile H2D start -> histogram on the host) on
the bench workload: what each resource of the pipeline does ALONE, then the pipeline with its per-chunk timeline.
   python tools/stream_phases.py [bases] [index: seq|full]
 (a) encoder only : the host threads pack the whole assembly (mfx_pack_bases, T threads, pinned and pageable sources)
 (b) H2D only     : the packed planes (0.375 B/base) pinned -> device, one stream
 (c) kernel only  : the resident -hist launch
 (d) the pipeline : MFX_STREAM_TIMING=2 timeline of one streamed run (after warm-up), pinned and pageable sources
plus where the threads and the memory are: NUMA node of the source buffers, CPUs allowed."""
import ctypes as C
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import merfin_amd as m
from tools import synth_torch as st

bases = int(float(sys.argv[1])) if len(sys.argv) > 1 else 3_000_000_000
kind = sys.argv[2] if len(sys.argv) > 2 else "seq"
L = m.load_library()
L.mfx_pack_bases.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]

print("CPUs allowed: %d  (%s)" % (len(os.sched_getaffinity(0)), open("/sys/fs/cgroup/cpu.max").read().strip() if os.path.exists("/sys/fs/cgroup/cpu.max") else "no cgroup quota file"))
try:
    nodes = sorted(d for d in os.listdir("/sys/devices/system/node") if d.startswith("node"))
    for nd in nodes:
        print("  %s cpus %s" % (nd, open("/sys/devices/system/node/%s/cpulist" % nd).read().strip()))
except Exception as e:
    print("  (no NUMA topology: %r)" % (e,))

ix, seqs, asm, info = st.build_world(m, bases, k=21, lam=26.0, ncontigs=24, seq_only=kind == "seq")
ev = m.Evaluator(ix, m.KParams.from_file(26.0, os.path.join(ROOT, "tests", "golden", "example_lookup_table.txt")))
ref = ev.hist(seqs)
lens = [int(a.numel()) for a in asm]
pins = [m.PinnedBuffer(n) for n in lens]
for pb, a in zip(pins, asm):
    torch.from_numpy(pb.array).copy_(a)
torch.cuda.synchronize()
pinned = [pb.array for pb in pins]
pageable = [np.array(x) for x in pinned]
total = sum(lens)

# (c) kernel only
t = []
for _ in range(5):
    torch.cuda.synchronize(); t0 = time.perf_counter(); ev.hist(seqs); t.append(time.perf_counter() - t0)
t_res = min(t)
print("(c) resident -hist (launch + reduce + D2H): %.2f ms  = %.1f G k-mers/s" % (t_res * 1e3, ref.kasm / t_res / 1e9))

# (a) encoder only
T = int(os.environ.get("MFX_HOST_THREADS", "0")) or min(len(os.sched_getaffinity(0)), 64)
try:
    q, per = open("/sys/fs/cgroup/cpu.max").read().split()
    if q != "max":
        T = max(1, min(T, int(float(q) / float(per))))
except Exception:
    pass
codes = m.PinnedBuffer(total // 4 + 64)
valid = m.PinnedBuffer(total // 8 + 64)
for name, src in (("pinned", pinned), ("pageable", pageable)):
    big = max(range(len(src)), key=lambda i: lens[i])
    a = src[big]
    n = lens[big] // (32 * T) * (32 * T)
    per = n // T
    def run(tt):
        L.mfx_pack_bases(a.ctypes.data + tt * per, per, codes.array.ctypes.data + tt * per // 4, valid.array.ctypes.data + tt * per // 8)
    best = 1e9
    for rep in range(4):
        th = [threading.Thread(target=run, args=(tt,)) for tt in range(T)]
        t0 = time.perf_counter(); [x.start() for x in th]; [x.join() for x in th]; best = min(best, time.perf_counter() - t0)
    print("(a) encoder only, %s source, %d threads, %d Mb contig: %.1f GB/s of bases  -> %.1f ms for the whole assembly" % (name, T, n // 10**6, n / best / 1e9, total / (n / best) * 1e3))

# (b) H2D only
nb = total * 3 // 8
src = m.PinnedBuffer(nb)
dst = torch.empty(nb, dtype=torch.uint8, device="cuda")
hs = torch.from_numpy(src.array)
for _ in range(2):
    dst.copy_(hs, non_blocking=True)
torch.cuda.synchronize()
t = []
for _ in range(4):
    t0 = time.perf_counter(); dst.copy_(hs, non_blocking=True); torch.cuda.synchronize(); t.append(time.perf_counter() - t0)
print("(b) H2D only, %.3f GB of packed planes, pinned: %.2f ms = %.1f GB/s" % (nb / 1e9, min(t) * 1e3, nb / min(t) / 1e9))
del dst, src, hs

# (d) the pipeline
s2 = m.Sequences.create(lens)
for name, bufs in (("pinned", pinned), ("pageable", pageable)):
    os.environ.pop("MFX_STREAM_TIMING", None)
    for _ in range(2):
        ev.hist_streamed(s2, bufs)
    tt = []
    for _ in range(5):
        t0 = time.perf_counter(); r = ev.hist_streamed(s2, bufs); tt.append(time.perf_counter() - t0)
    ok = r.kasm == ref.kasm and r.kmissing == ref.kmissing and r.koverCpy == ref.koverCpy
    print("(d) streamed, %s source: best %.2f ms median %.2f ms = %.1f G k-mers/s (%.2f of resident); result %s" %
          (name, min(tt) * 1e3, sorted(tt)[len(tt) // 2] * 1e3, ref.kasm / min(tt) / 1e9, t_res / min(tt), "== resident" if ok else "DIFFERS"), flush=True)
    os.environ["MFX_STREAM_TIMING"] = "2"
    sys.stderr.flush()
    ev.hist_streamed(s2, bufs)
    sys.stderr.flush()
os.environ.pop("MFX_STREAM_TIMING", None)
