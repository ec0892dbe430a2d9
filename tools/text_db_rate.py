#!/usr/bin/env python3
"""Rate of the `meryl print` text ingest (uncompressed: parsed in pieces by the host threads).
   python tools/text_db_rate.py [million_lines]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import merfin_amd as m

n = int(float(sys.argv[1]) * 1e6) if len(sys.argv) > 1 else 20_000_000
k = 21
r = np.random.default_rng(3)
km = np.unique(r.integers(0, 1 << (2 * k), size=n, dtype=np.uint64))
n = len(km)
rows = np.empty((n, k + 4), dtype=np.uint8)
lut = np.frombuffer(b"ACTG", dtype=np.uint8)
for i in range(k):
    rows[:, i] = lut[((km >> np.uint64(2 * (k - 1 - i))) & np.uint64(3)).astype(np.intp)]
v = r.integers(10, 100, size=n)
rows[:, k] = 9
rows[:, k + 1] = 48 + v // 10
rows[:, k + 2] = 48 + v % 10
rows[:, k + 3] = 10
path = os.environ.get("MFX_TMP", "/tmp") + "/rate.txt"
rows.tofile(path)
size = os.path.getsize(path)
for threads in ("1", None):
    if threads:
        os.environ["MFX_HOST_THREADS"] = threads
    else:
        os.environ.pop("MFX_HOST_THREADS", None)
    t0 = time.time()
    info = m.db_probe(path)
    t1 = time.time()
    ix = m.Index(k, n + 1024)
    ix.load_db(path, 0)
    t2 = time.time()
    assert info["n_kmers"] == n and ix.info()["distinct"] == n
    print("threads=%s: %d lines, %.2f GB: probe %.2f s (%.0f M lines/s), load %.2f s (%.0f M lines/s, %.2f GB/s)"
          % (threads or "all", n, size / 1e9, t1 - t0, n / (t1 - t0) / 1e6, t2 - t1, n / (t2 - t1) / 1e6, size / (t2 - t1) / 1e9), flush=True)
    del ix
