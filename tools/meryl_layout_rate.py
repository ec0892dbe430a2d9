#!/usr/bin/env python3
"""Rate of the meryl-directory decoder (csrc/mfx_db.cpp, unvalidated layout: tests/meryl_layout.py writes it) on this box.
   python tools/meryl_layout_rate.py [million_kmers]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import merfin_amd as m
from tests import meryl_layout

n = int(float(sys.argv[1]) * 1e6) if len(sys.argv) > 1 else 10_000_000
k = 21
r = np.random.default_rng(5)
km = np.unique(r.integers(0, 1 << (2 * k), size=n, dtype=np.uint64))
v = r.integers(1, 200, size=len(km)).astype(np.uint32)
path = os.environ.get("MFX_TMP", "/tmp") + "/rate.meryl"
t0 = time.time()
meryl_layout.write_db(path, k, km, v, prefix_bits=16)
size = sum(os.path.getsize(os.path.join(path, f)) for f in os.listdir(path))
print("wrote %d k-mers, %.1f MB (%.2f B/k-mer) in %.1f s" % (len(km), size / 1e6, size / len(km), time.time() - t0), flush=True)
for rep in range(2):
    t0 = time.time()
    info = m.db_probe(path)
    t1 = time.time()
    ix = m.Index(k, len(km) + 1024)
    ix.load_db(path, 0)
    t2 = time.time()
    assert info["n_kmers"] == len(km) and ix.info()["distinct"] == len(km)
    print("probe %.2f s (%.0f M k-mers/s), load %.2f s (%.0f M k-mers/s)" % (t1 - t0, len(km) / (t1 - t0) / 1e6, t2 - t1, len(km) / (t2 - t1) / 1e6), flush=True)
    del ix
