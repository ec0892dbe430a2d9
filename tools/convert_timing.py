#!/usr/bin/env python3
"""mfx_db_convert on the host of the GPU box: a `meryl print`-shaped text of n k-mers (k = 21) -> the delta-coded flat form.
python tools/convert_timing.py [n=2e8]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import merfin_amd as m

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 200_000_000
k = 21
out = os.environ.get("MFX_TMP", "/tmp/mfx_conv")
os.makedirs(out, exist_ok=True)
r = np.random.default_rng(5)
dec = np.frombuffer(b"ACTG", dtype=np.uint8)
t0 = time.time()
with open(out + "/db.txt", "wb") as f:
    base = 0
    CH = 10_000_000
    total = 0
    for c in range(0, n, CH):
        cnt = min(CH, n - c)
        # ascending keys: every chunk owns a key range
        span = (1 << 42) // ((n + CH - 1) // CH)
        keys = np.unique(r.integers(0, span, size=cnt + cnt // 8, dtype=np.uint64))[:cnt] + np.uint64(base)
        base += span
        vals = r.integers(1, 60, size=len(keys))
        cols = np.empty((len(keys), k + 4), dtype=np.uint8)
        for i in range(k):
            cols[:, i] = dec[((keys >> np.uint64(2 * (k - 1 - i))) & np.uint64(3)).astype(np.int64)]
        cols[:, k] = 9
        cols[:, k + 1] = 48 + vals // 10
        cols[:, k + 2] = 48 + vals % 10
        cols[:, k + 3] = 10
        f.write(cols.tobytes())
        total += len(keys)
print("text of %d k-mers, %.2f GB, written in %.1f s" % (total, os.path.getsize(out + "/db.txt") / 1e9, time.time() - t0), flush=True)
os.environ["MFX_DB_TIMING"] = "1"
for rep in range(2):
    t = time.time()
    nn = m.db_convert(out + "/db.txt", out + "/db.mfxk")
    dt = time.time() - t
    print("convert: %d k-mers in %.2f s = %.1f M k-mers/s; %.2f GB out (%.2f bytes per k-mer)" %
          (nn, dt, nn / dt / 1e6, os.path.getsize(out + "/db.mfxk") / 1e9, os.path.getsize(out + "/db.mfxk") / nn), flush=True)
