#!/usr/bin/env python3
"""Host-side rate of the sequence packer (csrc/mfx_pack.cpp) on this box: T threads, each encoding its share of a buffer.
   python tools/pack_rate.py [MB] """
import ctypes as C
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
L = C.CDLL(os.path.join(ROOT, "merfin_amd", "libmerfin_amd.so"))
L.mfx_pack_bases.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
mb = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
n = mb << 20
src = np.frombuffer(np.random.default_rng(1).choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=n).tobytes(), dtype=np.uint8)
codes = np.zeros(n // 32, dtype=np.uint64)
valid = np.zeros(n // 32, dtype=np.uint32)
for isa in ("", "avx2"):
    if isa:
        os.environ["MFX_PACK_ISA"] = isa
    for T in (1, 4, 8, 16, 32):
        per = n // T // 32 * 32
        def run(t):
            L.mfx_pack_bases(src.ctypes.data + t * per, per, codes.ctypes.data + t * per // 4, valid.ctypes.data + t * per // 8)
        best = 1e9
        for rep in range(3):
            th = [threading.Thread(target=run, args=(t,)) for t in range(T)]
            t0 = time.time()
            [x.start() for x in th]
            [x.join() for x in th]
            best = min(best, time.time() - t0)
        print("isa=%-6s threads=%2d  %.1f GB/s (%.2f per thread)" % (isa or "best", T, T * per / best / 1e9, per / best / 1e9), flush=True)
