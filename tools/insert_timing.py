#!/usr/bin/env python3
"""Index-build kernel rates (device-resident inputs, no PCIe): fresh inserts, repeated inserts (every key already
present) and the assembly k-mer counter.  python tools/insert_timing.py [n_keys_log2=28] [bases=1e9]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    import merfin_amd as m
    from tools import synth_torch as st
    lg = int(sys.argv[1]) if len(sys.argv) > 1 else 28
    bases = int(float(sys.argv[2])) if len(sys.argv) > 2 else 1_000_000_000
    n, k = 1 << lg, 21
    mask = (1 << (2 * k)) - 1
    h = st._hash_range(12345, 0, n, "cuda") & mask
    x, r = h, torch.zeros_like(h)
    for _ in range(k):
        r = (r << 2) | ((x & 3) ^ 2)
        x = x >> 2
    keys = torch.minimum(h, r).contiguous()
    vals = torch.ones(n, dtype=torch.int32, device="cuda")
    for lf in ("0.7", "0.5"):
        os.environ["MFX_LOAD_FACTOR"] = lf
        ix = m.Index(k, n + 1024)
        torch.cuda.synchronize()
        out = []
        for rep in ("fresh", "again"):
            t = time.perf_counter()
            ix.add_read(keys, vals)          # synchronous
            dt = time.perf_counter() - t
            out.append("%s %.2f G inserts/s (%.1f ms)" % (rep, n / dt / 1e9, dt * 1e3))
        print("table_add  n=2^%d  load factor %s: %s; distinct %d" % (lg, lf, ", ".join(out), ix.info()["distinct"]), flush=True)
        del ix
    del os.environ["MFX_LOAD_FACTOR"]
    seq = st.random_bases(bases, 777, "cuda")
    sq = m.Sequences.from_device([seq.data_ptr()], [bases])
    ix = m.Index(k, bases + 1024)
    for rep in ("fresh", "again"):
        t = time.perf_counter()
        ix.count_asm(sq)
        dt = time.perf_counter() - t
        print("count_asm  %d bases (%s): %.2f G k-mers/s (%.1f ms)" % (bases, rep, (bases - k + 1) / dt / 1e9, dt * 1e3), flush=True)


if __name__ == "__main__":
    main()
