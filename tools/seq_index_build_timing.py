#!/usr/bin/env python3
"""Build-kernel rates of the SEQUENCE-ONLY index (device-resident inputs, no PCIe): claiming the k-mers of a sequence
(mfx_count_kernel, count = 0), counting them (count = 1), and update-only adds of a read database of which half the
k-mers are in the sequence (mfx_table_update_kernel).  python tools/seq_index_build_timing.py [bases=1e9] [k=21]
Prints one line per kernel; a checksum of the resulting table (distinct / dropped) so that A/B builds can be compared."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    import merfin_amd as m
    from tools import synth_torch as st
    bases = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
    k = int(sys.argv[2]) if len(sys.argv) > 2 else 21
    mask = (1 << (2 * k)) - 1
    seq = st.random_bases(bases, 777, "cuda")
    sq = m.Sequences.from_device([seq.data_ptr()], [bases])
    # the read database: the canonical k-mers of the first 2^27 positions (present) + 2^27 random canonical k-mers
    # (absent, up to chance), shuffled together as a hash placement sees a sorted database: no locality either way
    n_half = 1 << 27
    pres, ok = st.canonical_kmers(seq[: n_half + k - 1], k)
    pres = pres[ok][:n_half]
    h = st._hash_range(4242, 0, n_half, "cuda") & mask
    x, r = h, torch.zeros_like(h)
    for _ in range(k):
        r = (r << 2) | ((x & 3) ^ 2)
        x = x >> 2
    absent = torch.minimum(h, r)
    keys = torch.cat([pres, absent])
    keys = keys[torch.argsort(st._hash_range(99, 0, keys.numel(), "cuda"))].contiguous()
    vals = torch.full((keys.numel(),), 3, dtype=torch.int32, device="cuda")
    del pres, absent, h, x, r
    for mode in ("claim", "count"):
        ix = m.Index.for_seq(k, bases + 1024)
        torch.cuda.synchronize()
        for rep in ("fresh", "again"):
            t = time.perf_counter()
            (ix.claim_seq if mode == "claim" else ix.count_asm)(sq)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t
            print("%s_seq  %d bases k=%d (%s): %.2f G k-mers/s (%.1f ms)" % (mode, bases, k, rep, (bases - k + 1) / dt / 1e9, dt * 1e3),
                  flush=True)
        if mode == "count":
            break
        for rep in range(3):
            t = time.perf_counter()
            ix.add_read(keys, vals)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t
            print("update  n=%d (half present) rep %d: %.2f G k-mers/s (%.1f ms)" % (keys.numel(), rep, keys.numel() / dt / 1e9, dt * 1e3),
                  flush=True)
        i = ix.info()
        print("index: distinct %d dropped %d compact %d table %.1f GB" % (i["distinct"], i["dropped"], i["compact"], i["bytes"] / 1e9),
              flush=True)
        del ix


if __name__ == "__main__":
    main()
