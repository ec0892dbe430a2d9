#!/usr/bin/env python3
"""A k = 31 read database applied to a sequence-only index: k-mer-sorted (mfxk-delta) against PLACED (P >> 1 + strand bit, mfx_place.h),
the same table either way.  Times mfx_index_build_for_hist (claim + count the sequence, then the database update-only) per form.

  python tools/placed_k31_load.py [bases=5e8] [k=31]
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    bases = int(float(sys.argv[1])) if len(sys.argv) > 1 else 500_000_000
    k = int(sys.argv[2]) if len(sys.argv) > 2 else 31
    import numpy as np
    import torch
    import merfin_amd as m
    from tools import synth_torch as st
    from tools import e2e_inputs
    out = os.environ.get("MFX_TMP", "/tmp/mfx_k31")
    os.makedirs(out, exist_ok=True)
    lam = 26.0
    t0 = time.time()
    ix, seqs, asm, info = st.build_world(m, bases, k=k, lam=lam, ncontigs=24, seq_only=False)
    ek, er, ea = ix.export(sort=False)
    del ix
    torch.cuda.empty_cache()
    sk, sv = e2e_inputs.sorted_nonzero(torch, ek, er, k)
    del ek, er, ea
    flat, placed = out + "/read.mfxk", out + "/read.placed.mfxk"
    m.db_write_flat(flat, k, sk, sv)
    n = len(sk)
    del sk, sv
    print("world: %d bases, k = %d, read database %d k-mers, %.2f GB sorted (%.1f s)" % (bases, k, n, os.path.getsize(flat) / 1e9, time.time() - t0), flush=True)
    t0 = time.time()
    assert m.db_convert_placed(flat, placed) == n
    print("converted to the placed form on the host in %.1f s: %.2f GB (%.2f bytes per k-mer; sorted: %.2f)" %
          (time.time() - t0, os.path.getsize(placed) / 1e9, os.path.getsize(placed) / n, os.path.getsize(flat) / n), flush=True)
    nb = bases
    tables = {}
    for rep in range(3):
        for name, path in (("sorted", flat), ("placed", placed)):
            px = m.Index.for_seq(k, nb + 1024, load_factor=0.4)
            torch.cuda.synchronize()
            t0 = time.time()
            px.build_for_hist(seqs, path)
            dt = time.time() - t0
            inf = px.info()
            print("%-6s rep %d: build_for_hist %.3f s  (%d k-mers in the table, %d records dropped)" % (name, rep, dt, inf["distinct"], inf["dropped"]), flush=True)
            if rep == 0:
                ek, er, ea = px.export(sort=False)
                o = np.argsort(ek, kind="stable")
                tables[name] = (ek[o], er[o], ea[o])
            del px
            torch.cuda.empty_cache()
    a, b = tables["sorted"], tables["placed"]
    print("the two tables hold the same k-mers and counts: %s" % all(np.array_equal(x, y) for x, y in zip(a, b)), flush=True)


if __name__ == "__main__":
    main()
