#!/usr/bin/env python3
"""-hist kernel rates by k-mer size on one random sequence (every k-mer of the sequence present with a read count):
python tools/hist_rates_by_k.py [bases=5e8] [k ...]   -- the index as `merfin -hist` builds it (sequence-only for k <= 31,
the 128-bit tables above), the evaluation timed over three runs with the sequence resident."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import numpy as np
    import torch
    import merfin_amd as m
    from tools import synth_torch as st
    bases = int(float(sys.argv[1])) if len(sys.argv) > 1 else 500_000_000
    ks = [int(x) for x in sys.argv[2:]] or [21, 22, 31, 32, 41, 63]
    seq = st.random_bases(bases, 777, "cuda")
    sq = m.Sequences.from_device([seq.data_ptr()], [bases])
    for k in ks:
        kinds = ("seq", "full") if k <= 31 else ("full",)
        if os.environ.get("MFX_RATES_KINDS"):                      # e.g. MFX_RATES_KINDS=seq: one index kind only (PMC passes)
            kinds = tuple(x for x in kinds if x in os.environ["MFX_RATES_KINDS"].split(","))
        for kind in kinds:
            ix = m.Index.for_seq(k, bases + 1024) if kind == "seq" else m.Index(k, bases + 1024)
            t = time.perf_counter()
            ix.count_asm(sq)
            torch.cuda.synchronize()
            t_count = time.perf_counter() - t
            # read counts: the table's own k-mers, count 20 each (an update of every slot)
            ek, _, _ = ix.export(sort=False)
            kd = torch.from_numpy(np.ascontiguousarray(ek).view(np.int64)).cuda()
            vd = torch.full((len(ek),), 20, dtype=torch.int32, device="cuda")
            ix.add_read(kd, vd)
            del ek, kd, vd
            ev = m.Evaluator(ix, m.KParams(20.0))
            dts = []
            for _ in range(3):
                torch.cuda.synchronize()
                t = time.perf_counter()
                h = ev.hist(sq)
                dts.append(time.perf_counter() - t)
            n = bases - k + 1
            info = ix.info()
            print("k=%2d %-4s index: count %.1f G k-mers/s, table %.1f GB%s;  -hist %.2f ms = %.1f G k-mers/s  (kmissing %d)" %
                  (k, kind, n / t_count / 1e9, info["bytes"] / 1e9, " compact" if info["compact"] else "", min(dts) * 1e3, n / min(dts) / 1e9, h.kmissing),
                  flush=True)
            del ev, ix, h
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
