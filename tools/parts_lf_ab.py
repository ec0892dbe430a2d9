"""-hist by PARTS on one GPU (8 slots, each the sequence-only index of its contigs) at several table load factors:
   python tools/parts_lf_ab.py [bases=3e9] [k=31] [lf ...]        ("auto" = the library's own choice)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import numpy as np
    import torch
    import merfin_amd as m
    from tools import synth_torch as st
    bases = int(float(sys.argv[1])) if len(sys.argv) > 1 else 3_000_000_000
    k = int(sys.argv[2]) if len(sys.argv) > 2 else 31
    lfs = sys.argv[3:] or ["auto", "0.225"]
    nslots, ncont = 8, 24
    per = bases // ncont
    asm = [st.random_bases(per, 1000 + c, "cuda") for c in range(ncont)]
    whole = m.Sequences.from_device([a.data_ptr() for a in asm], [per] * ncont)
    ids = [list(range(d, ncont, nslots)) for d in range(nslots)]
    for rep in (1, 2):
        for lf in lfs:
            if lf == "auto":
                os.environ.pop("MFX_LOAD_FACTOR", None)
            else:
                os.environ["MFX_LOAD_FACTOR"] = lf
            ixs, own = [], []
            for mine in ids:
                sq = m.Sequences.from_device([asm[i].data_ptr() for i in mine], [per] * len(mine))
                ix = m.Index.for_seq(k, per * len(mine) + 1024)
                ix.claim_seq(sq)
                ix.count_claimed(whole)
                ek, _, _ = ix.export(sort=False)
                kd = torch.from_numpy(np.ascontiguousarray(ek).view(np.int64)).cuda()
                vd = torch.full((len(ek),), 20, dtype=torch.int32, device="cuda")
                ix.add_read(kd, vd)
                del ek, kd, vd
                ixs.append(ix)
                own.append(sq)
            evs = [m.Evaluator(ix, m.KParams(20.0)) for ix in ixs]
            r = m.hist_parts(evs, own, ids, ncont)
            ts = []
            for _ in range(5):
                torch.cuda.synchronize()
                t = time.perf_counter()
                r = m.hist_parts(evs, own, ids, ncont)
                ts.append(time.perf_counter() - t)
            # one slot alone, for comparison
            t = time.perf_counter()
            evs[0].hist(own[0])
            t1 = time.perf_counter() - t
            gb = sum(ix.info()["bytes"] for ix in ixs) / 1e9
            print("lf %-5s rep%d: tables %.1f GB; -hist by parts %s ms -> best %.1f G k-mers/s; slot 0 alone %.2f ms = %.1f G k-mers/s (kmissing %d)" %
                  (lf, rep, gb, " ".join("%.1f" % (x * 1e3) for x in ts), r.kasm / min(ts) / 1e9, t1 * 1e3, per * len(ids[0]) / t1 / 1e9, r.kmissing), flush=True)
            for e in evs:
                e.close() if hasattr(e, "close") else None
            del evs, ixs, own, r
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
