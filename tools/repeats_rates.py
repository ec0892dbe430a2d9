#!/usr/bin/env python3
"""-hist kernel rate on the `repeats` worlds (tools/synth_torch.py: inject_repeats) next to SURVEY 8(d)'s i.i.d. world:
for every level the resident kernel time (HIP events on the launch stream), the share of the k-mers whose lookup ended in
the side table (a saturated 11-bit count field; mfx_eval_debug_counters) and the other endings of the probe.

  python tools/repeats_rates.py [--bases 3e9] [--levels 0,1,3,10] [--k 21] [--steps 5]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--bases", type=float, default=3e9)
    ap.add_argument("--levels", default="0,1,3,10")
    ap.add_argument("--k", type=int, default=21)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--index", choices=("seq", "full"), default="seq")
    args = ap.parse_args()
    import torch
    import merfin_amd as m
    from tools import synth_torch as st
    torch.cuda.set_device(0)
    lam = 26.0
    kp = m.KParams.from_file(lam, os.path.join(ROOT, "tests", "golden", "example_lookup_table.txt"))
    rows = []
    for level in [int(x) for x in args.levels.split(",")]:
        t0 = time.time()
        ix, seqs, asm, info = st.build_world(m, int(args.bases), k=args.k, lam=lam, ncontigs=24, seq_only=args.index == "seq", repeats=level)
        del asm
        ev = m.Evaluator(ix, kp)
        counts = torch.zeros(m.hist_words(ev.nbins, seqs.ncontigs), dtype=torch.int64, device="cuda")
        kover = torch.zeros(1, dtype=torch.float64, device="cuda")
        stream = torch.cuda.current_stream().cuda_stream
        for _ in range(2):
            ev.hist_launch(seqs, 0, seqs.ntiles, counts, kover, stream=stream)
        torch.cuda.synchronize()
        try:
            ev.take_overflow()
        except Exception as e:
            print("take_overflow: %r" % (e,), file=sys.stderr)
        ms = []
        for _ in range(args.steps):
            counts.zero_(); kover.zero_()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); ev.hist_launch(seqs, 0, seqs.ntiles, counts, kover, stream=stream); e1.record()
            torch.cuda.synchronize()
            ms.append(e0.elapsed_time(e1))
        nb = ev.nbins
        kasm, kmissing, novf = (int(counts[2 * nb + i].item()) for i in range(3))
        dbg = None
        try:
            ev.debug(True)
            counts.zero_(); kover.zero_()
            ev.hist_launch(seqs, 0, seqs.ntiles, counts, kover, stream=stream)
            torch.cuda.synchronize()
            dc = ev.debug_counters()
            dbg = [dc["first_pass"], dc["second_pass"], dc["side_table"], dc["line_scans"], dc.get("side_not_in_two_slots", 0), dc.get("ended_per_lane", 0)]
            ev.debug(False)
        except Exception as e:
            print("debug counters: %r" % (e,), file=sys.stderr)
        row = {"level": level, "k": args.k, "bases": int(args.bases), "kasm": kasm, "kmissing": kmissing, "novf": novf, "kernel_ms": ms, "kernel_ms_min": min(ms),
               "kmers_per_s": kasm / (min(ms) * 1e-3), "index_gb": info["bytes"] / 1e9, "distinct": int(info["distinct"]), "build_s": time.time() - t0,
               "repeats": info.get("repeats"), "dbg": dbg,
               "side_share": (dbg[2] / kasm) if dbg else None, "displaced_share": (dbg[0] / kasm) if dbg else None}
        print(json.dumps(row), flush=True)
        rows.append(row)
        del ev, ix, seqs, counts, kover
        torch.cuda.empty_cache()
    base = [r for r in rows if r["level"] == 0]
    if base:
        for r in rows:
            print("level %2d: %.1f G k-mers/s (%.3f of the i.i.d. world), side-table share %s" %
                  (r["level"], r["kmers_per_s"] / 1e9, r["kmers_per_s"] / base[0]["kmers_per_s"],
                   "%.4f" % r["side_share"] if r["side_share"] is not None else "n/a"), flush=True)


if __name__ == "__main__":
    main()
