"""Inputs of an end-to-end `merfin -hist` run made from the synthetic world of bench.py: the assembly as a FASTA
file (80 bases per line) and the read database -- every read k-mer of the FULL tables, sorted as `meryl print` lists them --
as this repo's delta-coded flat file (mfx_db_write_flat; what `merfin -convert` makes of a meryl database once).
Bench / profiling infrastructure: used by bench.py's `e2e` leg and tools/cfg2_cli_timing.py-style timings."""
import os
import time

import numpy as np


def sorted_nonzero(torch, ek, ev, k, placed_with=None):
    """the entries of an (unsorted) index export that have a count, sorted by k-mer as `meryl print` lists them (what mfx_db_write_flat
    delta-codes) -- on the GPU, in key ranges (torch.sort takes < 2^31 elements); k <= 31.  placed_with = the merfin_amd module: the keys
    are replaced by their placement numbers (mfx_db_place_keys, csrc/mfx_place.h) first -- the order of a PLACED database"""
    kd = torch.from_numpy(np.ascontiguousarray(ek).view(np.int64)).cuda()
    vd = torch.from_numpy(np.ascontiguousarray(ev).view(np.int32)).cuda()
    nq = max(1, int(np.ceil(kd.numel() / 7e8)))
    shift = 2 * k
    if placed_with is not None:
        CHP = 1 << 28
        for o in range(0, kd.numel(), CHP):
            placed_with.db_place_keys(k, kd[o:o + CHP], out=kd[o:o + CHP])
        shift = max(2 * k + 3, 41)
    ks, vs = [], []
    CH = 1 << 30
    for q in range(nq):
        lo, hi = (q << shift) // nq, ((q + 1) << shift) // nq
        kp, vp = [], []
        for o in range(0, kd.numel(), CH):
            kc, vc = kd[o:o + CH], vd[o:o + CH]
            sel = (kc >= lo) & (kc < hi) & (vc != 0)
            kp.append(kc[sel])
            vp.append(vc[sel])
        kq, o = torch.sort(torch.cat(kp))
        vq = torch.cat(vp)[o]
        ks.append(kq.cpu().numpy().view(np.uint64))
        vs.append(vq.cpu().numpy().view(np.uint32))
        del kp, vp, kq, o, vq, sel
    del kd, vd
    torch.cuda.empty_cache()
    return np.concatenate(ks), np.concatenate(vs)


def write_inputs(m, st, torch, bases, outdir, ncontigs=24, k=21, lam=26.0, seed=None, log=lambda *a: None, placed=False):
    """returns {"fasta": path, "readdb": path, "read_kmers": n, "db_bytes": b, "fasta_bytes": b, "lens": [...], "write_s": s}"""
    t0 = time.time()
    os.makedirs(outdir, exist_ok=True)
    # the world's own full table only has to leave room for its export: a crowded table is fine here
    os.environ["MFX_LOAD_FACTOR"] = "0.85"
    try:
        kw = {} if seed is None else {"seed": seed}
        ix, seqs, asm, info = st.build_world(m, bases, k=k, lam=lam, ncontigs=ncontigs, **kw)
    finally:
        os.environ.pop("MFX_LOAD_FACTOR", None)
    ek, er, _ea = ix.export(sort=False)
    del _ea
    contigs = [a.cpu().numpy() for a in asm]
    ix.close()
    seqs.close()
    del ix, seqs, asm
    torch.cuda.empty_cache()
    log("world built and exported: %.1fs" % (time.time() - t0))
    rk, rv = sorted_nonzero(torch, ek, er, k)
    readdb = os.path.join(outdir, "read.mfxk")
    m.db_write_flat(readdb, k, rk, rv)
    text_sample = os.path.join(outdir, "text_sample.npz")      # the first k-mers of the database, for text_convert_sample
    np.savez(text_sample, km=rk[:20_000_000], rv=rv[:20_000_000])
    placeddb = None
    if placed:                                                 # the same database in the PLACED form (what `merfin -convert -placed` makes), next to it
        del rk, rv
        rk, rv = sorted_nonzero(torch, ek, er, k, placed_with=m)
        placeddb = os.path.join(outdir, "read.placed.mfxk")
        m.db_write_flat_placed(placeddb, k, rk, rv)
    del ek, er
    n_read = len(rk)
    del rk, rv
    torch.cuda.empty_cache()                                   # (the caller measures OTHER processes next: nothing of this one's stays cached in HBM)
    log("read database written: %.1fs" % (time.time() - t0))
    fasta = os.path.join(outdir, "asm.fasta")
    with open(fasta, "wb") as f:
        for ci, a in enumerate(contigs):
            f.write(b">contig_%d synthetic\n" % (ci + 1))
            rows = len(a) // 80
            out_ = np.empty((rows, 81), dtype=np.uint8)
            out_[:, :80] = a[:rows * 80].reshape(rows, 80)
            out_[:, 80] = 10
            f.write(out_.tobytes())
            if rows * 80 < len(a):
                f.write(a[rows * 80:].tobytes() + b"\n")
            del out_
    return {"fasta": fasta, "readdb": readdb, "text_sample": text_sample, "placeddb": placeddb, "placed_db_bytes": os.path.getsize(placeddb) if placeddb else None, "read_kmers": n_read, "db_bytes": os.path.getsize(readdb), "fasta_bytes": os.path.getsize(fasta),
            "lens": [len(a) for a in contigs], "write_s": time.time() - t0}


def run_cli_hist(root, inp, peak=26.0, prob=None, out_hist=None, env=None):
    """one `merfin -hist` process on the inputs; returns (returncode, wall seconds, {phase: seconds}, stderr)"""
    import subprocess
    exe = os.path.join(root, "merfin_amd", "bin", "merfin")
    cmd = [exe, "-hist", "-sequence", inp["fasta"], "-readmers", inp["readdb"], "-peak", str(peak), "-output", out_hist or os.path.join(os.path.dirname(inp["fasta"]), "out.hist")]
    if prob:
        cmd += ["-prob", prob]
    t = time.time()
    r = subprocess.run(cmd, stdin=subprocess.DEVNULL, capture_output=True, text=True, env=dict(os.environ, MFX_CLI_TIMING="2", **(env or {})))
    wall = time.time() - t
    phases = {}
    for line in r.stderr.splitlines():
        if line.startswith("-- timing"):                        # "-- timing:  name 0.12s  other name 3.40s ..."
            for tok in line.split(":", 1)[1].split("  "):
                tok = tok.strip()
                if tok.endswith("s") and " " in tok:
                    name, sec = tok.rsplit(" ", 1)
                    try:
                        phases[name.strip()] = float(sec[:-1])
                    except ValueError:
                        pass
    return r.returncode, wall, phases, r.stderr


def text_convert_sample(root, m, inp, outdir, k, n_lines=20_000_000):
    """What a user who starts from `meryl print` TEXT pays once: the first n_lines k-mers of the read database written as text
    (<kmer>\t<count>), converted by `merfin -convert` (host only), timed; the full database's conversion is extrapolated from the rate."""
    import subprocess
    import numpy as np
    if not inp.get("text_sample") or not os.path.exists(inp["text_sample"]):
        return None
    z = np.load(inp["text_sample"])
    km, rv = z["km"][:n_lines].astype(np.uint64), z["rv"][:n_lines]
    n = len(km)
    if n == 0:
        return None
    rows = np.empty((n, k + 7), dtype=np.uint8)
    lut = np.frombuffer(b"ACTG", dtype=np.uint8)
    for i in range(k):
        rows[:, i] = lut[((km >> np.uint64(2 * (k - 1 - i))) & np.uint64(3)).astype(np.intp)]
    v = np.minimum(rv, 99999).astype(np.int64)
    rows[:, k] = 9
    for d in range(5):
        rows[:, k + 1 + d] = 48 + (v // 10 ** (4 - d)) % 10
    rows[:, k + 6] = 10
    txt = os.path.join(outdir, "sample.txt")
    rows.tofile(txt)
    del rows
    exe = os.path.join(root, "merfin_amd", "bin", "merfin")
    t = time.time()
    r = subprocess.run([exe, "-convert", txt, "-output", os.path.join(outdir, "sample.mfxk")], stdin=subprocess.DEVNULL, capture_output=True, text=True)
    dt = time.time() - t
    size = os.path.getsize(txt)
    os.unlink(txt)
    if r.returncode != 0:
        raise RuntimeError("merfin -convert failed: " + r.stderr[-300:])
    rate = n / dt
    return {"text_kmers": n, "text_bytes": size, "convert_s": dt, "convert_kmers_per_s": rate,
            "convert_s_full_db_extrapolated": inp["read_kmers"] / rate,
            "note": "`merfin -convert` of a %d-line `meryl print`-style text sample of the same database (host only: parse, sort check, delta-code); the full database's "
                    "conversion time is this rate times its k-mers -- paid once per database, not per run" % n}
