#!/usr/bin/env python3
"""End-to-end timing of the C++ CLI on a BASELINE-config-2-sized case (64 Mb,
one contig, k=21, 30x-like read counts): writes FASTA + flat k-mer DBs, then
runs `merfin -hist` and `merfin -dump` and reports wall times.  Run on the GPU
box:  python tools/cfg2_cli_timing.py [bases]"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import merfin_amd as m
from tools import synth_torch as st

bases = int(float(sys.argv[1])) if len(sys.argv) > 1 else 64_000_000
out = os.environ.get("MFX_TMP", "/tmp/mfx_cfg2")
os.makedirs(out, exist_ok=True)
t0 = time.time()
big = bases > 1_500_000_000                                # a 3 Gb genome: only the read database is written, only -hist without -seqmers runs
if big:
    os.environ["MFX_LOAD_FACTOR"] = "0.85"                  # the world's own table (127 GB) must leave room for its export (96 GB)
ix, seqs, asm, info = st.build_world(m, bases, k=21, lam=26.0, ncontigs=1)
os.environ.pop("MFX_LOAD_FACTOR", None)
ek, er, ea = ix.export(sort=False)
seq = asm[0].cpu().numpy().tobytes()
del ix, seqs, asm
torch.cuda.empty_cache()
# sorted as `meryl print` lists a database (the delta-coded flat form needs that; any order gives packed records) -- sort, filter
# and gather on the GPU, so that the host only ever holds the arrays it writes
kd = torch.from_numpy(ek.view(np.int64)).cuda()
del ek


def sorted_db(values):
    """(k-mers, counts) of the entries with a count, ascending -- eight key ranges, each sorted on its own (torch.sort takes < 2^31 elements)"""
    vd = torch.from_numpy(values.view(np.int32)).cuda()
    ks, vs = [], []
    CH = 1 << 30                                               # (mask selection, like sort, wants < 2^31 elements at a time)
    for q in range(8):
        kp, vp = [], []
        for o in range(0, kd.numel(), CH):
            kc, vc = kd[o:o + CH], vd[o:o + CH]
            sel = ((kc >> 39) == q) & (vc != 0)
            kp.append(kc[sel])
            vp.append(vc[sel])
        kq, o = torch.sort(torch.cat(kp))
        vq = torch.cat(vp)[o]
        ks.append(kq.cpu().numpy().view(np.uint64))
        vs.append(vq.cpu().numpy().view(np.uint32))
        del kp, vp, kq, o, vq, sel
    return np.concatenate(ks), np.concatenate(vs)


rk, rv = sorted_db(er)
del er
m.db_write_flat(out + "/read.mfxk", 21, rk, rv)
if not big and os.environ.get("MFX_TIMING_FORMS", "1") != "0":   # the packed-record form of the same database, for the comparison below
    os.environ["MFX_FLAT_DELTA"] = "0"
    m.db_write_flat(out + "/read_packed.mfxk", 21, rk, rv)
    del os.environ["MFX_FLAT_DELTA"]
n_read = len(rk)
del rk, rv
n_asm = 0
if not big:
    ak, av = sorted_db(ea)
    n_asm = len(ak)
    m.db_write_flat(out + "/asm.mfxk", 21, ak, av)
    del ak, av
del ea, kd
torch.cuda.empty_cache()
print("read database: %.2f GB delta-coded%s" % (os.path.getsize(out + "/read.mfxk") / 1e9,
      ", %.2f GB as packed records" % (os.path.getsize(out + "/read_packed.mfxk") / 1e9) if os.path.exists(out + "/read_packed.mfxk") else ""), flush=True)
with open(out + "/asm.fasta", "wb") as f:
    f.write(b">chr20_like synthetic\n")
    a = np.frombuffer(seq, dtype=np.uint8)                    # 80 bases per line, as assemblies come
    rows = len(a) // 80
    out_ = np.empty((rows, 81), dtype=np.uint8)
    out_[:, :80] = a[:rows * 80].reshape(rows, 80)
    out_[:, 80] = 10
    f.write(out_.tobytes())
    if rows * 80 < len(a):
        f.write(seq[rows * 80:] + b"\n")
    del a, out_
del seq
print("inputs written in %.1fs: %d read k-mers, %d asm k-mers" % (time.time() - t0, n_read, n_asm), flush=True)
exe = os.path.join(ROOT, "merfin_amd", "bin", "merfin")
prob = os.path.join(ROOT, "tests", "golden", "example_lookup_table.txt")
common = ["-sequence", out + "/asm.fasta", "-readmers", out + "/read.mfxk", "-seqmers", out + "/asm.mfxk", "-peak", "26", "-prob", prob]
modes = (("-hist", out + "/out.hist"), ("-dump", out + "/out.dump")) if bases <= 128_000_000 else (("-hist", out + "/out.hist"),)
if big:
    modes = ()
    for stale in ("/read_packed.mfxk", "/asm.mfxk"):
        if os.path.exists(out + stale):
            os.remove(out + stale)
for mode, o in modes:
    t = time.time()
    r = subprocess.run([exe, mode] + common + ["-output", o], capture_output=True, text=True, env=dict(os.environ, MFX_CLI_TIMING="2"))
    dt = time.time() - t
    tail = [l for l in r.stderr.splitlines() if l and not l.startswith("Copy-number")][-14:]
    print("%s: rc=%d wall=%.2fs output=%.1f MB" % (mode, r.returncode, dt, os.path.getsize(o) / 1e6))
    for l in tail:
        print("    " + l)
# -hist without -seqmers: assembly k-mers counted on the GPU
t = time.time()
r = subprocess.run([exe, "-hist", "-sequence", out + "/asm.fasta", "-readmers", out + "/read.mfxk", "-peak", "26", "-prob", prob,
                    "-output", out + "/out2.hist"], capture_output=True, text=True, env=dict(os.environ, MFX_CLI_TIMING="2"))
if big:
    import shutil
    shutil.copy(out + "/out2.hist", out + "/out.hist")       # (nothing to compare with at this size: the later runs against the first)
print("-hist (asm counted on GPU): rc=%d wall=%.2fs same_hist=%s" % (r.returncode, time.time() - t,
      open(out + "/out.hist").read() == open(out + "/out2.hist").read()))
print("    " + "\n    ".join(l for l in r.stderr.splitlines() if "timing" in l))
for rep in range(2):
    for env, what in (({"MFX_CLI_OVERLAP": "0"}, "sequence read first"), ({"MFX_CLI_OVERLAP": "1"}, "sequence read under the index build")):
        t = time.time()
        r = subprocess.run([exe, "-hist", "-sequence", out + "/asm.fasta", "-readmers", out + "/read.mfxk", "-peak", "26", "-prob", prob,
                            "-output", out + "/out3.hist"], capture_output=True, text=True, env=dict(os.environ, MFX_CLI_TIMING="2", **env))
        print("-hist without -seqmers, %s: wall=%.2fs" % (what, time.time() - t))
        print("    " + "\n    ".join(l for l in r.stderr.splitlines() if "timing" in l))
if os.path.exists(out + "/read_packed.mfxk"):
    for rep in range(2):
        for db, what in (("/read.mfxk", "delta-coded blocks"), ("/read_packed.mfxk", "packed records")):
            t = time.time()
            r = subprocess.run([exe, "-hist", "-sequence", out + "/asm.fasta", "-readmers", out + db, "-peak", "26", "-prob", prob,
                                "-output", out + "/out4.hist"], capture_output=True, text=True, env=dict(os.environ, MFX_CLI_TIMING="2"))
            print("-hist without -seqmers, read database as %s: wall=%.2fs same_hist=%s" % (what, time.time() - t,
                  open(out + "/out.hist").read() == open(out + "/out4.hist").read()))
            print("    " + "\n    ".join(l for l in r.stderr.splitlines() if "timing" in l))
for sweep in [x for x in os.environ.get("MFX_TIMING_SWEEP", "").split(";") if x]:
    env = dict(kv.split("=", 1) for kv in sweep.split())
    for rep in range(2):
        t = time.time()
        r = subprocess.run([exe, "-hist", "-sequence", out + "/asm.fasta", "-readmers", out + "/read.mfxk", "-peak", "26", "-prob", prob,
                            "-output", out + "/out5.hist"], capture_output=True, text=True, env=dict(os.environ, MFX_CLI_TIMING="2", MFX_INGEST_TIMING="1", **env))
        print("-hist without -seqmers, %s: wall=%.2fs same_hist=%s" % (sweep, time.time() - t, open(out + "/out.hist").read() == open(out + "/out5.hist").read()))
        print("    " + "\n    ".join(l for l in r.stderr.splitlines() if "timing" in l or "ingest:" in l or "read_fasta" in l))
