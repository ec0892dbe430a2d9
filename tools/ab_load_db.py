"""A/B of the kernels that build the sequence-only index from FILES: per library (tools/ab_build.sh; "default" = the in-tree one)
a child process claims + counts the k-mers of a FASTA and loads the delta-coded read database twice (the second load has its
staging lanes pinned already), and prints the times and the table's counters (the same in every variant).
   python tools/ab_load_db.py <bases> <label>=<lib path | default> ...        (on the GPU box; inputs as bench.py's e2e leg)"""
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(fasta, readdb):
    import numpy as np
    import merfin_amd as m
    raw = np.fromfile(fasta, dtype=np.uint8)
    nl = np.flatnonzero(raw == 10)
    starts = np.concatenate([[0], nl[:-1] + 1])
    hdr = raw[starts] == ord(">")
    hs, he = starts[hdr], nl[hdr]
    contigs = []
    for i in range(len(hs)):
        body = raw[he[i] + 1:(hs[i + 1] if i + 1 < len(hs) else len(raw))]
        contigs.append(body[body != 10].tobytes())
    del raw
    bases = sum(len(c) for c in contigs)
    k = m.db_probe(readdb)["k"]
    sq = m.Sequences(contigs)
    m.device_warm(0)
    ix = m.Index.for_seq(k, bases + 1024)
    t0 = time.time()
    ix.count_asm(sq)
    t1 = time.time()
    ix.load_db(readdb, 0)
    t2 = time.time()
    ix.load_db(readdb, 0)
    t3 = time.time()
    info = ix.info()
    n = m.db_probe(readdb)["n_kmers"]
    print("count %.3f s (%.1f G k-mers/s)  load %.3f s, again %.3f s (%.1f G k-mers/s)  table %.1f GB  distinct %d dropped %s" %
          (t1 - t0, bases / (t1 - t0) / 1e9, t2 - t1, t3 - t2, n / (t3 - t2) / 1e9, info["bytes"] / 1e9, info["distinct"], info.get("dropped")), flush=True)


def main():
    if sys.argv[1] == "--child":
        return child(sys.argv[2], sys.argv[3])
    import torch
    import merfin_amd as m
    from tools import synth_torch as st, e2e_inputs
    bases = int(float(sys.argv[1]))
    tmp = tempfile.mkdtemp(prefix="mfx_abl_", dir="/dev/shm" if os.path.isdir("/dev/shm") else "/tmp")
    try:
        inp = e2e_inputs.write_inputs(m, st, torch, bases, tmp, ncontigs=24, k=int(os.environ.get("MFX_AB_K", "21")), lam=26.0)
        torch.cuda.empty_cache()
        for rep in (1, 2):
            for spec in sys.argv[2:]:
                label, lib = spec.split("=", 1)
                env = dict(os.environ)
                if lib != "default":
                    env["MFX_LIB"] = os.path.join(ROOT, lib)
                r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", inp["fasta"], inp["readdb"]], capture_output=True, text=True, env=env)
                out = [l for l in r.stdout.splitlines() if l.startswith("count")]
                print("%-10s rep%d  %s" % (label, rep, out[0] if out else "rc %d %s" % (r.returncode, r.stderr[-300:])), flush=True)
                time.sleep(2)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
