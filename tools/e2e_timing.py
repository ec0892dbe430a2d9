#!/usr/bin/env python3
"""`merfin -hist` end to end (FASTA + delta-coded read database in tmpfs -> histogram file) on the bench workload at any size,
with the CLI's phase and index-build timing and the ingest pipeline's own account; a few environments side by side.
   python tools/e2e_timing.py [bases=3e9] ["ENV=V ENV2=V2" ...]"""
import os
import shutil
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import merfin_amd as m
from tools import synth_torch as st
from tools import e2e_inputs

bases = int(float(sys.argv[1])) if len(sys.argv) > 1 else 3_000_000_000
envs = [dict(kv.split("=", 1) for kv in spec.split()) for spec in sys.argv[2:]] or [{}]
tmp = tempfile.mkdtemp(prefix="mfx_e2e_", dir="/dev/shm" if os.path.isdir("/dev/shm") else "/tmp")
try:
    inp = e2e_inputs.write_inputs(m, st, torch, bases, tmp, ncontigs=24, log=lambda s: print("  " + s, flush=True))
    print("inputs: %d bases in 24 contigs (FASTA %.2f GB), read database %d k-mers = %.2f GB delta-coded (%.2f B per k-mer); written in %.1f s" %
          (bases, inp["fasta_bytes"] / 1e9, inp["read_kmers"], inp["db_bytes"] / 1e9, inp["db_bytes"] / inp["read_kmers"], inp["write_s"]), flush=True)
    prob = os.path.join(ROOT, "tests", "golden", "example_lookup_table.txt")
    ref = None
    time.sleep(3)                                               # let the writer's threads and the driver settle: the first run is the figure of a quiet box
    for rep in range(2):
        for env in envs:
            time.sleep(3)
            rc, wall, ph, err = e2e_inputs.run_cli_hist(ROOT, inp, prob=prob, out_hist=os.path.join(tmp, "o.hist"), env=dict(env, MFX_INGEST_TIMING="1"))
            h = open(os.path.join(tmp, "o.hist")).read() if rc == 0 else None
            ref = ref or h
            print("%-40s rc=%d wall %.2f s  %s  same_hist=%s" % (" ".join("%s=%s" % kv for kv in env.items()) or "(default)", rc, wall,
                  "  ".join("%s %.2f" % kv for kv in ph.items()), h == ref), flush=True)
            for l in err.splitlines():
                if l.startswith("-- timing (index)") or l.startswith("-- ingest"):
                    print("      " + l)
finally:
    shutil.rmtree(tmp, ignore_errors=True)
