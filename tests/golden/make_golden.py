#!/usr/bin/env python3
"""Generates the committed golden fixtures of this directory with the CPU
oracle (oracle/merfin_oracle.c + merfin_oracle_variants.cpp):

  case1.fasta            3 small contigs (repeats, N runs, lower case)
  case1.read.kmers.txt   read k-mer counts,  `meryl print` text  (k = 21)
  case1.asm.kmers.txt    assembly k-mer counts
  case1.vcf              variant calls against case1.fasta
  case1.hist / .summary  expected -hist output      (-peak 17.3 -prob example_lookup_table.txt)
  case1.dump             expected -dump output
  case1.polish.vcf / case1.filter.vcf / case1.loose.vcf   expected variant-mode outputs (-comb 8)

The fixtures are DATA (inputs + expected outputs).  They pin both sides: the
oracle must keep reproducing them (tests/test_golden.py, CPU) and the HIP path
must match them byte for byte (GPU).  Re-run only when the expected behaviour
is meant to change:  python tests/golden/make_golden.py
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

from oracle import pyoracle as po  # noqa: E402
from tests import synth  # noqa: E402

K, PEAK, COMB = 21, 17.3, 8


def write_kmers(path, k, kmers, values):
    dec = np.array(list(b"ACTG"), dtype=np.uint8)
    with open(path, "w") as f:
        for km, v in zip(kmers.tolist(), values.tolist()):
            f.write("%s\t%d\n" % (bytes(dec[[(km >> (2 * (k - 1 - i))) & 3 for i in range(k)]]).decode(), v))


def main():
    names, asm, vcf, read, amers = synth.variant_world(k=K, peak=PEAK, seed=20260928, sizes=(4000, 1500, 120), decoys=25)
    # decorate the assembly a little (N run, lower case) without moving coordinates
    a0 = bytearray(asm[0])
    a0[3000:3010] = b"N" * 10
    a0[3500:3560] = bytes(a0[3500:3560]).lower()
    asm = [bytes(a0)] + list(asm[1:])
    # a fourth contig duplicating part of the first (asmK = 2 there: `undr` bins, koverCpy > 0) ...
    names = list(names) + ["ctg3_dup"]
    asm.append(asm[0][500:1500] + asm[1][100:400][::-1])
    ak, av = po.count_kmers(K, asm)
    # ... and some read k-mers at 2x / 3x / 0.4x coverage (`over` bins, low-count rows of the -prob table)
    rk, rv = read
    rv = rv.copy()
    sel = (rk % 11 == 0)
    rv[sel] = rv[sel] * 2
    sel = (rk % 37 == 0)
    rv[sel] = rv[sel] * 3
    sel = (rk % 23 == 0)
    rv[sel] = np.maximum(1, (rv[sel] * 0.4).astype(np.uint32))
    read = (rk, rv)
    p = lambda n: os.path.join(HERE, n)
    with open(p("case1.fasta"), "wb") as f:
        for n, c in zip(names, asm):
            f.write(b">" + n.encode() + b" golden case 1\n")
            for o in range(0, len(c), 70):
                f.write(c[o:o + 70] + b"\n")
    write_kmers(p("case1.read.kmers.txt"), K, *read)
    write_kmers(p("case1.asm.kmers.txt"), K, ak, av)
    open(p("case1.vcf"), "w").write(vcf)
    probK, probP = po.load_kmetric(p("example_lookup_table.txt"))
    prm = po.Params(K, PEAK, probK, probP)
    R, A = po.Lookup(K, *read), po.Lookup(K, ak, av)
    g, ka, km, _ = po.hist_run(prm, R, A, asm, threads=1)
    po.report_histogram(prm, g, p("case1.hist"), p("case1.summary"))
    for i, (n, c) in enumerate(zip(names, asm)):
        rk, akk, kmm, _, _ = po.process_dump(prm, R, A, c)
        po.output_dump(p("case1.dump"), n, rk, akk, kmm, append=i > 0)
    for mode in ("polish", "filter", "loose"):
        po.variants_run(prm, R, A, mode, p("case1.vcf"), names, asm, p("case1.%s.vcf" % mode), comb=COMB)
    for f in sorted(os.listdir(HERE)):
        if f.startswith("case1"):
            print("%-28s %8d bytes" % (f, os.path.getsize(p(f))))


if __name__ == "__main__":
    main()
