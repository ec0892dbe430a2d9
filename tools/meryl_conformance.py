#!/usr/bin/env python3
"""Conformance check of the meryl-directory decoder (merfin_amd/csrc/mfx_db.cpp) against a REAL database.

  python tools/meryl_conformance.py <db.meryl> <meryl-print.txt[.gz]>      (on a box with an MI355X)

<meryl-print.txt> is the output of `meryl print <db.meryl>` (k-mer <TAB> count per line), produced by upstream meryl.
Both are loaded through the library (mfx_db_probe + mfx_index_load_db -> device table -> mfx_index_export) and the
two (k-mer, count) sets are compared exactly.  This is SURVEY.md App. C's conformance plan, steps (1)-(3):
  (1) the master index opens, its magics / k are accepted;
  (2) all 64 files decode: strictly increasing k-mers, file number == top 6 prefix bits, sum of the block headers ==
      the statistics' distinct count, value sum == total, ones == unique (the decoder enforces these on every load);
  (3) the decoded set equals `meryl print` of the same database.
No meryl database or meryl binary exists in the build container, so this has never run against real data: until it
has, the directory decoder stays labelled UNVALIDATED and `meryl print` text is the verified ingest."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def load(m, path, k):
    info = m.db_probe(path)
    ix = m.Index(k or info["k"], info["n_kmers"] + 1024)
    ix.load_db(path, 0)
    ek, er, _ = ix.export()
    o = np.argsort(ek)
    return info, ek[o], er[o]


def main():
    if len(sys.argv) != 3:
        print(__doc__)
        return 2
    import merfin_amd as m
    di, dk, dv = load(m, sys.argv[1], 0)
    print("directory: format %s  k=%d  %d k-mers (block headers)  %d decoded" % (di["format"], di["k"], di["n_kmers"], len(dk)))
    ti, tk, tv = load(m, sys.argv[2], di["k"])
    print("text     : format %s  k=%d  %d k-mers" % (ti["format"], ti["k"], len(tk)))
    ok = di["format"] == "meryl" and di["k"] == ti["k"] and len(dk) == len(tk) == di["n_kmers"]
    if ok:
        bad_k = np.nonzero(dk != tk)[0]
        bad_v = np.nonzero(dv != tv)[0]
        ok = len(bad_k) == 0 and len(bad_v) == 0
        if len(bad_k):
            print("first differing k-mer at rank %d: directory %x, text %x" % (bad_k[0], dk[bad_k[0]], tk[bad_k[0]]))
        elif len(bad_v):
            print("first differing count at k-mer %x: directory %d, text %d" % (dk[bad_v[0]], dv[bad_v[0]], tv[bad_v[0]]))
    print("CONFORMANT" if ok else "MISMATCH")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
