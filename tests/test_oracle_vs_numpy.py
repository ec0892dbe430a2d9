"""Second, independent restatement of the reference formulas in plain Python
(IEEE doubles, dict-based k-mer counts, string reverse complement) checked
against the C oracle on random small inputs.  Two restatements written in
different styles agreeing bit-for-bit is the strongest pin available here:
the reference itself cannot be built (see oracle/merfin_oracle.h)."""
import math

import numpy as np
import pytest

from oracle import pyoracle as po

COMP = {"A": "T", "C": "G", "G": "C", "T": "A"}
CODE = {"A": 0, "C": 1, "T": 2, "G": 3}


def enc(s):
    v = 0
    for ch in s:
        v = (v << 2) | CODE[ch]
    return v


def revcomp(s):
    return "".join(COMP[c] for c in reversed(s))


def c_round(x):          # C round(): half away from zero (Python's round() is half-to-even)
    return math.floor(x + 0.5) if x >= 0 else -math.floor(-x + 0.5)


def getK(peak, probK, probP, readV, asmV):           # merfin-globals.C:66-98
    readK, prob = 0.0, 1.0
    if readV == 0:
        readK = 0.0
    elif readV < peak:
        readK = 1.0
    else:
        readK = float(c_round(readV / peak))
    if 0 < readV <= len(probK):
        readK, prob = float(probK[readV - 1]), float(probP[readV - 1])
    return readK, float(asmV), prob


def kmetric(readK, asmK):                            # merfin-globals.H:248-261
    if readK == 0:
        return 0.0
    if asmK > readK:
        return (asmK / readK - 1) * -1
    if asmK < readK:
        return readK / asmK - 1
    return 0.0


def py_hist(k, peak, probK, probP, contig, R, A):    # merfin-histogram.C:54-91
    undr, over = {}, {}
    kasm = kmissing = 0
    kover = 0.0
    s = contig.upper()
    for i in range(len(s) - k + 1):
        w = s[i:i + k]
        if any(c not in "ACGT" for c in w):
            continue
        kasm += 1
        f, r = enc(w), enc(revcomp(w))
        readV = (R.get(f, 0) + R.get(r, 0)) & 0xffffffff
        asmV = (A.get(f, 0) + A.get(r, 0)) & 0xffffffff
        readK, asmK, prob = getK(peak, probK, probP, readV, asmV)
        if readK == 0:
            kmissing += 1
            continue
        if asmK > readK:
            idx = int(((asmK / readK - 1) + 0.1) / 0.2)
            undr[idx] = undr.get(idx, 0) + 1
            kover += (1.0 - readK / asmK) * prob
        else:
            idx = int(((readK / asmK - 1) + 0.1) / 0.2)
            over[idx] = over.get(idx, 0) + 1
    return undr, over, kasm, kmissing, kover


@pytest.mark.parametrize("k,seed,use_prob", [(5, 1, False), (8, 2, True), (21, 3, False), (21, 4, True), (31, 5, False), (6, 6, False)])
def test_c_oracle_equals_python_restatement(k, seed, use_prob, golden_dir):
    r = np.random.default_rng(seed)
    n = 3000
    seq = "".join(r.choice(list("ACGT"), size=n))
    # repeats, N runs, lower case
    seq = seq[:500] + seq[100:400] + seq[500:900] + "NNNN" + seq[900:1500].lower() + seq[1500:1520] * 15 + "n" + seq[1520:]
    contigs = [seq, seq[50:700], "ACGT", ""]
    ak, av = po.count_kmers(k, [c.encode() for c in contigs])
    rv = r.poisson(7.0 * av).astype(np.uint32)
    if use_prob:
        probK, probP = po.load_kmetric(golden_dir + "/example_lookup_table.txt")
        rv = (rv * 3).astype(np.uint32)          # reach into the table's 9+ rows
    else:
        probK, probP = np.zeros(0, dtype=np.uint32), np.zeros(0)
    keep = rv > 0
    R = dict(zip(ak[keep].tolist(), rv[keep].tolist()))
    A = dict(zip(ak.tolist(), av.tolist()))
    peak = 7.3
    p = po.Params(k, peak, probK, probP)
    RL, AL = po.Lookup(k, ak[keep], rv[keep]), po.Lookup(k, ak, av)
    for c in contigs:
        undr, over, kasm, kmissing, kover = py_hist(k, peak, probK.tolist(), probP.tolist(), c, R, A)
        h = po.process_histogram(p, RL, AL, c.encode())
        assert (h.kasm, h.kmissing) == (kasm, kmissing)
        assert h.koverCpy == kover                   # same order of the same IEEE operations: bit-identical
        cu, co = h.undr(), h.over()
        assert {i: int(v) for i, v in enumerate(cu) if v} == undr
        assert {i: int(v) for i, v in enumerate(co) if v} == over
        # -dump triples (merfin-dump.C:44-67)
        rk, ak_, km_, dk, dm = po.process_dump(p, RL, AL, c.encode())
        s = c.upper()
        for i in range(max(len(s) - k + 1, 0)):
            w = s[i:i + k]
            if any(ch not in "ACGT" for ch in w):
                assert (rk[i], ak_[i], km_[i]) == (0, 0, 0)
                continue
            f, rr = enc(w), enc(revcomp(w))
            a, b, _ = getK(peak, probK.tolist(), probP.tolist(), (R.get(f, 0) + R.get(rr, 0)), A.get(f, 0) + A.get(rr, 0))
            assert (rk[i], ak_[i], km_[i]) == (a, b, kmetric(a, b))
    assert kasm == 0                                 # the empty contig came last


def test_counter_and_canonical_against_python():
    r = np.random.default_rng(9)
    seq = "".join(r.choice(list("ACGT"), size=700)) + "N" + "".join(r.choice(list("acgt"), size=300))
    for k in (4, 7, 21):
        d = {}
        s = seq.upper()
        for i in range(len(s) - k + 1):
            w = s[i:i + k]
            if "N" in w:
                continue
            c = min(enc(w), enc(revcomp(w)))
            d[c] = d.get(c, 0) + 1
        kk, vv = po.count_kmers(k, [seq.encode()])
        assert dict(zip(kk.tolist(), vv.tolist())) == d
        for km in list(d)[:50]:
            assert po.lib().orc_canonical(km, k) == km
