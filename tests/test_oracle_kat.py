"""Pins the CPU oracle against hand-derived IEEE-754 known-answer values for
every in-tree formula of the reference and against the
reference's one data fixture (the 184-row -prob table).  The reference ships
no tests/golden vectors and cannot be compiled here, so this table is the pin
(see oracle/merfin_oracle.h, "parity unpinned")."""
import math
import os

import numpy as np
import pytest

from oracle import pyoracle as po


# (hi, lo) -> idx, SURVEY.md Appendix B.  Bold rows are the FP traps.
BIN_KAT = [
    (1, 1, 0), (2, 1, 5), (3, 1, 10), (4, 1, 15), (11, 1, 50), (4, 2, 5), (5, 2, 8), (7, 2, 13),
    (3, 2, 2), (5, 4, 1), (6, 5, 1), (7, 5, 2), (8, 5, 3), (9, 5, 4), (4, 3, 2), (5, 3, 3), (7, 4, 4),
    (10, 9, 1), (11, 10, 1), (13, 10, 2), (17, 10, 3), (19, 10, 4),
]
BIN_KAT += [(h, 3, e) for h, e in zip(range(5, 15), [3, 5, 7, 8, 10, 12, 13, 15, 17, 18])]
BIN_KAT += [(h, 4, e) for h, e in zip(range(5, 16), [1, 2, 4, 5, 6, 8, 9, 10, 11, 13, 14])]
BIN_KAT += [(h, 5, e) for h, e in zip(range(6, 17), [1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11])]


def _hist_for_pair(readK, asmK, k=5):
    """Build a 1-contig world where the single k-mer has read copy number readK
    (via a -prob table row) and assembly count asmK, then run processHistogram."""
    seq = b"ACGTT"[:k]
    fm = po.lib().orc_encode(seq, k, None)
    can = po.lib().orc_canonical(fm, k)
    probK = np.array([readK], dtype=np.uint32)   # readV=1 -> readK
    probP = np.array([0.5], dtype=np.float64)
    p = po.Params(k, 30.0, probK, probP)
    R = po.Lookup(k, [can], [1])
    A = po.Lookup(k, [can], [asmK])
    return po.process_histogram(p, R, A, seq), p


@pytest.mark.parametrize("hi,lo,idx", BIN_KAT)
def test_bin_index_over(hi, lo, idx):
    # readK = hi > asmK = lo  -> over[idx]
    h, _ = _hist_for_pair(hi, lo)
    over = h.over()
    assert h.kasm == 1 and h.kmissing == 0
    assert over[idx] == 1 and over.sum() == 1 and h.undr().sum() == 0
    assert h.koverCpy == 0.0


@pytest.mark.parametrize("hi,lo,idx", [t for t in BIN_KAT if t[0] != t[1]])
def test_bin_index_undr(hi, lo, idx):
    # asmK = hi > readK = lo -> undr[idx], koverCpy += (1 - readK/asmK) * prob
    h, _ = _hist_for_pair(lo, hi)
    undr = h.undr()
    assert undr[idx] == 1 and undr.sum() == 1 and h.over().sum() == 0
    assert h.koverCpy == (1.0 - lo / hi) * 0.5


def test_equal_lands_in_over0():
    h, _ = _hist_for_pair(3, 3)
    assert h.over()[0] == 1 and h.undr().sum() == 0


def test_missing_not_binned():
    h, _ = _hist_for_pair(0, 4)           # present in reads, table says readK=0 -> missing
    assert h.kasm == 1 and h.kmissing == 1
    assert h.over().sum() == 0 and h.undr().sum() == 0


def test_readK_peak_rounding():
    p = po.Params(21, 30.0)
    got = [po.getK_values(p, v, 7)[0] for v in [0, 1, 29, 30, 44, 45, 46, 74, 75, 76, 105]]
    assert got == [0, 1, 1, 1, 1, 2, 2, 2, 3, 3, 4]
    p = po.Params(21, 10.5)
    got = [po.getK_values(p, v, 7)[0] for v in [10, 11, 15, 16, 26, 27]]
    assert got == [1, 1, 1, 2, 2, 3]
    # asmK is the raw value, prob defaults to 1
    assert po.getK_values(p, 11, 7)[1:] == (7.0, 1.0)


def test_prob_table_fixture(golden_dir):
    K, P = po.load_kmetric(os.path.join(golden_dir, "example_lookup_table.txt"))
    assert len(K) == 184
    # multiplicities 1-8 -> 0, 9-43 -> 1, 44-131 -> 2, 132-184 -> 4 (no 3)
    assert list(K[:8]) == [0] * 8 and set(K[8:43]) == {1} and set(K[43:131]) == {2} and set(K[131:]) == {4}
    p = po.Params(21, 30.0, K, P)
    assert po.getK_values(p, 5, 1) == (0.0, 1.0, 0.957157982701154)
    assert po.getK_values(p, 9, 1) == (1.0, 1.0, 0.683560785280716)
    assert po.getK_values(p, 184, 2) == (4.0, 2.0, 0.987488957253625)
    # outside the table: peak rule, prob 1
    assert po.getK_values(p, 185, 2) == (6.0, 2.0, 1.0)      # round(185/30) = 6
    assert po.getK_values(p, 0, 2) == (0.0, 2.0, 1.0)


def test_prob_table_invalid_lines(tmp_path):
    f = tmp_path / "t.csv"
    f.write_text("1,0.5\nbad line\n2,0.25,9\n\n3,0.125\n")
    K, P = po.load_kmetric(str(f))
    assert list(K) == [1, 3] and list(P) == [0.5, 0.125]


def test_histoQV():
    assert po.histoQV(100, 1e6, 21) == pytest.approx(53.22198613193983, abs=1e-12)
    assert "%.2f" % po.histoQV(100, 1e6, 21) == "53.22"
    assert po.histoQV(0, 1e6, 21) == math.inf


def test_getKmetric():
    assert po.getKmetric(0, 5) == 0
    assert po.getKmetric(2, 3) == -(3 / 2 - 1)
    assert po.getKmetric(3, 2) == 3 / 2 - 1
    assert po.getKmetric(4, 4) == 0


def test_base_codes_and_iterator():
    L = po.lib()
    for ch, c in zip(b"ACTGactg", [0, 1, 2, 3, 0, 1, 2, 3]):
        assert L.orc_base_code(ch) == c
    for ch in b"NnXRY-*\0>":
        assert L.orc_base_code(ch) == -1
    seq = b"ACGTNACGTAC"
    got = list(po.kiter(3, seq))
    # valid 3-mers start at 0,1 then (after N at 4) 5..8
    assert [g[0] for g in got] == [0, 1, 5, 6, 7, 8]
    f0 = (0 << 4) | (1 << 2) | 3           # ACG
    assert got[0][1] == f0
    # revcomp(ACG) = CGT
    assert got[0][2] == (1 << 4) | (3 << 2) | 2
    assert L.orc_revcomp(f0, 3) == got[0][2]
    # lower case is the same k-mer
    assert list(po.kiter(3, b"acgtnacgtac")) == got


def test_lookup_semantics():
    k = 7
    rng = np.random.default_rng(1)
    kmers = np.unique(rng.integers(0, 4 ** k, size=2000, dtype=np.uint64))
    vals = rng.integers(1, 50, size=len(kmers), dtype=np.uint32)
    for pb in (0, 1, 5, 14):
        Lk = po.Lookup(k, kmers, vals, prefix_bits=pb)
        d = dict(zip(kmers.tolist(), vals.tolist()))
        for q in range(0, 4 ** k, 7):
            assert Lk.value(q) == d.get(q, 0)
    # -min / -max drop at load (merfin.C:199-200)
    Lk = po.Lookup(k, kmers, vals, minV=10, maxV=20)
    for km, v in zip(kmers.tolist(), vals.tolist()):
        assert Lk.value(km) == (v if 10 <= v <= 20 else 0)


def test_hist_report_format(tmp_path):
    k = 5
    seq = b"ACGTTGCATGCCGATAGCTAGCTAGGATCNNACGTTGCA"
    ak, av = po.count_kmers(k, [seq])
    rv = (av * 3 + np.arange(len(av)) % 4).astype(np.uint32)
    p = po.Params(k, 3.0)
    R, A = po.Lookup(k, ak, rv), po.Lookup(k, ak, av)
    g, ka, km, _ = po.hist_run(p, R, A, [seq])
    hp, sp = tmp_path / "h", tmp_path / "s"
    po.report_histogram(p, g, str(hp), str(sp))
    rows = [l.split("\t") for l in hp.read_text().splitlines()]
    keys = [float(r[0]) for r in rows]
    assert keys == sorted(keys) and "0.0" in [r[0] for r in rows]
    assert sum(int(r[1]) for r in rows) == g.kasm - g.kmissing
    s = sp.read_text()
    assert "K-mers not found in reads (missing) : %d\n" % g.kmissing in s
    assert "K-mers found in the assembly: %d\n" % g.kasm in s
    assert "Merfin QV*: " in s and "Missing QV: " in s
    assert int(ka[0]) == g.kasm == len(seq) - 2 - 2 * (k - 1)
