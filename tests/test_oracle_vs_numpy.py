"""Second, independent restatement of the reference formulas in plain Python
(IEEE doubles, dict-based k-mer counts, string reverse complement) checked
against the C oracle on random small inputs.  Two restatements written in
different styles agreeing bit-for-bit is the strongest pin available here:
the reference itself cannot be built (see oracle/merfin_oracle.h)."""
import numpy as np
import pytest

from oracle import pyoracle as po

from oracle.plain import enc, revcomp, getK, kmetric, py_hist          # the plain-Python restatement lives in oracle/plain.py


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


def test_plain_completeness_and_filter_against_c_oracle():
    """the remaining pieces of oracle/plain.py (used as THE oracle for 32 <= k <= 64): per-piece completeness sums and the
    -min/-max read filter, against the C oracle at k = 21"""
    from oracle import plain
    from tests import synth
    k, peak = 21, 17.3
    contigs, read, asm = synth.world(k=k, peak=peak, seed=23, sizes=(6000, 2000, 300))
    R = dict(zip(read[0].tolist(), read[1].tolist()))
    A = dict(zip(asm[0].tolist(), asm[1].tolist()))
    p = po.Params(k, peak)
    wt, wu = plain.py_completeness(k, peak, [], [], R, A)
    for piece in range(64):
        lo, hi = piece << (2 * k - 6), (piece + 1) << (2 * k - 6)
        rs, as_ = (read[0] >= lo) & (read[0] < hi), (asm[0] >= lo) & (asm[0] < hi)
        assert (wt[piece], wu[piece]) == po.completeness_piece(p, read[0][rs], read[1][rs], asm[0][as_], asm[1][as_])
    RL, AL = po.Lookup(k, read[0], read[1], 4, 40), po.Lookup(k, *asm)
    for c in contigs:
        undr, over, kasm, kmissing, kover = plain.py_hist(k, peak, [], [], c.decode(), R, A, 4, 40)
        h = po.process_histogram(p, RL, AL, c)
        assert (h.kasm, h.kmissing, h.koverCpy) == (kasm, kmissing, kover)
        assert {i: int(v) for i, v in enumerate(h.undr()) if v} == undr and {i: int(v) for i, v in enumerate(h.over()) if v} == over
    assert plain.count_kmers(k, [c.decode() for c in contigs]) == A
    assert plain.dec(plain.enc("ACGTTGCA"), 8) == "ACGTTGCA"


@pytest.mark.parametrize("mode,k,seed", [("polish", 21, 11), ("filter", 15, 12), ("loose", 21, 13), ("strict", 31, 14), ("better", 11, 15)])
def test_variants_with_injected_text_lookups_equal_the_kmer_iterator_form(mode, k, seed, tmp_path):
    """orc_variants_run_cb -- varMer::score's lookups supplied as a function of the k-mer TEXT, here the plain-Python getK
    over dict counts -- must reproduce orc_variants_run (kmerIterator + merylExactLookup restatement) byte for byte: it is
    the form that pins the variant modes at 32 <= k <= 64, where the C oracle's 64-bit k-mers end"""
    from oracle import plain
    from tests import synth
    peak = 17.3
    names, asm, vcf, read, amers = synth.variant_world(k=k, peak=peak, seed=seed, sizes=(6000, 2500, 300))
    vp = str(tmp_path / "in.vcf")
    open(vp, "w").write(vcf)
    p = po.Params(k, peak)
    n0 = po.variants_run(p, po.Lookup(k, *read), po.Lookup(k, *amers), mode, vp, names, asm, str(tmp_path / "a.vcf"), comb=9,
                         debug_path=str(tmp_path / "a.dbg"), log_path=str(tmp_path / "a.log"))
    R = dict(zip(read[0].tolist(), read[1].tolist()))
    A = dict(zip(amers[0].tolist(), amers[1].tolist()))

    def getk(text):
        rv, av = plain.values(k, text.upper(), R, A)
        return plain.getK(peak, [], [], rv, av)

    n1 = po.variants_run_text(k, getk, mode, vp, names, asm, str(tmp_path / "b.vcf"), comb=9, debug_path=str(tmp_path / "b.dbg"),
                              log_path=str(tmp_path / "b.log"))
    assert n0 == n1 and n0 > 10
    for ext in ("vcf", "dbg", "log"):
        assert (tmp_path / ("a." + ext)).read_bytes() == (tmp_path / ("b." + ext)).read_bytes(), ext
