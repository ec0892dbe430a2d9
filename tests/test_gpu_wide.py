"""32 <= k <= 64: k-mers of up to 128 bits (the reference's kmer type holds 2k <= 128 bits: merfin-globals.C:183,
varMer.C:108).  The C oracle's k-mers are 64-bit, so the oracle here is the plain-Python restatement (oracle/plain.py:
arbitrary-precision integers, dict counts, string reverse complement), itself pinned bit for bit against the C oracle
at k <= 31 (tests/test_oracle_vs_numpy.py).  Bar as everywhere: integers exact, koverCpy to 1e-12.
At the C ABI a k-mer of k > 31 is two uint64 words {low, high}: numpy rows [lo, hi]."""
import os

import numpy as np
import pytest

from oracle import plain
from tests import meryl_layout

pytestmark = pytest.mark.gpu


def to_rows(ints):
    a = np.zeros((len(ints), 2), dtype=np.uint64)
    for i, x in enumerate(ints):
        a[i, 0] = x & 0xffffffffffffffff
        a[i, 1] = x >> 64
    return a


def small_world(k, seed, n=9000):
    """contigs (str) with repeats, N runs, lower case; read counts R and assembly counts A as {canonical k-mer: count}"""
    r = np.random.default_rng(seed)
    seq = "".join(r.choice(list("ACGT"), size=n))
    truth = seq[:2500] + seq[300:1200] + seq[2500:5000] + seq[2500:2600] * 6 + seq[5000:]
    asm = truth[:1000] + "NNNNN" + truth[1005:3000].lower() + truth[3000:3400] + truth[3000:3400] + truth[3400:4094] + "N" + truth[4095:]
    asm = asm[:7000] + ("T" if asm[7000] != "T" else "G") + asm[7001:]          # a substitution error: k missing k-mers
    contigs = [asm, asm[100:100 + k], asm[200:200 + k - 1], "", "G" * (k + 5), truth[4000:4000 + 4096 + k]]
    T = plain.count_kmers(k, [truth])
    R = {}
    for x, c in T.items():
        v = int(r.poisson(9.0 * c))
        if v:
            R[x] = v
    for _ in range(300):                                      # error k-mers (random, low count)
        w = "".join(r.choice(list("ACGT"), size=k))
        x = min(plain.enc(w), plain.enc(plain.revcomp(w)))
        R.setdefault(x, int(r.integers(1, 4)))
    A = plain.count_kmers(k, contigs)
    return contigs, R, A


def build(m, k, R, A):
    ix = m.Index(k, len(R) + len(A) + 16)
    rk = sorted(R)
    ak = sorted(A)
    if k > 31:
        ix.add_read(to_rows(rk), np.array([R[x] for x in rk], dtype=np.uint32))
        ix.add_asm(to_rows(ak), np.array([A[x] for x in ak], dtype=np.uint32))
    else:
        ix.add_read(np.array(rk, dtype=np.uint64), np.array([R[x] for x in rk], dtype=np.uint32))
        ix.add_asm(np.array(ak, dtype=np.uint64), np.array([A[x] for x in ak], dtype=np.uint32))
    return ix


def assert_hist(res, contigs, k, peak, probK, probP, R, A):
    undr, over, kasm, kmis, kover, ck, cm = {}, {}, 0, 0, 0.0, [], []
    for c in contigs:                                         # outputHistogram: per-contig results added in order
        u, o, ka, km, kv = plain.py_hist(k, peak, probK, probP, c, R, A)
        for d, s in ((undr, u), (over, o)):
            for i, v in s.items():
                d[i] = d.get(i, 0) + v
        kasm += ka
        kmis += km
        kover += kv
        ck.append(ka)
        cm.append(km)
    assert (res.kasm, res.kmissing) == (kasm, kmis)
    assert {i: int(v) for i, v in enumerate(res.undr()) if v} == undr
    assert {i: int(v) for i, v in enumerate(res.over()) if v} == over
    assert res.contig_kasm().tolist() == ck and res.contig_kmissing().tolist() == cm
    assert res.koverCpy == pytest.approx(kover, rel=1e-12, abs=1e-9)
    return kasm, kmis, kover


@pytest.mark.parametrize("k,use_prob", [(32, False), (41, False), (41, True), (55, False), (63, False), (64, False), (31, False)])
def test_hist_dump_completeness_match_plain_oracle(k, use_prob, golden_dir):
    import merfin_amd as m
    from oracle import pyoracle as po
    contigs, R, A = small_world(k, 400 + k)
    peak = 9.0
    probK, probP = ([], [])
    if use_prob:
        K, P = po.load_kmetric(os.path.join(golden_dir, "example_lookup_table.txt"))
        probK, probP = K.tolist(), P.tolist()
    ix = build(m, k, R, A)
    info = ix.info()
    assert info["k"] == k and info["distinct"] == len(set(R) | set(A)) and info["canonical"]
    ev = m.Evaluator(ix, m.KParams(peak, probK or None, probP or None))
    seqs = m.Sequences([c.encode() for c in contigs])
    kasm, kmis, kover = assert_hist(ev.hist(seqs), contigs, k, peak, probK, probP, R, A)
    assert kasm > 8000 and kmis >= k and kover > 0
    # the streamed upload and the several-slots reduction are key-width agnostic
    assert_hist(ev.hist_streamed(m.Sequences.create([len(c) for c in contigs]), [c.encode() for c in contigs]), contigs, k, peak, probK, probP, R, A)
    assert_hist(m.hist_multi([ev, m.Evaluator(ix, m.KParams(peak, probK or None, probP or None))], [seqs, seqs]), contigs, k, peak, probK, probP, R, A)
    # -dump raw values per start position (merfin-dump.C:44-67)
    for ci in (0, 1, 5):
        rv, av, ka, km = ev.dump_values(seqs, ci, 0, len(contigs[ci]))
        want = plain.py_dump(k, peak, probK, probP, contigs[ci], R, A)
        assert ka == len(want) and km == sum(1 for v in want.values() if v[2] == 0)
        for i in range(len(contigs[ci])):
            assert (int(rv[i]), int(av[i])) == (want[i][:2] if i in want else (0, 0)), (ci, i)
    # value(kmer) and the exported table
    some = sorted(R)[::7]
    keys = to_rows(some) if k > 31 else np.array(some, dtype=np.uint64)
    got_r, got_a = ix.value(keys)
    assert got_r.tolist() == [R[x] for x in some] and got_a.tolist() == [A.get(x, 0) for x in some]
    ek, er, ea = ix.export()
    allk = sorted(set(R) | set(A))
    assert (meryl_layout.to_ints(ek) if k > 31 else ek.tolist()) == allk
    assert er.tolist() == [R.get(x, 0) for x in allk] and ea.tolist() == [A.get(x, 0) for x in allk]
    # -completeness: the reference's 64 per-piece sums (merfin-completeness.C:70-123)
    t64, u64 = ev.completeness_pieces()
    wt, wu = plain.py_completeness(k, peak, probK, probP, R, A)
    assert t64.tolist() == wt and u64.tolist() == wu and sum(wt) > 0


@pytest.mark.parametrize("k", [33, 41, 64])
def test_assembly_counter_and_two_strand_path(k, monkeypatch):
    """mfx_index_count_asm (`meryl count` of -sequence) at k > 31, and value(f) + value(r) when the database is not
    canonical / k is even (merfin-globals.C:107-108)"""
    import merfin_amd as m
    contigs, R, A = small_world(k, 500 + k, n=6000)
    seqs = m.Sequences([c.encode() for c in contigs])
    ix = m.Index(k, len(A) + len(R) + 16)
    rk = sorted(R)
    ix.add_read(to_rows(rk), np.array([R[x] for x in rk], dtype=np.uint32))
    ix.count_asm(seqs)
    ek, er, ea = ix.export()
    got = {x: int(v) for x, v in zip(meryl_layout.to_ints(ek), ea.tolist()) if v}
    assert got == A
    ev = m.Evaluator(ix, m.KParams(9.0))
    assert_hist(ev.hist(seqs), contigs, k, 9.0, [], [], R, A)
    monkeypatch.setenv("MFX_FORCE_TWO_STRAND", "1")           # both strands probed and summed: same answer for a canonical DB
    assert_hist(m.Evaluator(ix, m.KParams(9.0)).hist(seqs), contigs, k, 9.0, [], [], R, A)
    monkeypatch.delenv("MFX_FORCE_TWO_STRAND")
    # a NON-canonical database: forward-strand k-mers stored as they are
    fwd = {}
    for c in contigs[:1]:
        for _, w in plain.valid_kmers(c, k):
            fwd[plain.enc(w)] = fwd.get(plain.enc(w), 0) + 1
    ix2 = m.Index(k, len(fwd) + 16)
    fk = sorted(fwd)
    ix2.add_read(to_rows(fk), np.array([3 * fwd[x] for x in fk], dtype=np.uint32))
    ix2.add_asm(to_rows(fk), np.array([fwd[x] for x in fk], dtype=np.uint32))
    assert not ix2.info()["canonical"]
    R2 = {x: 3 * v for x, v in fwd.items()}
    assert_hist(m.Evaluator(ix2, m.KParams(9.0)).hist(m.Sequences([contigs[0].encode()])), contigs[:1], k, 9.0, [], [], R2, fwd)


def test_wide_database_forms_and_cli(tmp_path):
    """k = 41 through the three on-disk forms (flat binary, `meryl print` text, meryl-layout directory: the decoder's
    two-piece >64-bit suffixes) and end to end through the C++ CLI; histogram text vs the plain oracle's counts run
    through the same report writer."""
    import subprocess
    import merfin_amd as m
    k, peak = 41, 9.0
    contigs, R, A = small_world(k, 77, n=7000)
    rk, ak = sorted(R), sorted(A)
    rrows, rvals = to_rows(rk), np.array([R[x] for x in rk], dtype=np.uint32)
    flat, text, mdir = str(tmp_path / "r.mfxk"), str(tmp_path / "r.txt.gz"), str(tmp_path / "r.meryl")
    m.db_write_flat(flat, k, rrows, rvals)
    import gzip
    with gzip.open(text, "wt") as f:
        for x in rk:
            f.write("%s\t%d\n" % (plain.dec(x, k), R[x]))
    meryl_layout.write_db(mdir, k, rrows, rvals, prefix_bits=14)
    for path, fmt in ((flat, "flat"), (text, "text"), (mdir, "meryl")):
        assert m.db_probe(path) == {"k": k, "format": fmt, "n_kmers": len(rk)}
        ix = m.Index(k, len(rk) + 16)
        ix.load_db(path, 0)
        ek, er, _ = ix.export()
        assert meryl_layout.to_ints(ek) == rk and er.tolist() == rvals.tolist()
    # CLI: -hist with -seqmers as text, and with the assembly counted on the GPU
    fa = str(tmp_path / "a.fasta")
    with open(fa, "w") as f:
        for i, c in enumerate(contigs):
            f.write(">ctg%d\n%s\n" % (i, c))
    with open(tmp_path / "a.txt", "w") as f:
        for x in ak:
            f.write("%s\t%d\n" % (plain.dec(x, k), A[x]))
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "merfin_amd", "bin", "merfin")
    undr, over, kasm, kmis, kover = {}, {}, 0, 0, 0.0
    for c in contigs:
        u, o, ka, km, kv = plain.py_hist(k, peak, [], [], c, R, A)
        for d, s in ((undr, u), (over, o)):
            for i, v in s.items():
                d[i] = d.get(i, 0) + v
        kasm, kmis, kover = kasm + ka, kmis + km, kover + kv
    nb = 2048
    from merfin_amd import distributed as D
    img = D.pack_counts(nb, len(contigs), np.array([undr.get(i, 0) for i in range(nb)], dtype=np.uint64),
                        np.array([over.get(i, 0) for i in range(nb)], dtype=np.uint64), kasm, kmis, [0] * len(contigs), [0] * len(contigs))
    m.result_from_counts(nb, img, kover, len(contigs)).report(k, str(tmp_path / "o.hist"), str(tmp_path / "o.sum"))
    for extra in (["-seqmers", str(tmp_path / "a.txt")], []):
        r = subprocess.run([exe, "-hist", "-sequence", fa, "-readmers", text, "-peak", str(peak), "-output", str(tmp_path / "g.hist")] + extra,
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        assert (tmp_path / "g.hist").read_bytes() == (tmp_path / "o.hist").read_bytes()
        assert "K-mers found in the assembly: %d\n" % kasm in r.stderr and "K-mers not found in reads (missing) : %d\n" % kmis in r.stderr
    # a sharded index is a k <= 31 feature: refused, not silently wrong
    with pytest.raises(m.MfxError):
        m.Index(k, 100).set_shard(0, 2)


def test_wide_index_image_and_replica(tmp_path):
    """the device-format image (-index) and the device-to-device replica carry 32-byte slots unchanged"""
    import merfin_amd as m
    k = 47
    contigs, R, A = small_world(k, 91, n=6000)
    ix = build(m, k, R, A)
    img = str(tmp_path / "w.mfxi")
    ix.set_fingerprint(0xABCDEF)
    ix.save(img)
    for other in (m.Index.load(img), ix.replicate(0)):
        assert other.info() == ix.info() and other.origin() == ix.origin()
        for a, b in zip(other.export(), ix.export()):
            np.testing.assert_array_equal(a, b)
        seqs = m.Sequences([c.encode() for c in contigs])
        assert_hist(m.Evaluator(other, m.KParams(9.0)).hist(seqs), contigs, k, 9.0, [], [], R, A)
    # an image of a narrow table cannot be mistaken for a wide one
    ix31 = build(m, 31, *small_world(31, 92, n=6000)[1:])
    ix31.save(str(tmp_path / "n.mfxi"))
    assert m.Index.load(str(tmp_path / "n.mfxi")).info()["k"] == 31


@pytest.mark.parametrize("seed", list(range(int(os.environ.get("MFX_RANDOM_SEEDS", "6")))))
def test_randomized_wide_k_against_plain_oracle(seed, monkeypatch):
    """seeded sweep over 32 <= k <= 64 (and table fill, peak): -hist resident and streamed against the plain-Python oracle"""
    import merfin_amd as m
    r = np.random.default_rng(9000 + seed)
    k = int(r.integers(32, 65))
    peak = float(r.choice([2.5, 9.0, 26.0]))
    monkeypatch.setenv("MFX_LOAD_FACTOR", str(r.choice([0.3, 0.6, 0.9])))
    contigs, R, A = small_world(k, 9100 + seed, n=int(r.choice([5200, 7000, 9000])))
    ix = build(m, k, R, A)
    ev = m.Evaluator(ix, m.KParams(peak))
    seqs = m.Sequences([c.encode() for c in contigs])
    assert_hist(ev.hist(seqs), contigs, k, peak, [], [], R, A)
    assert_hist(ev.hist_streamed(m.Sequences.create([len(c) for c in contigs]), [c.encode() for c in contigs]), contigs, k, peak, [], [], R, A)


@pytest.mark.parametrize("mode,k,seed", [("polish", 33, 21), ("filter", 41, 22), ("better", 64, 23), ("strict", 41, 24), ("loose", 33, 25), ("polish", 64, 26)])
def test_variant_modes_with_128_bit_kmers(mode, k, seed, tmp_path):
    """-filter / -polish / -better / -strict / -loose at 32 <= k <= 64 (the reference's kmer holds 2k <= 128 bits:
    varMer.C:108): VCF and -debug text byte-identical to the C++ restatement of merfin-variants.C / varMer.C with its
    lookups supplied by the k-agnostic plain-Python getK (oracle.pyoracle.variants_run_text; that form equals the 64-bit
    k-mer-iterator form byte for byte at k <= 31: tests/test_oracle_vs_numpy.py)"""
    import merfin_amd as m
    from oracle import pyoracle as po
    from tests import synth
    peak = 9.0
    names, asm, vcf, truth = synth.variant_world(k=k, peak=peak, seed=seed, sizes=(6000, 2500, 300), tables=False)
    vp = str(tmp_path / "in.vcf")
    open(vp, "w").write(vcf)
    r = np.random.default_rng(seed)
    T = plain.count_kmers(k, [t.decode() for t in truth])
    R = {}
    for x, c in T.items():
        v = int(r.poisson(peak * c))
        if v:
            R[x] = v
    for _ in range(300):                                      # error k-mers
        w = "".join(r.choice(list("ACGT"), size=k))
        R.setdefault(min(plain.enc(w), plain.enc(plain.revcomp(w))), int(r.integers(1, 4)))
    A = plain.count_kmers(k, [a.decode() for a in asm])

    def getk(text):
        rv, av = plain.values(k, text.upper(), R, A)
        return plain.getK(peak, [], [], rv, av)

    n_o = po.variants_run_text(k, getk, mode, vp, names, asm, str(tmp_path / "o.vcf"), comb=9, debug_path=str(tmp_path / "o.dbg"),
                               log_path=str(tmp_path / "o.log"))
    ev = m.Evaluator(build(m, k, R, A), m.KParams(peak))
    n_g = ev.variants(mode, vp, names, asm, str(tmp_path / "g.vcf"), comb=9, debug_path=str(tmp_path / "g.dbg"), log_path=str(tmp_path / "g.log"))
    assert n_g == n_o and n_o > 10
    assert open(tmp_path / "g.vcf").read() == open(tmp_path / "o.vcf").read()
    assert open(tmp_path / "g.dbg").read() == open(tmp_path / "o.dbg").read()
    assert len([l for l in open(tmp_path / "g.vcf") if not l.startswith("#")]) > 10


def test_wide_index_is_not_sharded():
    """a sharded index handles k <= 31 (BASELINE config 5, the sharded one, uses k = 31): the 128-bit path refuses cleanly"""
    import merfin_amd as m
    ix = m.Index(41, 1000)
    with pytest.raises(m.MfxError, match="k <= 31"):
        ix.set_shard(0, 2)
    with pytest.raises(m.MfxError, match="k <= 31"):
        m.Router(ix, 2, 4)
