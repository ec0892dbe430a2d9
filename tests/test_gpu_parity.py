"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on the
same seeded inputs.  Bar: bit-exact for every integer (histogram bins, kasm,
kmissing, per-contig counts, dump values, index values); |rel| <= 1e-6 for
koverCpy / QV / QV* (north_star); in practice koverCpy agrees to ~1e-15."""
import os

import numpy as np
import pytest

from oracle import pyoracle as po
from tests import synth

pytestmark = pytest.mark.gpu

REL_TOL = 1e-6      # north_star tolerance for K*/QV* floats


def _mfx():
    import merfin_amd as m
    if m.device_count() < 1:
        pytest.fail("no HIP device visible: the GPU tests must run on the MI355X box")
    return m


def _trim(a):
    a = np.asarray(a)
    nz = np.nonzero(a)[0]
    return a[: (nz[-1] + 1 if len(nz) else 0)]


def build_index(m, k, read, asm, minV=0, maxV=2**64 - 1, cap=None):
    rk, rv = read
    ak, av = asm
    ix = m.Index(k, cap or (len(rk) + len(ak) + 16))
    ix.add_read(rk, rv, minV, maxV)
    ix.add_asm(ak, av)
    return ix


def oracle_hist(k, peak, contigs, read, asm, probK=None, probP=None, minV=0, maxV=2**64 - 1):
    p = po.Params(k, peak, probK, probP)
    R = po.Lookup(k, read[0], read[1], minV, maxV)
    A = po.Lookup(k, asm[0], asm[1])
    g, ka, km, _ = po.hist_run(p, R, A, contigs, threads=4, mode=0)
    return p, g, ka, km


def assert_hist_equal(res, g, ka, km, k, tight=True):
    assert res.kasm == g.kasm
    assert res.kmissing == g.kmissing
    np.testing.assert_array_equal(_trim(res.undr()), _trim(g.undr()))
    np.testing.assert_array_equal(_trim(res.over()), _trim(g.over()))
    np.testing.assert_array_equal(res.contig_kasm(), ka)
    np.testing.assert_array_equal(res.contig_kmissing(), km)
    assert res.koverCpy == pytest.approx(g.koverCpy, rel=REL_TOL, abs=1e-9)
    # tighter than the bar: only the summation order differs (tight=False: millions of terms near 1.0 -- the ORACLE's sequential
    # fp64 sum, the reference's own (merfin-histogram.C:81), is then itself ~1e-11 off the exact sum the device's integer sum gives)
    if tight:
        assert res.koverCpy == pytest.approx(g.koverCpy, rel=1e-12, abs=1e-9)
    if g.kasm:
        import merfin_amd as m
        qv_g = po.histoQV(g.kmissing, g.kasm, k)
        qv_r = m.histoQV(res.kmissing, res.kasm, k)
        assert qv_r == qv_g or (np.isinf(qv_r) and np.isinf(qv_g))
        qs_g = po.histoQV(g.kmissing + g.koverCpy, g.kasm, k)
        qs_r = m.histoQV(res.kmissing + res.koverCpy, res.kasm, k)
        assert qs_r == pytest.approx(qs_g, rel=REL_TOL) or (np.isinf(qs_r) and np.isinf(qs_g))


@pytest.mark.parametrize("k,peak,use_prob", [(21, 17.3, False), (21, 26.0, True), (31, 17.3, False), (15, 9.0, False)])
def test_hist_matches_oracle(k, peak, use_prob, golden_dir):
    m = _mfx()
    contigs, read, asm = synth.world(k=k, peak=peak)
    probK = probP = None
    if use_prob:
        probK, probP = po.load_kmetric(os.path.join(golden_dir, "example_lookup_table.txt"))
    p, g, ka, km = oracle_hist(k, peak, contigs, read, asm, probK, probP)
    ix = build_index(m, k, read, asm)
    assert ix.info()["canonical"]
    ev = m.Evaluator(ix, m.KParams(peak, probK, probP))
    res = ev.hist(m.Sequences(contigs))
    assert g.kasm > 40000 and g.kmissing > 0 and g.koverCpy > 0 and g.undr().sum() > 0
    assert_hist_equal(res, g, ka, km, k)


@pytest.mark.parametrize("mode,w", [("plain", "2"), ("mz", "2"), ("mz", "3"), ("mz", "5")])
def test_hist_matches_oracle_under_every_placement(mode, w, monkeypatch, golden_dir):
    """The index placement (plain k-mer hash / minimizer-keyed with w windows) is an
    internal layout choice: results must not depend on it."""
    m = _mfx()
    monkeypatch.setenv("MFX_HOME_MODE", mode)
    monkeypatch.setenv("MFX_MZ_W", w)
    for k, lf in ((21, "0.7"), (9, "0.9")):
        monkeypatch.setenv("MFX_LOAD_FACTOR", lf)
        contigs, read, asm = synth.world(k=k, peak=9.0, seed=71, err_kmers=3000 if k > 12 else 0)
        p, g, ka, km = oracle_hist(k, 9.0, contigs, read, asm)
        ix = build_index(m, k, read, asm, cap=len(np.union1d(read[0], asm[0])))
        res = m.Evaluator(ix, m.KParams(9.0)).hist(m.Sequences(contigs))
        assert_hist_equal(res, g, ka, km, k)
        rv, av = ix.value(asm[0][::5])
        np.testing.assert_array_equal(av, asm[1][::5])


def test_hist_report_text_identical(tmp_path, golden_dir):
    m = _mfx()
    k, peak = 21, 26.0
    contigs, read, asm = synth.world(k=k, peak=peak, seed=7)
    probK, probP = po.load_kmetric(os.path.join(golden_dir, "example_lookup_table.txt"))
    p, g, ka, km = oracle_hist(k, peak, contigs, read, asm, probK, probP)
    po.report_histogram(p, g, str(tmp_path / "o.hist"), str(tmp_path / "o.sum"))
    ix = build_index(m, k, read, asm)
    res = m.Evaluator(ix, m.KParams(peak, probK, probP)).hist(m.Sequences(contigs))
    res.report(k, str(tmp_path / "g.hist"), str(tmp_path / "g.sum"))
    assert (tmp_path / "g.hist").read_bytes() == (tmp_path / "o.hist").read_bytes()
    # summary: identical text (koverCpy printed %.2f)
    assert (tmp_path / "g.sum").read_text() == (tmp_path / "o.sum").read_text()


@pytest.mark.parametrize("k", [6, 8])
def test_even_k_palindromes_two_strand_path(k):
    """Even k: palindromic k-mers are their own reverse complement and the
    reference adds value(f)+value(r) twice (merfin-globals.C:107-108)."""
    m = _mfx()
    contigs, read, asm = synth.world(k=k, peak=3.0, sizes=(3000, 700, 4100), err_kmers=0, tandem=None)
    p, g, ka, km = oracle_hist(k, 3.0, contigs, read, asm)
    ix = build_index(m, k, read, asm)
    res = m.Evaluator(ix, m.KParams(3.0)).hist(m.Sequences(contigs))
    assert_hist_equal(res, g, ka, km, k)


def test_non_canonical_db_two_strand_path():
    """A DB holding forward (non-canonical) k-mers: both strands are probed and summed."""
    m = _mfx()
    k = 11
    r = synth.rng(3)
    contigs = synth.as_bytes(synth.decorate(r, synth.make_truth(r, (6000, 2500))))
    # forward-strand k-mer counts (not canonicalised)
    fw = {}
    for c in contigs:
        for _, f, _r in po.kiter(k, c):
            fw[f] = fw.get(f, 0) + 1
    ak = np.array(sorted(fw), dtype=np.uint64)
    av = np.array([fw[x] for x in ak.tolist()], dtype=np.uint32)
    rv = (av * 5 + (ak % 3).astype(np.uint32)).astype(np.uint32)
    p, g, ka, km = oracle_hist(k, 5.0, contigs, (ak, rv), (ak, av))
    ix = build_index(m, k, (ak, rv), (ak, av))
    assert not ix.info()["canonical"]
    res = m.Evaluator(ix, m.KParams(5.0)).hist(m.Sequences(contigs))
    assert_hist_equal(res, g, ka, km, k)


def test_forced_two_strand_equals_canonical(monkeypatch):
    m = _mfx()
    k, peak = 21, 17.3
    contigs, read, asm = synth.world(k=k, peak=peak, seed=11)
    ix = build_index(m, k, read, asm)
    ev = m.Evaluator(ix, m.KParams(peak))
    seqs = m.Sequences(contigs)
    a = ev.hist(seqs)
    monkeypatch.setenv("MFX_FORCE_TWO_STRAND", "1")
    b = ev.hist(seqs)
    assert a.kasm == b.kasm and a.kmissing == b.kmissing
    np.testing.assert_array_equal(a.undr(), b.undr())
    np.testing.assert_array_equal(a.over(), b.over())


def test_min_max_filter():
    m = _mfx()
    k, peak = 21, 17.3
    contigs, read, asm = synth.world(k=k, peak=peak, seed=5)
    p, g, ka, km = oracle_hist(k, peak, contigs, read, asm, minV=4, maxV=40)
    ix = build_index(m, k, read, asm, minV=4, maxV=40)
    res = m.Evaluator(ix, m.KParams(peak)).hist(m.Sequences(contigs))
    assert_hist_equal(res, g, ka, km, k)
    # value() applies the same load-time filter semantics
    R = po.Lookup(k, read[0], read[1], 4, 40)
    q = read[0][:5000]
    rv, _ = ix.value(q)
    np.testing.assert_array_equal(rv, np.array([R.value(x) for x in q.tolist()], dtype=np.uint32))


def test_index_value_and_export():
    m = _mfx()
    k = 21
    contigs, read, asm = synth.world(k=k, seed=9)
    ix = build_index(m, k, read, asm)
    R, A = po.Lookup(k, *read), po.Lookup(k, *asm)
    r = synth.rng(1)
    q = np.concatenate([read[0][::7], asm[0][::5], r.integers(0, 4 ** k, size=3000, dtype=np.uint64)])
    rv, av = ix.value(q)
    np.testing.assert_array_equal(rv, np.array([R.value(x) for x in q.tolist()], dtype=np.uint32))
    np.testing.assert_array_equal(av, np.array([A.value(x) for x in q.tolist()], dtype=np.uint32))
    ek, er, ea = ix.export()
    union = np.union1d(read[0], asm[0])
    np.testing.assert_array_equal(ek, union)
    assert ix.info()["distinct"] == len(union)
    np.testing.assert_array_equal(er, np.array([R.value(x) for x in ek.tolist()], dtype=np.uint32))
    np.testing.assert_array_equal(ea, np.array([A.value(x) for x in ek.tolist()], dtype=np.uint32))


def test_count_asm_matches_meryl_count_restatement():
    """mfx_index_count_asm replaces `meryl count` of -sequence (merfin-globals.C:182-186)."""
    m = _mfx()
    k = 21
    contigs, read, asm = synth.world(k=k, seed=13)
    ix = m.Index(k, len(read[0]) + len(asm[0]) + 16)
    ix.add_read(*read)
    seqs = m.Sequences(contigs)
    ix.count_asm(seqs)
    _, av = ix.value(asm[0])
    np.testing.assert_array_equal(av, asm[1])
    ek, er, ea = ix.export()
    assert int(ea.sum()) == int(asm[1].sum())
    assert ix.info()["canonical"]


def test_dump_values_and_text(tmp_path, golden_dir):
    m = _mfx()
    k, peak = 21, 26.0
    contigs, read, asm = synth.world(k=k, peak=peak, seed=21, sizes=(20000, 4096, 4117, 300, 10))
    probK, probP = po.load_kmetric(os.path.join(golden_dir, "example_lookup_table.txt"))
    p = po.Params(k, peak, probK, probP)
    R, A = po.Lookup(k, *read), po.Lookup(k, *asm)
    ix = build_index(m, k, read, asm)
    kp = m.KParams(peak, probK, probP)
    ev = m.Evaluator(ix, kp)
    seqs = m.Sequences(contigs)
    opath, gpath = str(tmp_path / "o.dump"), str(tmp_path / "g.dump")
    for ci, c in enumerate(contigs):
        rk, ak_, km_, kasm, kmiss = po.process_dump(p, R, A, c)
        po.output_dump(opath, "ctg%d" % ci, rk, ak_, km_, append=ci > 0)
        rv, av, gkasm, gkmiss = ev.dump_values(seqs, ci, 0, len(c))
        assert (gkasm, gkmiss) == (kasm, kmiss)
        # readK/asmK/K* recomputed on the host from the raw values are bit-identical
        for i in range(0, len(c), 37):
            a, b, _ = m.getK(kp, int(rv[i]), int(av[i]))
            assert (a, b, m.getKmetric(a, b)) == (rk[i], ak_[i], km_[i])
        # a sub-range that does not start on a tile boundary
        if len(c) > 5000:
            rv2, av2, _, _ = ev.dump_values(seqs, ci, 4099, 4999)
            np.testing.assert_array_equal(rv2, rv[4099:4999])
            np.testing.assert_array_equal(av2, av[4099:4999])
        ka2, km2 = ev.dump_contig(seqs, ci, "ctg%d" % ci, gpath, append=ci > 0)
        assert (ka2, km2) == (kasm, kmiss)
    assert open(gpath, "rb").read() == open(opath, "rb").read()
    assert os.path.getsize(gpath) > 100000


def test_completeness_matches_merge_loop():
    m = _mfx()
    k, peak = 21, 17.3
    contigs, read, asm = synth.world(k=k, peak=peak, seed=17)
    p = po.Params(k, peak)
    # the reference merges 64 pieces (top 6 bits of the k-mer); sums are of integers, so exact
    tot = und = 0.0
    for piece in range(64):
        lo, hi = piece << (2 * k - 6), (piece + 1) << (2 * k - 6)
        rs = (read[0] >= lo) & (read[0] < hi)
        as_ = (asm[0] >= lo) & (asm[0] < hi)
        t, u = po.completeness_piece(p, read[0][rs], read[1][rs], asm[0][as_], asm[1][as_])
        tot += t
        und += u
    ix = build_index(m, k, read, asm)
    ev = m.Evaluator(ix, m.KParams(peak))
    t, u = ev.completeness()
    assert (t, u) == (tot, und) and tot > 0 and und > 0
    t64, u64 = ev.completeness_pieces()
    for piece in range(64):
        lo, hi = piece << (2 * k - 6), (piece + 1) << (2 * k - 6)
        rs, as_ = (read[0] >= lo) & (read[0] < hi), (asm[0] >= lo) & (asm[0] < hi)
        assert (t64[piece], u64[piece]) == po.completeness_piece(p, read[0][rs], read[1][rs], asm[0][as_], asm[1][as_])


def test_large_bins_overflow_path():
    """asmK/readK ratios beyond the dense device bins go through the overflow list."""
    m = _mfx()
    k = 9
    seq = (b"ACGTTGCAAGGCTTAACCGGTTAAGCGCTAGCTAGGATCCGATCGATTACGCGCGATATATCGCGGCTA" * 3)
    ak, av = po.count_kmers(k, [seq])
    av = av.copy()
    av[0] = 3000000            # readK=1 -> idx ~ 1.5e7  (beyond nbins)
    av[1] = 7001               # idx 35000: beyond the LDS bins, inside the dense device bins
    av[2] = 1500               # idx 7495
    rv = np.ones(len(ak), dtype=np.uint32) * 5
    p, g, ka, km = oracle_hist(k, 5.0, [seq], (ak, rv), (ak, av))
    ix = build_index(m, k, (ak, rv), (ak, av))
    res = m.Evaluator(ix, m.KParams(5.0), nbins=40000).hist(m.Sequences([seq]))
    assert g.c.undrMax > 40000
    assert_hist_equal(res, g, ka, km, k)


def test_millions_of_kmers_in_far_bins_like_the_reference_unbounded_arrays():
    """A satellite array of the assembly that the reads under-represent: 2.6 M positions whose asmK / readK exceeds 13 107, i.e.
    more than 2^20 k-mer occurrences beyond the dense device bins.  The reference's arrays simply grow (increaseArray,
    merfin-histogram.C:74,87); here the far bins are aggregated on the device ({bin -> occurrences}: include/merfin_amd.h,
    mfx_hist_take_overflow), so the result is the oracle's whatever the number of occurrences -- on the whole-assembly run, the
    streamed one, several slots, the sequence-only index and a caller-side launch + take_overflow."""
    torch = pytest.importorskip("torch")
    m = _mfx()
    k, peak = 21, 5.0
    r = synth.rng(2026)
    flank = lambda n: synth.random_contig(r, n).tobytes()
    arr1 = b"GGAAT" * 300000                                   # 1.5 Mb: five distinct 21-mers, 300 000 copies each
    arr2 = (b"ACGTTGCATTGACCGTAAGCTTGGCATCGAT" + b"T") * 36000      # 32-mer unit x 36 000: 32 distinct k-mers, ~1.15 M positions
    contigs = [flank(70000) + arr1 + flank(50000), flank(4096 * 3 + 7), flank(30000) + arr2 + flank(9000)]
    ak, av = po.count_kmers(k, contigs)
    rv = np.full(len(ak), 5, dtype=np.uint32)                  # readK = 1 everywhere: the arrays' asmK / readK is their copy number
    rv[av > 100000] = 11                                       # ... and readK = 2 for arr1: bins ~ 750 000; arr2: ~ 180 000
    p, g, ka, km = oracle_hist(k, peak, contigs, (ak, rv), (ak, av))
    far = int(g.undr()[65536:].sum() + g.over()[65536:].sum())
    assert far > (1 << 20) + 1000000, far                      # well past the old list's 2^20 records
    ix = build_index(m, k, (ak, rv), (ak, av))
    seqs = m.Sequences(contigs)
    ev = m.Evaluator(ix, m.KParams(peak))
    assert_hist_equal(ev.hist(seqs), g, ka, km, k, tight=False)
    assert_hist_equal(ev.hist(seqs), g, ka, km, k, tight=False)             # the table of far bins is emptied between runs
    assert_hist_equal(ev.hist_streamed(m.Sequences.create([len(c) for c in contigs]), contigs), g, ka, km, k, tight=False)
    evs = [m.Evaluator(ix, m.KParams(peak)) for _ in range(3)]
    assert_hist_equal(m.hist_multi(evs, [seqs] * 3), g, ka, km, k, tight=False)
    # launch + take_overflow by hand: {far bin, occurrences} pairs, few of them, their occurrences the image's novf word
    counts = torch.zeros(m.hist_words(ev.nbins, seqs.ncontigs), dtype=torch.int64, device="cuda")
    kover = torch.zeros(1, dtype=torch.float64, device="cuda")
    ev.take_overflow()
    ev.hist_launch(seqs, 0, seqs.ntiles, counts, kover, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    h = counts.cpu().numpy().view(np.uint64)
    rec = ev.take_overflow()
    assert int(h[2 * ev.nbins + 2]) == far and int(rec[:, 1].sum()) == far and 2 <= len(rec) <= 64, (len(rec), far)
    assert_hist_equal(ev.result_from_counts(h, float(kover.item()), seqs.ncontigs).add_overflow(rec), g, ka, km, k, tight=False)
    assert len(ev.take_overflow()) == 0
    # the sequence-only (compact) index: the arrays' assembly counts are far beyond its 11-bit fields too
    six = m.Index.for_seq(k, sum(len(c) for c in contigs) + 1024)
    six.count_asm(seqs)
    six.add_read(ak, rv)
    assert_hist_equal(m.Evaluator(six, m.KParams(peak)).hist(seqs), g, ka, km, k, tight=False)


def test_tile_and_contig_edge_cases():
    """Empty / shorter-than-k / tile-sized contigs, N at tile edges, lower case."""
    m = _mfx()
    k = 21
    r = synth.rng(99)
    base = synth.random_contig(r, 3 * 4096 + 50)
    base[4095] = ord("N"); base[4096 + 20] = ord("n"); base[8191 - 10: 8191 + 10] |= 0x20
    contigs = [b"", b"ACGT", base.tobytes(), synth.random_contig(r, 4096).tobytes(), b"N" * 5000,
               synth.random_contig(r, 20).tobytes(), synth.random_contig(r, 21).tobytes(),
               synth.random_contig(r, 4096 + 20).tobytes(), b""]
    ak, av = po.count_kmers(k, contigs)
    rv = (3 + (ak % 40)).astype(np.uint32)
    p, g, ka, km = oracle_hist(k, 9.5, contigs, (ak, rv), (ak, av))
    ix = build_index(m, k, (ak, rv), (ak, av))
    ev = m.Evaluator(ix, m.KParams(9.5))
    seqs = m.Sequences(contigs)
    assert seqs.ncontigs == len(contigs) and seqs.nbases == sum(len(c) for c in contigs)
    res = ev.hist(seqs)
    assert_hist_equal(res, g, ka, km, k)
    assert list(ka[[0, 1, 4, 5, 8]]) == [0, 0, 0, 0, 0] and ka[6] == 1


def test_many_small_contigs():
    """A fragmented assembly: thousands of contigs shorter than a tile (and than the staging window logic's pieces)."""
    m = _mfx()
    k = 21
    r = synth.rng(123)
    contigs = [synth.random_contig(r, int(n)).tobytes() for n in r.integers(0, 900, size=3000)]
    contigs[17] = b""
    ak, av = po.count_kmers(k, contigs)
    rv = (2 + (ak % 30)).astype(np.uint32)
    p, g, ka, km = oracle_hist(k, 7.5, contigs, (ak, rv), (ak, av))
    ix = build_index(m, k, (ak, rv), (ak, av))
    seqs = m.Sequences(contigs)
    assert seqs.ncontigs == 3000
    res = m.Evaluator(ix, m.KParams(7.5)).hist(seqs)
    assert_hist_equal(res, g, ka, km, k)


def test_result_independent_of_grid_and_repeatable(monkeypatch):
    """Tiles are handed out dynamically, so which block evaluates which tile varies from launch to launch;
    every integer and koverCpy (kept per (tile, wave), summed in fixed order) must not: bit-identical for
    any persistent grid size and across repeated launches, on an assembly with very uneven tiles."""
    m = _mfx()
    k, peak = 21, 17.3
    contigs, read, asm = synth.world(k=k, peak=peak, seed=41)
    r = synth.rng(5)
    contigs = list(contigs) + [synth.random_contig(r, int(n)).tobytes() for n in r.integers(1, 9000, size=300)]
    contigs.insert(3, b"N" * 50000)
    ak, av = po.count_kmers(k, contigs)
    ix = build_index(m, k, read, (ak, av))
    seqs = m.Sequences(contigs)
    outs = []
    for bpc in ("1", "3", None, None):
        if bpc is None:
            monkeypatch.delenv("MFX_BLOCKS_PER_CU", raising=False)
        else:
            monkeypatch.setenv("MFX_BLOCKS_PER_CU", bpc)
        res = m.Evaluator(ix, m.KParams(peak)).hist(seqs)
        outs.append((res.kasm, res.kmissing, np.float64(res.koverCpy).tobytes(), res.undr().tobytes(), res.over().tobytes(),
                     res.contig_kasm().tobytes(), res.contig_kmissing().tobytes()))
    assert all(o == outs[0] for o in outs[1:])
    assert outs[0][0] > 0 and np.frombuffer(outs[0][2], dtype=np.float64)[0] > 0


def test_sharded_launch_equals_whole(golden_dir):
    """Position-tile sharding (the multi-GPU decomposition): any split of the tile
    range accumulates to the same integers; koverCpy to the same value within 1e-12."""
    torch = pytest.importorskip("torch")
    m = _mfx()
    k, peak = 21, 17.3
    contigs, read, asm = synth.world(k=k, peak=peak, seed=23)
    ix = build_index(m, k, read, asm)
    ev = m.Evaluator(ix, m.KParams(peak))
    seqs = m.Sequences(contigs)
    whole = ev.hist(seqs)
    T = seqs.ntiles
    words = m.hist_words(ev.nbins, seqs.ncontigs)
    for nshard in (2, 3, 8):
        counts = torch.zeros(words, dtype=torch.int64, device="cuda")
        kover = torch.zeros(1, dtype=torch.float64, device="cuda")
        s = torch.cuda.current_stream().cuda_stream
        for rnk in range(nshard):
            ev.hist_launch(seqs, T * rnk // nshard, T * (rnk + 1) // nshard, counts, kover, stream=s)
        torch.cuda.synchronize()
        res = ev.result_from_counts(counts.cpu().numpy().view(np.uint64), float(kover.item()), seqs.ncontigs)
        assert res.kasm == whole.kasm and res.kmissing == whole.kmissing
        np.testing.assert_array_equal(res.undr(), whole.undr())
        np.testing.assert_array_equal(res.over(), whole.over())
        np.testing.assert_array_equal(res.contig_kasm(), whole.contig_kasm())
        assert res.koverCpy == pytest.approx(whole.koverCpy, rel=1e-12)


@pytest.mark.parametrize("nranks,blk", [(2, 1), (3, 4), (8, 2), (5, 256)])
def test_cyclic_launches_equal_whole(nranks, blk):
    """block-cyclic tile partition (multi-GPU -hist): the ranks' shares cover every tile once, so they
    accumulate to the whole-assembly integers; and each share equals the contiguous launches of its runs"""
    torch = pytest.importorskip("torch")
    m = _mfx()
    from merfin_amd import distributed as D
    k, peak = 21, 17.3
    contigs, read, asm = synth.world(k=k, peak=peak, seed=29)
    ix = build_index(m, k, read, asm)
    ev = m.Evaluator(ix, m.KParams(peak))
    seqs = m.Sequences(contigs)
    whole = ev.hist(seqs)
    T = seqs.ntiles
    assert T > 2 * blk or blk == 256
    words = m.hist_words(ev.nbins, seqs.ncontigs)
    counts = torch.zeros(words, dtype=torch.int64, device="cuda")
    kover = torch.zeros(1, dtype=torch.float64, device="cuda")
    for rnk in range(nranks):
        c1 = torch.zeros(words, dtype=torch.int64, device="cuda")
        k1 = torch.zeros(1, dtype=torch.float64, device="cuda")
        ev.hist_launch_cyclic(seqs, rnk, nranks, c1, k1, block_tiles=blk)
        c2 = torch.zeros(words, dtype=torch.int64, device="cuda")
        k2 = torch.zeros(1, dtype=torch.float64, device="cuda")
        for lo, hi in D.cyclic_tiles(T, rnk, nranks, blk):
            ev.hist_launch(seqs, lo, hi, c2, k2)
        torch.cuda.synchronize()
        assert torch.equal(c1, c2) and float(k1.item()) == pytest.approx(float(k2.item()), rel=1e-12, abs=1e-12)
        counts += c1
        kover += k1
    res = ev.result_from_counts(counts.cpu().numpy().view(np.uint64), float(kover.item()), seqs.ncontigs)
    assert res.kasm == whole.kasm and res.kmissing == whole.kmissing
    np.testing.assert_array_equal(res.undr(), whole.undr())
    np.testing.assert_array_equal(res.over(), whole.over())
    np.testing.assert_array_equal(res.contig_kasm(), whole.contig_kasm())
    np.testing.assert_array_equal(res.contig_kmissing(), whole.contig_kmissing())
    assert res.koverCpy == pytest.approx(whole.koverCpy, rel=1e-12)
    with pytest.raises(m.MfxError):
        ev.hist_launch_cyclic(seqs, 0, 2, counts, kover, block_tiles=3)


def test_revcomp_symmetry():
    """Property: the reverse-complemented assembly has the same histogram (both strands are summed)."""
    m = _mfx()
    k, peak = 21, 17.3
    contigs, read, asm = synth.world(k=k, peak=peak, seed=29, sizes=(30000, 5000))
    comp = bytes.maketrans(b"ACGTacgt", b"TGCAtgca")
    rc = [c.translate(comp)[::-1] for c in contigs]
    ix = build_index(m, k, read, asm)
    ev = m.Evaluator(ix, m.KParams(peak))
    a, b = ev.hist(m.Sequences(contigs)), ev.hist(m.Sequences(rc))
    assert (a.kasm, a.kmissing) == (b.kasm, b.kmissing)
    np.testing.assert_array_equal(a.undr(), b.undr())
    np.testing.assert_array_equal(a.over(), b.over())
    assert a.koverCpy == pytest.approx(b.koverCpy, rel=1e-12)


@pytest.mark.parametrize("lf", ["0.9", "0.3"])
def test_heavy_line_overflow_and_sparse_tables(lf, monkeypatch):
    """Load factor 0.9: ~25 % of the k-mers overflow their 128-byte home line, so the
    continuation path of the cooperative probe carries real weight; 0.3: sparse lines."""
    m = _mfx()
    monkeypatch.setenv("MFX_LOAD_FACTOR", lf)
    k, peak = 21, 17.3
    contigs, read, asm = synth.world(k=k, peak=peak, seed=37, err_kmers=20000)
    p, g, ka, km = oracle_hist(k, peak, contigs, read, asm)
    ix = build_index(m, k, read, asm, cap=len(np.union1d(read[0], asm[0])))
    info = ix.info()
    assert abs(info["distinct"] / info["capacity"] - float(lf)) < 0.05 or info["capacity"] == 8192
    ev = m.Evaluator(ix, m.KParams(peak))
    seqs = m.Sequences(contigs)
    assert_hist_equal(ev.hist(seqs), g, ka, km, k)
    # the per-lane probe (dump / value kernels) agrees with the cooperative one
    R, A = po.Lookup(k, *read), po.Lookup(k, *asm)
    q = np.concatenate([read[0][::3], asm[0][::3]])
    rv, av = ix.value(q)
    np.testing.assert_array_equal(rv, np.array([R.value(x) for x in q.tolist()], dtype=np.uint32))
    np.testing.assert_array_equal(av, np.array([A.value(x) for x in q.tolist()], dtype=np.uint32))


def test_index_full_is_reported():
    m = _mfx()
    k = 21
    r = synth.rng(4)
    kmers = np.unique(r.integers(0, 4 ** k, size=200000, dtype=np.uint64))
    ix = m.Index(k, 1000)        # far too small (1024 lines = 8192 slots minimum)
    with pytest.raises(m.MfxError) as e:
        ix.add_asm(kmers, np.ones(len(kmers), dtype=np.uint32))
    assert e.value.code == -4


@pytest.mark.parametrize("insert_mode,count_mode", [("0", "0"), ("1", "1"), ("0", "1"), ("1", "0")])
def test_both_insert_strategies_build_the_same_table(insert_mode, count_mode, monkeypatch):
    """The index build has two insert strategies per kernel (cooperative batched passes / per-lane walk; defaults: table
    adds cooperative, assembly counter per-lane).  Either must produce the same table: duplicates summed, one slot per
    k-mer, results equal to the oracle -- including many inserts of the SAME k-mer racing each other."""
    m = _mfx()
    monkeypatch.setenv("MFX_INSERT_MODE", insert_mode)
    monkeypatch.setenv("MFX_COUNT_MODE", count_mode)
    k, peak = 21, 17.3
    contigs, read, asm = synth.world(k=k, peak=peak, seed=61, sizes=(60000, 9000, 4097, 20, 0), tandem=(5, 400))
    p, g, ka, km = oracle_hist(k, peak, contigs, read, asm)
    seqs = m.Sequences(contigs)
    ix = m.Index(k, len(read[0]) + len(asm[0]) + 16)
    # every read k-mer inserted as `value` separate inserts of 1, shuffled: heavy same-key contention
    r = synth.rng(5)
    occ = np.repeat(read[0], np.minimum(read[1], 40).astype(np.int64))
    rest = (read[1] - np.minimum(read[1], 40)).astype(np.uint32)
    r.shuffle(occ)
    ix.add_read(occ, np.ones(len(occ), dtype=np.uint32))
    ix.add_read(read[0][rest > 0], rest[rest > 0])
    ix.count_asm(seqs)                                        # homopolymer / tandem stretches: the same k-mer from neighbouring lanes
    ek, er, ea = ix.export()
    union = np.union1d(read[0], asm[0])
    np.testing.assert_array_equal(ek, union)                  # one slot per k-mer
    np.testing.assert_array_equal(er[np.isin(ek, read[0])], read[1])
    np.testing.assert_array_equal(ea[np.isin(ek, asm[0])], asm[1])
    assert_hist_equal(m.Evaluator(ix, m.KParams(peak)).hist(seqs), g, ka, km, k)


@pytest.mark.parametrize("seed", list(range(int(os.environ.get("MFX_RANDOM_SEEDS", "24")))))      # MFX_RANDOM_SEEDS=500 for a soak
def test_randomized_worlds_match_oracle(seed, monkeypatch):
    """A seeded sweep over what the fixed cases hold constant: k (3...31, odd and even), peak (also < 1 and huge), random
    -prob tables (readK 0 for some counts, rows beyond and below the LDS look-up tables), read counts up to 200000, -min/-max,
    table fill, minimizer windows, contig shapes (empty, shorter than k, N runs, lower case), -hist and -dump."""
    m = _mfx()
    r = np.random.default_rng(1000 + seed)
    k = int(r.integers(3, 32))
    peak = float(r.choice([0.37, 1.0, 2.5, 9.0, 17.3, 26.0, 333.3, 1e6]))
    monkeypatch.setenv("MFX_LOAD_FACTOR", str(r.choice([0.3, 0.5, 0.7, 0.9])))
    monkeypatch.setenv("MFX_MZ_W", str(int(r.integers(1, 6))))
    if r.random() < 0.2:
        monkeypatch.setenv("MFX_HOME_MODE", "plain")
    sizes = tuple(int(x) for x in r.choice([0, 1, k - 1, k, k + 1, 37, 500, 4095, 4096, 4097, 9000, 20000], size=int(r.integers(1, 9))))
    contigs, read, asm = synth.world(k=k, peak=max(peak, 1.0) if peak < 1e5 else 20.0, seed=2000 + seed, sizes=sizes, err_kmers=int(r.integers(0, 3000)) if k > 8 else 0)
    rk, rv = read
    rv = rv.astype(np.uint64)
    if len(rv):
        big = r.random(len(rv)) < 0.02                            # a few enormous and a few tiny counts
        rv[big] = r.choice([1, 2, 1023, 1024, 1025, 65535, 200000], size=int(big.sum()))   # bins up to ~2.7 M (beyond the dense device image)
    rv = rv.astype(np.uint32)
    read = (rk, rv)
    probK = probP = None
    if r.random() < 0.6:
        n = int(r.choice([1, 8, 184, 1500]))
        probK = r.integers(0, 12, size=n).astype(np.uint32)
        probK[r.random(n) < 0.3] = 0                              # present in the reads but treated as missing
        probP = np.round(r.random(n), 4)
    lo, hi = 0, 2**64 - 1
    if r.random() < 0.4:
        lo, hi = int(r.integers(0, 5)), int(r.choice([30, 1000, 2**31]))
    p, g, ka, km = oracle_hist(k, peak, contigs, read, asm, probK, probP, lo, hi)
    ix = build_index(m, k, read, asm, lo, hi)
    ev = m.Evaluator(ix, m.KParams(peak, probK, probP))
    seqs = m.Sequences(contigs)
    assert_hist_equal(ev.hist(seqs), g, ka, km, k)
    R, A = po.Lookup(k, read[0], read[1], lo, hi), po.Lookup(k, *asm)
    kp = m.KParams(peak, probK, probP)
    # the sequence-only index of the same world (what the CLI builds for -hist / -dump): k-mers claimed from the sequence,
    # assembly counts from the database, read counts update-only -- same answers, raw values included
    six = m.Index.for_seq(k, sum(len(c) for c in contigs) + 16)
    six.claim_seq(seqs)
    six.add_asm(*asm)
    six.add_read(read[0], read[1], lo, hi)
    sev = m.Evaluator(six, m.KParams(peak, probK, probP))
    assert_hist_equal(sev.hist(seqs), g, ka, km, k)
    for c, ctg in enumerate(contigs[:3]):
        n = len(ctg)
        if n == 0:
            continue
        gv, av, dka, dkm = ev.dump_values(seqs, c, 0, n)
        sgv, sav, sdka, sdkm = sev.dump_values(seqs, c, 0, n)
        np.testing.assert_array_equal(sgv, gv)
        np.testing.assert_array_equal(sav, av)
        assert (sdka, sdkm) == (dka, dkm)
        rkk, akk, kmm, oka, okm = po.process_dump(p, R, A, ctg)
        assert (dka, dkm) == (oka, okm)
        for i in range(0, max(n - k + 1, 0), max(1, n // 300)):
            a, b, _ = m.getK(kp, int(gv[i]), int(av[i]))
            assert (a, b) == (rkk[i], akk[i]) and m.getKmetric(a, b) == kmm[i]


def test_bin_index_beyond_the_references_array_bound_is_an_error():
    """merfin-histogram.C:74,87 grow the bin arrays under a uint32 bound in steps of 1024: a bin index above 2^32 - 1025
    overflows that bound in the reference (undefined behaviour).  Here it is a clean error -- not a 34 GB allocation with
    a wrapped bound, which is what a read count of 2^32 - 1 at -peak 1 used to produce."""
    m = _mfx()
    k = 11
    contigs = [b"ACGTTGCATGCAAGCTTGCATGCAAGGTACCATGG"]
    ak, av = po.count_kmers(k, contigs)
    rv = np.full(len(ak), 3, dtype=np.uint32)
    one = int(np.argmax(av == 1))                             # a k-mer that occurs once: bin of 4294967295 / 1
    rv[one] = 2**32 - 1
    ix = build_index(m, k, (ak, rv), (ak, av))
    ev = m.Evaluator(ix, m.KParams(1.0))
    with pytest.raises(m.MfxError, match="32-bit array bound"):
        ev.hist(m.Sequences(contigs))
    rv[one] = 400000                                          # a large but representable bin: 2 M bins, fine
    ix2 = build_index(m, k, (ak, rv), (ak, av))
    res = m.Evaluator(ix2, m.KParams(1.0)).hist(m.Sequences(contigs))
    p, g, ka, km = oracle_hist(k, 1.0, contigs, (ak, rv), (ak, av))
    assert_hist_equal(res, g, ka, km, k)
    assert len(_trim(res.over())) > 1000000
