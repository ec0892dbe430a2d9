"""The SEQUENCE-ONLY index (mfx_index_create_for_seq): the lookup object of -hist and -dump, which only ever ask for the
k-mers of -sequence (merfin-histogram.C:54-64, merfin-dump.C:44-61).  It holds the k-mers claimed from the sequence;
loads only update.  For k <= 31 it takes the compact layout (8-byte slots, 16 per line, saturated counts in a side
table; 22 <= k <= 31: the slot's key field holds the k-mer's QUOTIENT -- what its line does not already say).  Everything it answers must equal what the full index -- and the oracle -- answer."""
import os

import numpy as np
import pytest

from oracle import pyoracle as po
from tests import synth
from tests.test_gpu_parity import assert_hist_equal, oracle_hist, build_index

pytestmark = pytest.mark.gpu


def _mfx():
    import merfin_amd as m
    if m.device_count() < 1:
        pytest.fail("no HIP device visible: the GPU tests must run on the MI355X box")
    return m


def seq_index(m, k, contigs, read, asm=None, lo=0, hi=2**64 - 1, seqs=None):
    """asm None: the assembly side is counted from the sequence (no -seqmers); else claimed, then loaded from `asm`"""
    seqs = seqs or m.Sequences(contigs)
    ix = m.Index.for_seq(k, sum(len(c) for c in contigs) + 16)
    if asm is None:
        ix.count_asm(seqs)
    else:
        ix.claim_seq(seqs)
        ix.add_asm(*asm)
    ix.add_read(read[0], read[1], lo, hi)
    return ix, seqs


@pytest.mark.parametrize("k,peak,use_prob,compact", [(21, 17.3, False, "1"), (21, 26.0, True, "1"), (21, 26.0, True, "0"), (31, 17.3, False, "1"),
                                                     (15, 9.0, False, "1"), (8, 9.0, False, "1"), (12, 3.0, True, "0"),
                                                     (31, 17.3, False, "0"), (22, 9.0, False, "1"), (24, 26.0, True, "1"), (27, 17.3, False, "1"),
                                                     (30, 9.0, False, "1")])
def test_seq_only_hist_and_dump_match_oracle(k, peak, use_prob, compact, golden_dir, monkeypatch):
    m = _mfx()
    monkeypatch.setenv("MFX_SEQ_COMPACT", compact)
    contigs, read, asm = synth.world(k=k, peak=peak, seed=300 + k)
    probK = probP = None
    if use_prob:
        probK, probP = po.load_kmetric(os.path.join(golden_dir, "example_lookup_table.txt"))
    p, g, ka, km = oracle_hist(k, peak, contigs, read, asm, probK, probP)
    for from_db in (False, True):
        ix, seqs = seq_index(m, k, contigs, read, asm if from_db else None)
        info = ix.info()
        assert info["seq_only"] and info["compact"] == (compact == "1" and k <= 31)
        assert info["distinct"] == len(asm[0])                    # one slot per distinct canonical k-mer of the sequence
        # every read k-mer that is not a k-mer of the sequence was dropped, nothing else
        assert info["dropped"] == int(np.count_nonzero(~np.isin(read[0], asm[0]) & (read[1] > 0)))
        ev = m.Evaluator(ix, m.KParams(peak, probK, probP))
        assert_hist_equal(ev.hist(seqs), g, ka, km, k)
        assert_hist_equal(ev.hist_streamed(m.Sequences.create([len(c) for c in contigs]), contigs), g, ka, km, k)
        # value(): the pair of every k-mer of the sequence, 0 for anything else
        rd = dict(zip(read[0].tolist(), read[1].tolist()))
        rv, av = ix.value(asm[0])
        np.testing.assert_array_equal(av, asm[1])
        np.testing.assert_array_equal(rv, np.array([rd.get(x, 0) for x in asm[0].tolist()], dtype=np.uint32))
        foreign = read[0][~np.isin(read[0], asm[0])][:500]
        rv, av = ix.value(foreign)
        assert not rv.any() and not av.any()
        ek, er, ea = ix.export()
        np.testing.assert_array_equal(ek, asm[0])
        np.testing.assert_array_equal(ea, asm[1])
        # -dump raw values and text
        full = m.Evaluator(build_index(m, k, read, asm), m.KParams(peak, probK, probP))
        for c in range(min(3, len(contigs))):
            if len(contigs[c]):
                a = ev.dump_values(seqs, c, 0, len(contigs[c]))
                b = full.dump_values(seqs, c, 0, len(contigs[c]))
                np.testing.assert_array_equal(a[0], b[0])
                np.testing.assert_array_equal(a[1], b[1])
                assert a[2:] == b[2:]


@pytest.mark.parametrize("seed", list(range(12)))
def test_seq_only_saturated_counts_and_crowded_lines(seed, monkeypatch):
    """counts at and beyond the 11-bit fields (2046, 2047, 2048, 200000; reached in one add and by several adds), assembly
    counts beyond them too, -min/-max, and tables so full that lines overflow (second cooperative pass, per-lane path)"""
    m = _mfx()
    monkeypatch.setenv("MFX_MZ_W", str(3 + seed % 3))
    monkeypatch.setenv("MFX_LOAD_FACTOR", str([0.25, 0.5, 0.85][seed % 3]))
    r = np.random.default_rng(5000 + seed)
    k = int(r.choice([9, 13, 17, 21]))
    peak = float(r.choice([2.5, 9.0, 26.0]))
    contigs, read, asm = synth.world(k=k, peak=peak, seed=5100 + seed, sizes=(20000, 6000, 4097, 30, 0), err_kmers=1500)
    rv = read[1].astype(np.uint64)
    big = r.random(len(rv)) < 0.03
    rv[big] = r.choice([2046, 2047, 2048, 5000, 200000], size=int(big.sum()))
    read = (read[0], rv.astype(np.uint32))
    av = asm[1].copy()
    av[r.random(len(av)) < 0.01] = 3000                        # saturated assembly counts too
    asm = (asm[0], av)
    lo, hi = (2, 2500) if seed % 2 else (0, 2**64 - 1)
    p, g, ka, km = oracle_hist(k, peak, contigs, read, asm, None, None, lo, hi)
    seqs = m.Sequences(contigs)
    ix = m.Index.for_seq(k, sum(len(c) for c in contigs) + 16)
    ix.claim_seq(seqs)
    ix.add_asm(*asm)
    # the read counts arrive in three instalments: fields saturate both in one step and by accumulation
    parts = [read[1] // 3, read[1] // 3, read[1] - 2 * (read[1] // 3)]
    for part in parts:
        ix.add_read(read[0], part, lo, hi)
    ev = m.Evaluator(ix, m.KParams(peak))
    assert_hist_equal(ev.hist(seqs), g, ka, km, k)
    rd = dict(zip(read[0].tolist(), read[1].tolist()))
    want = np.array([rd.get(x, 0) for x in asm[0].tolist()], dtype=np.uint32)
    want_f = np.where((want < lo) | (want > min(hi, 2**32 - 1)), 0, want)
    gr, ga = ix.value(asm[0])
    np.testing.assert_array_equal(ga, asm[1])
    np.testing.assert_array_equal(gr, want_f)
    ek, er, ea = ix.export()                                  # raw counts, unfiltered
    np.testing.assert_array_equal(er, want)
    # the same counts when every k-mer OCCURRENCE of the sequence is counted one by one (fields saturate by +1 steps)
    cx = m.Index.for_seq(k, sum(len(c) for c in contigs) + 16)
    cx.count_asm(seqs)
    true_asm = po.count_kmers(k, contigs)
    _, ga2 = cx.value(true_asm[0])
    np.testing.assert_array_equal(ga2, true_asm[1])


@pytest.mark.parametrize("k,lf,seed", [(22, "0.5", 1), (22, "0.85", 2), (23, "0.85", 3), (24, "0.7", 4), (24, "0.85", 5), (25, None, 6), (28, None, 7)])
def test_quotient_form_crowded_tables_and_the_side_table(k, lf, seed, monkeypatch):
    """22 <= k <= 31: the slot keeps the k-mer's quotient (mfx_q_place), a k-mer has THREE candidate lines (d rides in the key
    field) and lives in the side table under its full key beyond them.  Tables so full that all of that happens (the smallest
    table of a k grows with k -- k = 22..24 can be crowded by a test-sized world): claims, counts by unit steps, loads in
    instalments with saturating fields, -hist (wave path and the generic kernel), value(), the export (quotient -> k-mer) and
    -dump must all be the oracle's"""
    m = _mfx()
    if lf:
        monkeypatch.setenv("MFX_LOAD_FACTOR", lf)
    r = np.random.default_rng(9000 + seed)
    peak = float(r.choice([2.5, 9.0, 26.0]))
    contigs, read, asm = synth.world(k=k, peak=peak, seed=9100 + seed, sizes=(60000, 20000, 4097, 30, 0), err_kmers=1500)
    rv = read[1].astype(np.uint64)
    big = r.random(len(rv)) < 0.03
    rv[big] = r.choice([2046, 2047, 2048, 5000, 200000], size=int(big.sum()))
    read = (read[0], rv.astype(np.uint32))
    p, g, ka, km = oracle_hist(k, peak, contigs, read, asm)
    seqs = m.Sequences(contigs)
    ix = m.Index.for_seq(k, sum(len(c) for c in contigs) + 16)
    ix.count_asm(seqs)                                            # every occurrence counted one by one
    info = ix.info()
    assert info["compact"] and info["distinct"] == len(asm[0])
    parts = [read[1] // 2, read[1] - read[1] // 2]
    for part in parts:
        ix.add_read(read[0], part)
    ek, er, ea = ix.export()
    np.testing.assert_array_equal(ek, asm[0])
    np.testing.assert_array_equal(ea, asm[1])
    rd = dict(zip(read[0].tolist(), read[1].tolist()))
    want = np.array([rd.get(x, 0) for x in asm[0].tolist()], dtype=np.uint32)
    np.testing.assert_array_equal(er, want)
    gr, ga = ix.value(asm[0])
    np.testing.assert_array_equal(ga, asm[1])
    np.testing.assert_array_equal(gr, want)
    foreign = read[0][~np.isin(read[0], asm[0])][:2000]
    fr, fa = ix.value(foreign)
    assert not fr.any() and not fa.any()
    ev = m.Evaluator(ix, m.KParams(peak))
    assert_hist_equal(ev.hist(seqs), g, ka, km, k)
    assert_hist_equal(ev.hist_streamed(m.Sequences.create([len(c) for c in contigs]), contigs), g, ka, km, k)
    monkeypatch.setenv("MFX_HIST_GENERIC", "1")
    assert_hist_equal(m.Evaluator(ix, m.KParams(peak)).hist(seqs), g, ka, km, k)
    monkeypatch.delenv("MFX_HIST_GENERIC")
    full = m.Evaluator(build_index(m, k, read, asm), m.KParams(peak))
    for c in range(2):
        a = ev.dump_values(seqs, c, 0, len(contigs[c]))
        b = full.dump_values(seqs, c, 0, len(contigs[c]))
        np.testing.assert_array_equal(a[0], b[0])
        np.testing.assert_array_equal(a[1], b[1])
        assert a[2:] == b[2:]


def test_seq_only_counts_accumulate_beyond_the_field_by_unit_steps():
    """a 3000-copy tandem repeat: its k-mers' assembly counts cross 2047 one occurrence at a time, from many lanes at once"""
    m = _mfx()
    k = 21
    r = synth.rng(77)
    unit = synth.random_contig(r, 37).tobytes()
    contigs = [synth.random_contig(r, 5000).tobytes() + unit * 3000 + synth.random_contig(r, 5000).tobytes()]
    ak, av = po.count_kmers(k, contigs)
    assert av.max() >= 2990
    seqs = m.Sequences(contigs)
    ix = m.Index.for_seq(k, len(contigs[0]) + 16)
    ix.count_asm(seqs)
    read = (ak, (av.astype(np.uint64) * 17 % 5000).astype(np.uint32))
    ix.add_read(*read)
    gr, ga = ix.value(ak)
    np.testing.assert_array_equal(ga, av)
    np.testing.assert_array_equal(gr, read[1])
    p, g, ka, km = oracle_hist(k, 17.0, contigs, read, (ak, av))
    assert_hist_equal(m.Evaluator(ix, m.KParams(17.0)).hist(seqs), g, ka, km, k)


def test_seq_only_rules():
    m = _mfx()
    k = 21
    contigs, read, asm = synth.world(k=k, peak=9.0, seed=5)
    ix, seqs = seq_index(m, k, contigs, read)
    with pytest.raises(m.MfxError, match="claimed before"):      # no claims once counts arrived
        ix.count_asm(seqs)
    with pytest.raises(m.MfxError, match="claimed before"):
        ix.claim_seq(seqs)
    # one sequence-only index answers for ONE sequence object: a claim from a second, different one is refused where it is made
    # (it would leave the index bound to the last sequence and the evaluation of the first refused); the same content again is fine
    two = m.Index.for_seq(k, sum(len(c) for c in contigs) + 16)
    two.claim_seq(m.Sequences(contigs[:2]))
    two.claim_seq(m.Sequences(contigs[:2]))
    with pytest.raises(m.MfxError, match="already claimed the k-mers of another sequence"):
        two.claim_seq(m.Sequences(contigs[2:]))
    ev = m.Evaluator(ix, m.KParams(9.0))
    with pytest.raises(m.MfxError, match="sequence-only"):
        ev.completeness()
    with pytest.raises(m.MfxError, match="sequence-only"):
        ix.set_shard(0, 2)
    with pytest.raises(m.MfxError, match="sequence-only"):
        m.Router(ix, 2, 4)
    with pytest.raises(m.MfxError, match="sequence-only"):
        m.Index(k, 100).claim_seq(seqs)                          # a full index takes no claims
    with pytest.raises(m.MfxError):
        m.Index.for_seq(33, 100)                                 # k <= 31
    # a non-canonical database cannot be answered from one slot per canonical k-mer
    def rc(x):
        y = 0
        for _ in range(k):
            y = (y << 2) | ((x & 3) ^ 2)
            x >>= 2
        return y
    nc = np.array([rc(int(x)) for x in asm[0][:50].tolist()], dtype=np.uint64)
    nc = nc[nc > asm[0][:50]]
    assert len(nc)
    ix2 = m.Index.for_seq(k, sum(len(c) for c in contigs) + 16)
    ix2.count_asm(seqs)
    with pytest.raises(m.MfxError, match="non-canonical") as ei:
        ix2.add_read(nc, np.ones(len(nc), dtype=np.uint32))
    assert ei.value.code == m.E_NONCANON


def test_seq_only_index_refuses_another_sequence(tmp_path):
    """a sequence-only index answers for the k-mers of the sequence it was claimed from: k-mers it never claimed would read as
    absent (a full table would answer their read count), so evaluating ANOTHER or an edited sequence on it is an error, for
    -hist (resident, launch by launch, streamed), -dump, an image and a replica alike; the same content in another form
    (another object, lower case, other invalid bytes, packed planes or bytes) is the same sequence (advisor, round 3)."""
    m = _mfx()
    k, peak = 21, 9.0
    contigs, read, asm = synth.world(k=k, peak=peak, seed=11)
    ix, seqs = seq_index(m, k, contigs, read)
    ev = m.Evaluator(ix, m.KParams(peak))
    ref = ev.hist(seqs)
    # the same content, held differently
    low = [bytes(c).lower().replace(b"n", b"x") for c in contigs]
    for same in (m.Sequences(contigs), m.Sequences(low)):
        r = ev.hist(same)
        assert (r.kasm, r.kmissing) == (ref.kasm, ref.kmissing) and r.koverCpy == ref.koverCpy
    r = ev.hist_streamed(m.Sequences.create([len(c) for c in contigs]), low)
    assert (r.kasm, r.kmissing) == (ref.kasm, ref.kmissing)
    # one substituted base; one contig fewer; the contigs in another order
    edited = [bytearray(c) for c in contigs]
    big = max(range(len(edited)), key=lambda i: len(edited[i]))
    pos = next(i for i in range(100, len(edited[big])) if edited[big][i:i + 1] in (b"A", b"C", b"G", b"T"))
    edited[big][pos:pos + 1] = b"C" if edited[big][pos:pos + 1] != b"C" else b"G"
    edited = [bytes(c) for c in edited]
    others = [edited, [c for c in contigs if len(c)][:-1], list(reversed(contigs))]
    img = str(tmp_path / "img")
    ix.save(img)
    for holder in (ix, m.Index.load(img), ix.replicate(0)):
        e2 = m.Evaluator(holder, m.KParams(peak))
        for o in others:
            if [bytes(x) for x in o] == [bytes(x) for x in contigs]:
                continue
            so = m.Sequences(o)
            with pytest.raises(m.MfxError, match="ANOTHER sequence") as ei:
                e2.hist(so)
            assert ei.value.code == -1
            with pytest.raises(m.MfxError, match="ANOTHER sequence"):
                e2.hist_streamed(m.Sequences.create([len(c) for c in o]), o)
            with pytest.raises(m.MfxError, match="ANOTHER sequence"):
                e2.dump_values(so, 0, 0, len(o[0]))
        r = e2.hist(seqs)
        assert (r.kasm, r.kmissing) == (ref.kasm, ref.kmissing)
    # a full index has no such tie
    full = m.Evaluator(build_index(m, k, read, asm), m.KParams(peak))
    full.hist(m.Sequences(edited))


@pytest.mark.parametrize("k", [21, 27])
def test_seq_only_image_roundtrip_and_replica(k, tmp_path):
    """the device-format image (-index cache) and the replica made for another slot carry the compact layout, its side
    table and the sequence-only flag"""
    m = _mfx()
    peak = 17.3
    contigs, read, asm = synth.world(k=k, peak=peak, seed=9)
    read = (read[0], np.where(np.arange(len(read[1])) % 40 == 0, 60000, read[1]).astype(np.uint32))    # saturated fields in the image
    p, g, ka, km = oracle_hist(k, peak, contigs, read, asm)
    ix, seqs = seq_index(m, k, contigs, read)
    ix.save(str(tmp_path / "img"))
    for other in (m.Index.load(str(tmp_path / "img")), ix.replicate(0)):
        info = other.info()
        assert info["seq_only"] and info["compact"] == (k <= 31) and info["distinct"] == len(asm[0])
        assert_hist_equal(m.Evaluator(other, m.KParams(peak)).hist(seqs), g, ka, km, k)
        with pytest.raises(m.MfxError, match="claimed before"):
            other.claim_seq(seqs)


def test_seq_only_multi_slot_and_even_k():
    m = _mfx()
    for k in (20, 21):                                            # even k: both strands are probed and summed
        peak = 9.0
        contigs, read, asm = synth.world(k=k, peak=peak, seed=40 + k)
        p, g, ka, km = oracle_hist(k, peak, contigs, read, asm)
        ix, seqs = seq_index(m, k, contigs, read)
        evs = [m.Evaluator(ix, m.KParams(peak)) for _ in range(3)]
        assert_hist_equal(m.hist_multi(evs, [seqs] * 3), g, ka, km, k)


@pytest.mark.gpu
@pytest.mark.parametrize("k", [21, 31, 41])
def test_packed_upload_equals_the_byte_upload(k, monkeypatch):
    """mfx_seq_upload sends large sequences as their packed planes (2-bit codes + validity bits encoded by the host threads;
    one byte per base is made on the device on demand).  The same contigs -- N runs, lowercase, IUPAC codes, empty and
    one-base contigs -- through both transports: counted, evaluated and dumped alike, and as the oracle says"""
    m = _mfx()
    peak = 17.3
    contigs, read, asm = synth.world(k=min(k, 31), peak=peak, seed=911)
    results = []
    for pm in ("0", str(1 << 40)):
        monkeypatch.setenv("MFX_UPLOAD_PACKED_MIN", pm)
        seqs = m.Sequences(contigs)
        ix = m.Index(k, sum(len(c) for c in contigs) + 64)
        ix.count_asm(seqs)
        ev = m.Evaluator(ix, m.KParams(peak))
        h = ev.hist(seqs)
        d = [ev.dump_values(seqs, c, 0, len(contigs[c])) for c in range(len(contigs)) if len(contigs[c])]
        results.append((ix.export(), h, d))
    (ea, ha, da), (eb, hb, db) = results
    for x, y in zip(ea, eb):
        np.testing.assert_array_equal(x, y)
    assert ha.kasm == hb.kasm and ha.kmissing == hb.kmissing and ha.koverCpy == hb.koverCpy
    np.testing.assert_array_equal(ha.undr(), hb.undr())
    np.testing.assert_array_equal(ha.over(), hb.over())
    np.testing.assert_array_equal(ha.contig_kasm(), hb.contig_kasm())
    np.testing.assert_array_equal(ha.contig_kmissing(), hb.contig_kmissing())
    for x, y in zip(da, db):
        np.testing.assert_array_equal(x[0], y[0])
        np.testing.assert_array_equal(x[1], y[1])
        assert x[2:] == y[2:]
    if k <= 31:
        np.testing.assert_array_equal(ea[0], asm[0])
        np.testing.assert_array_equal(ea[2], asm[1])


@pytest.mark.parametrize("k", list(range(13, 32)))
def test_mod_minimizer_placement_every_k(k, monkeypatch):
    """the compact layout places a k-mer by the window its smallest t-mer samples (mod-minimizer: t = 4..7 by k, windows of
    8 / 12 / 16 t-mers).  The evaluation kernel finds those windows for a whole wave at once, the build kernels and the by-key
    lookups one k-mer at a time: both must agree for every k the layout takes -- sequences with N runs, lowercase, contigs
    that end inside and at the edges of tiles, low-complexity stretches (every t-mer of a window equal: all ties) -- and the
    results are the oracle's, with the specialised and the generic kernel, and the same with the placement switched off"""
    m = _mfx()
    peak = 11.0
    r = np.random.default_rng(700 + k)
    sizes = (30000, 4096, 4097, 4095 + k, 8192 + k - 1, 500, k, k - 1, 0, 12345)
    contigs, read, asm = synth.world(k=k, peak=peak, seed=800 + k, sizes=sizes)
    # low-complexity inserts: homopolymers and dinucleotide repeats longer than a k-mer, both strands' worth
    contigs = list(contigs)
    c0 = bytearray(contigs[0])
    for at, unit in ((1000, b"A"), (2000, b"AC"), (3000, b"T"), (4090, b"GA"), (8180, b"ACG"), (12000, b"C")):
        rep = (unit * 80)[:70]
        c0[at:at + len(rep)] = rep
    contigs[0] = bytes(c0)
    # the assembly's own k-mers, recounted by the plain-Python oracle (the inserts changed them); the reads keep their counts
    import oracle.plain as plain
    p = po.Params(k, peak)
    amer = plain.count_kmers(k, [c.decode() for c in contigs])
    ks = np.array(sorted(amer), dtype=np.uint64)
    asm = (ks, np.array([amer[int(x)] for x in ks], dtype=np.uint32))
    g, ka, km_, _ = po.hist_run(p, po.Lookup(k, *read), po.Lookup(k, *asm), contigs, threads=2)
    # (MFX_MZ_W=5: five windows of k - 4 bases sampled by a t-mer with (k - t) % 5 == 4 -- the direct form (k <= 21) takes it, the
    # quotient form is built on four windows and falls back to 16-byte slots)
    envs = ({}, {"MFX_HIST_GENERIC": "1"}, {"MFX_MZ_MOD": "0"}, {"MFX_MZ_W": "5"}, {"MFX_MZ_W": "5", "MFX_HIST_GENERIC": "1"})
    for env in envs:
        for kk, vv in env.items():
            monkeypatch.setenv(kk, vv)
        ix, seqs = seq_index(m, k, contigs, read)
        assert ix.info()["compact"] == (k <= 21 or ("MFX_MZ_MOD" not in env and "MFX_MZ_W" not in env))     # the quotient form (k > 21) needs the mod-minimizer on four windows
        ev = m.Evaluator(ix, m.KParams(peak))
        assert_hist_equal(ev.hist(seqs), g, ka, km_, k)
        ek, er, ea = ix.export()
        np.testing.assert_array_equal(ek, asm[0])
        np.testing.assert_array_equal(ea, asm[1])
        rd = dict(zip(read[0].tolist(), read[1].tolist()))
        np.testing.assert_array_equal(er, np.array([rd.get(x, 0) for x in ek.tolist()], dtype=np.uint32))
        rv, av = ix.value(asm[0][::7])                            # by-key lookups walk the same placement
        np.testing.assert_array_equal(av, asm[1][::7])
        for kk in env:
            monkeypatch.delenv(kk)


def _export_sorted(ix):
    k_, r_, a_ = ix.export()
    o = np.argsort(k_, kind="stable") if k_.ndim == 1 else np.lexsort((k_[:, 0], k_[:, 1]))
    return k_[o], r_[o], a_[o]


@pytest.mark.parametrize("k,form,env", [(21, "flat", {}), (31, "flat", {}), (21, "text", {}),
                                        (21, "flat", {"MFX_INGEST_CHUNK_LOG2": "16", "MFX_INGEST_RING_MB": "1"}),      # many chunks through a two-buffer ring
                                        (21, "flat", {"MFX_INGEST_CHUNK_LOG2": "16", "MFX_INGEST_RING_MB": "64"}),     # ... and through a ring deeper than the load
                                        (41, "flat", {})])
def test_build_for_hist_equals_count_then_load(k, form, env, tmp_path, monkeypatch):
    """mfx_index_build_for_hist (the claim / count kernel still running while the database's chunks cross the link; the inserts
    wait for it on the device) builds the table mfx_index_count_asm + mfx_index_load_db build, -min / -max included; the staging
    ring smaller and larger than the load; the 128-bit tables through the same call"""
    from oracle import plain
    m = _mfx()
    peak = 17.3
    if k > 31:
        from tests.test_gpu_wide import small_world, to_rows
        wc, R, _A = small_world(k, 77, n=7000)
        contigs = [c.encode() for c in wc]
        rk = sorted(R)
        read = (to_rows(rk), np.array([R[x] for x in rk], dtype=np.uint32))
        asm = None
    else:
        contigs, read, asm = synth.world(k=k, peak=peak, seed=21 + k)
    if k <= 31:
        # enough k-mers for several chunks of 2^16: the read database repeated with other counts would change nothing (update-only
        # of the same k-mers), so pad it with k-mers that are NOT in the sequence (dropped by the index, still staged and decoded)
        rng = np.random.default_rng(5)
        extra = np.unique(rng.integers(0, 1 << (2 * k - 1), 300000, dtype=np.uint64))
        x, r = extra.copy(), np.zeros_like(extra)
        for _ in range(k):                                        # reverse complement: base order reversed, code ^ 2
            r = (r << np.uint64(2)) | ((x & np.uint64(3)) ^ np.uint64(2))
            x >>= np.uint64(2)
        canon = np.minimum(extra, r)
        canon = np.setdiff1d(np.unique(canon), read[0])
        keys = np.concatenate([read[0], canon])
        vals = np.concatenate([read[1], np.full(len(canon), 3, dtype=np.uint32)])
        o = np.argsort(keys, kind="stable")
        keys, vals = keys[o], vals[o]
    else:
        keys, vals = read
    path = str(tmp_path / ("r.mfxk" if form == "flat" else "r.txt"))
    if form == "flat":
        m.db_write_flat(path, k, keys, vals)
    else:
        with open(path, "w") as f:
            for kk, vv in zip(keys.tolist(), vals.tolist()):
                f.write("%s\t%d\n" % (plain.dec(kk, k), vv))
    for name, value in env.items():
        monkeypatch.setenv(name, value)
    seqs = m.Sequences(contigs)
    lo, hi = 2, 1000

    def two_calls():
        ix = m.Index.for_seq(k, sum(len(c) for c in contigs) + 16) if k <= 31 else m.Index(k, len(keys) + sum(len(c) for c in contigs) + 16)
        ix.count_asm(seqs)
        ix.load_db(path, 0, lo, hi)
        return ix

    def one_call(overlap):
        monkeypatch.setenv("MFX_BUILD_OVERLAP", overlap)
        ix = m.Index.for_seq(k, sum(len(c) for c in contigs) + 16) if k <= 31 else m.Index(k, len(keys) + sum(len(c) for c in contigs) + 16)
        ix.build_for_hist(seqs, path, lo, hi)
        monkeypatch.delenv("MFX_BUILD_OVERLAP")
        return ix
    want = _export_sorted(two_calls())
    for overlap in ("1", "0", "1"):
        ix = one_call(overlap)
        got = _export_sorted(ix)
        for a_, b_ in zip(got, want):
            np.testing.assert_array_equal(a_, b_)
        # the index stays usable: a second load through the same lanes adds the same counts again
        ix.load_db(path, 0, lo, hi)
        again = _export_sorted(ix)
        np.testing.assert_array_equal(again[0], want[0])
        rv2 = want[1].astype(np.uint64) * 2
        np.testing.assert_array_equal(again[1].astype(np.uint64), rv2)
    if k <= 31:
        p, g, ka, km = oracle_hist(k, peak, contigs, (read[0], np.where((read[1] >= lo) & (read[1] <= hi), read[1], 0).astype(np.uint32)), asm)
        assert_hist_equal(m.Evaluator(one_call("1"), m.KParams(peak)).hist(seqs), g, ka, km, k)
    # the STAGED form (mfx_db_stage_begin + mfx_index_build_for_hist_staged: the database moves into device memory on a thread of its own from
    # the moment the stage exists): same table, for the delta-coded flat form; every other form is not staged and says so
    stage = m.DbStage(path)
    if form == "flat" and k <= 31:
        assert stage.ok, stage.why
        for load_factor in (0.0, 0.4):
            st = stage if load_factor == 0.0 else m.DbStage(path)
            ix = m.Index.for_seq(k, sum(len(c) for c in contigs) + 16, load_factor=load_factor)
            ix.build_for_hist_staged(seqs, st, lo, hi)
            st.close()
            for a_, b_ in zip(_export_sorted(ix), want):
                np.testing.assert_array_equal(a_, b_)
            assert_hist_equal(m.Evaluator(ix, m.KParams(peak)).hist(seqs), g, ka, km, k)
        # a stage that is never used is released cleanly (its worker may still be on the file)
        m.DbStage(path).close()
    else:
        assert not stage.ok and stage.why


def test_small_genome_at_k31_under_a_memory_limit():
    """the quotient form of the compact layout needs 2^(2(k-3)-31) table lines whatever the genome (4 GB at k = 31): a small genome
    under a -memory limit below that takes the 16-byte slots sized by the genome instead of failing (advisor, round 4), same results"""
    m = _mfx()
    k, peak = 31, 11.0
    contigs, read, asm = synth.world(k=k, peak=peak, seed=4242, sizes=(30000, 4097, 500))
    nb = sum(len(c) for c in contigs)
    assert m.load_library().mfx_index_estimate_gb_for_seq(k, nb + 16) < 0.1
    p, g, ka, km = oracle_hist(k, peak, contigs, read, asm)
    for max_gb, compact in ((1.0, False), (0.0, True)):
        ix = m.Index.for_seq(k, nb + 16, max_gb=max_gb)
        assert bool(ix.info()["compact"]) == compact
        seqs = m.Sequences(contigs)
        ix.count_asm(seqs)
        ix.add_read(*read)
        assert_hist_equal(m.Evaluator(ix, m.KParams(peak)).hist(seqs), g, ka, km, k)
    with pytest.raises(m.MfxError, match="Not enough memory"):
        m.Index.for_seq(k, nb + 16, max_gb=1e-5)
