"""k-mer database ingest (disk -> device index): flat binary, `meryl print`
text, and a meryl-layout directory written by tests/meryl_layout.py."""
import numpy as np
import pytest

from oracle import pyoracle as po
from tests import meryl_layout, synth
from tests.test_cli import _write_text_db


@pytest.mark.gpu
@pytest.mark.parametrize("k,prefix_bits", [(21, 12), (15, 8), (31, 14)])
def test_three_db_forms_load_identically(tmp_path, k, prefix_bits):
    import merfin_amd as m
    contigs, read, asm = synth.world(k=k, seed=51, sizes=(6000, 2500, 300), err_kmers=300)
    rk, rv = read
    flat, text, mdir = str(tmp_path / "r.mfxk"), str(tmp_path / "r.txt"), str(tmp_path / "r.meryl")
    m.db_write_flat(flat, k, rk, rv)
    _write_text_db(text, k, rk, rv)
    meryl_layout.write_db(mdir, k, rk, rv, prefix_bits=prefix_bits)
    for path, fmt in ((flat, "flat"), (text, "text"), (mdir, "meryl")):
        info = m.db_probe(path)
        assert info == {"k": k, "format": fmt, "n_kmers": len(rk)}
        ix = m.Index(k, len(rk) + 16)
        ix.load_db(path, 0, 3, 60)                       # -min 3 -max 60
        ek, er, ea = ix.export()
        np.testing.assert_array_equal(ek, rk)            # raw values are stored ...
        np.testing.assert_array_equal(er, rv)
        got, _ = ix.value(rk)                            # ... and filtered at query time
        np.testing.assert_array_equal(got, np.where((rv >= 3) & (rv <= 60), rv, 0))


@pytest.mark.gpu
def test_db_errors_are_reported(tmp_path):
    import merfin_amd as m
    with pytest.raises(m.MfxError) as e:
        m.db_probe(str(tmp_path / "nope"))
    assert e.value.code == -6
    bad = tmp_path / "bad.meryl"
    bad.mkdir()
    (bad / "merylIndex").write_bytes(b"\0" * 256)
    with pytest.raises(m.MfxError) as e:
        m.db_probe(str(bad))
    assert e.value.code == -7                            # refuses, never guesses
    (tmp_path / "t.txt").write_text("ACGTACGTA\t5\nACGT\t3\n")
    with pytest.raises(m.MfxError) as e:
        m.db_probe(str(tmp_path / "t.txt"))
    assert e.value.code == -7 and "differs" in str(e.value)
    k = 9
    m.db_write_flat(str(tmp_path / "k9.mfxk"), k, np.array([5, 9], dtype=np.uint64), np.array([1, 2], dtype=np.uint32))
    ix = m.Index(11, 100)
    with pytest.raises(m.MfxError) as e:
        ix.load_db(str(tmp_path / "k9.mfxk"), 0)
    assert e.value.code == -1


@pytest.mark.gpu
def test_corrupt_meryl_piece_is_reported_from_the_decoder_threads(tmp_path, monkeypatch):
    """the 64 data files are decoded on several host threads; a failure in any of them must reach the caller
    (code + the message naming the lowest-numbered bad file), and a single-threaded run must say the same"""
    import glob
    import os
    import merfin_amd as m
    k = 21
    contigs, read, asm = synth.world(k=k, seed=52, sizes=(20000,), err_kmers=0)
    mdir = str(tmp_path / "r.meryl")
    meryl_layout.write_db(mdir, k, read[0], read[1], prefix_bits=12)
    files = sorted(f for f in glob.glob(mdir + "/*.merylData") if os.path.getsize(f) > 200)
    assert len(files) >= 8
    for f in (files[5], files[2]):                           # damage two pieces: the payload after the block tables
        raw = bytearray(open(f, "rb").read())
        raw[-64:] = b"\xff" * 64
        raw[40:48] = b"\x00" * 8
        open(f, "wb").write(bytes(raw))
    msgs = []
    for threads in ("8", "1"):
        monkeypatch.setenv("MFX_HOST_THREADS", threads)
        ix = m.Index(k, len(read[0]) + 16)
        with pytest.raises(m.MfxError) as e:
            ix.load_db(mdir, 0)
        assert e.value.code == -7
        msgs.append(str(e.value))
    assert msgs[0] == msgs[1] and os.path.basename(files[2]) in msgs[0]


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["mz", "plain"])
def test_index_image_round_trip(tmp_path, mode, monkeypatch):
    """save -> load of the device-format index: identical contents and identical -hist."""
    import merfin_amd as m
    from tests.test_gpu_parity import assert_hist_equal, oracle_hist
    monkeypatch.setenv("MFX_HOME_MODE", mode)
    k, peak = 21, 17.3
    contigs, read, asm = synth.world(k=k, peak=peak, seed=91)
    ix = m.Index(k, len(read[0]) + len(asm[0]) + 16)
    ix.add_read(read[0], read[1], 2, 500)
    ix.add_asm(*asm)
    img = str(tmp_path / "index.mfxi")
    ix.save(img)
    monkeypatch.setenv("MFX_HOME_MODE", "plain" if mode == "mz" else "mz")      # the image carries its own placement
    ix2 = m.Index.load(img)
    assert ix2.info() == ix.info()
    for a, b in zip(ix.export(), ix2.export()):
        np.testing.assert_array_equal(a, b)
    p, g, ka, km = oracle_hist(k, peak, contigs, read, asm, minV=2, maxV=500)
    res = m.Evaluator(ix2, m.KParams(peak)).hist(m.Sequences(contigs))
    assert_hist_equal(res, g, ka, km, k)
    (tmp_path / "bad").write_bytes(b"not an image" * 10)
    with pytest.raises(m.MfxError) as e:
        m.Index.load(str(tmp_path / "bad"))
    assert e.value.code == -7


@pytest.mark.gpu
def test_in_memory_image_moves_a_built_table(monkeypatch):
    """header -> empty index of the same geometry -> raw copy of lines + meta -> commit: what broadcast_index does
    between ranks, here between two indexes on one GPU; contents and -hist must be identical"""
    torch = pytest.importorskip("torch")
    import merfin_amd as m
    from merfin_amd.distributed import _DeviceBytes
    k, peak = 21, 17.3
    contigs, read, asm = synth.world(k=k, peak=peak, seed=57)
    src = m.Index(k, len(read[0]) + len(asm[0]) + 16)
    src.add_read(read[0], read[1], 2, 1000)
    src.add_asm(*asm)
    hdr = src.image_header()
    assert hdr.nbytes == m.binding.INDEX_HEADER_BYTES
    dst = m.Index.from_header(hdr)
    sl, snl, sm, snm = src.device_image()
    dl, dnl, dm, dnm = dst.device_image()
    assert (snl, snm) == (dnl, dnm) and dl != sl
    for a, b, n in ((sl, dl, snl), (sm, dm, snm)):
        torch.as_tensor(_DeviceBytes(b, n), device="cuda").copy_(torch.as_tensor(_DeviceBytes(a, n), device="cuda"))
    torch.cuda.synchronize()
    dst.commit()
    assert dst.info() == src.info()
    for x, y in zip(src.export(), dst.export()):
        np.testing.assert_array_equal(x, y)
    seqs = m.Sequences(contigs)
    ra, rb = m.Evaluator(src, m.KParams(peak)).hist(seqs), m.Evaluator(dst, m.KParams(peak)).hist(seqs)
    assert (ra.kasm, ra.kmissing, ra.koverCpy) == (rb.kasm, rb.kmissing, rb.koverCpy) and ra.kmissing > 0
    np.testing.assert_array_equal(ra.over(), rb.over())
    bad = hdr.copy()
    bad[0] ^= 0xff
    with pytest.raises(m.MfxError) as e:
        m.Index.from_header(bad)
    assert e.value.code == -7


def _small_meryl(tmp_path, name="r.meryl", **kw):
    k = 21
    contigs, read, asm = synth.world(k=k, seed=53, sizes=(9000,), err_kmers=100)
    mdir = str(tmp_path / name)
    meryl_layout.write_db(mdir, k, read[0], read[1], prefix_bits=12, **kw)
    return k, read, mdir


def test_meryl_self_validation_on_probe(tmp_path, monkeypatch):
    """Conformance checks of SURVEY App. C that need no GPU (mfx_db_probe reads the block headers only): the 64 data
    files must all exist, block prefixes must increase and carry their file number, and the block headers must add up
    to the master index's statistics.  Each corrupt field fails loudly with its own message."""
    import glob
    import os
    import shutil
    import merfin_amd as m
    k, read, mdir = _small_meryl(tmp_path)
    assert m.db_probe(mdir) == {"k": k, "format": "meryl", "n_kmers": len(read[0])}
    # a database without the statistics block still opens (older index versions)
    _, _, nostat = _small_meryl(tmp_path, "nostat.meryl", stats=False)
    assert m.db_probe(nostat)["n_kmers"] == len(read[0])

    def damaged(name, fn):
        d = str(tmp_path / name)
        shutil.copytree(mdir, d)
        fn(d)
        with pytest.raises(m.MfxError) as e:
            m.db_probe(d)
        return e.value

    files = sorted(f for f in glob.glob(mdir + "/*.merylData") if os.path.getsize(f) > 200)
    victim = os.path.basename(files[3])
    # (1) a missing piece (partial copy): I/O error, not "no k-mers in that piece"
    e = damaged("missing.meryl", lambda d: os.remove(os.path.join(d, victim)))
    assert e.code == -6 and victim in str(e) and "64 data files" in str(e)
    # (2) an emptied piece: the headers no longer add up to the statistics
    e = damaged("emptied.meryl", lambda d: open(os.path.join(d, victim), "wb").close())
    assert e.code == -7 and "statistics say" in str(e)
    # (3) statistics that disagree with the data
    _, _, wrong = _small_meryl(tmp_path, "wrongstat.meryl", stats_override=(0, len(read[0]) + 1, 10 ** 9))
    with pytest.raises(m.MfxError) as e2:
        m.db_probe(wrong)
    assert e2.value.code == -7 and "distinct" in str(e2.value)
    monkeypatch.setenv("MFX_MERYL_LENIENT", "1")
    assert m.db_probe(wrong)["n_kmers"] == len(read[0])        # downgraded to a warning on request
    monkeypatch.delenv("MFX_MERYL_LENIENT")
    # (4) a piece under another file's name: the block prefix's top 6 bits must equal the file number
    other = os.path.basename(files[4])

    def swap(d):
        shutil.copyfile(os.path.join(d, other), os.path.join(d, victim))
    e = damaged("swapped.meryl", swap)
    assert e.code == -7 and "inconsistent data block" in str(e)
    # (5) blocks out of order inside a file: append the file's first block again
    def dup(d):
        raw = open(os.path.join(d, victim), "rb").read()
        import struct
        nbits = struct.unpack_from("<Q", raw, 24)[0]
        first = raw[:32 + 8 * ((nbits + 63) // 64)]
        open(os.path.join(d, victim), "ab").write(first)
    e = damaged("unordered.meryl", dup)
    assert e.code == -7 and "not strictly increasing" in str(e)


@pytest.mark.gpu
def test_meryl_value_statistics_checked_on_load(tmp_path):
    """a full load also decodes the values: their sum and the number of 1s must match the statistics"""
    import merfin_amd as m
    k, read, mdir = _small_meryl(tmp_path)
    ix = m.Index(k, len(read[0]) + 16)
    ix.load_db(mdir, 0)
    assert ix.info()["distinct"] == len(read[0])
    uniq, tot = int((read[1] == 1).sum()), int(read[1].astype(np.uint64).sum())
    for name, st, word in (("t.meryl", (uniq, len(read[0]), tot + 5), "total"), ("u.meryl", (uniq - 1, len(read[0]), tot), "unique")):
        _, _, d = _small_meryl(tmp_path, name, stats_override=st)
        ix2 = m.Index(k, len(read[0]) + 16)
        with pytest.raises(m.MfxError) as e:
            ix2.load_db(d, 0)
        assert e.value.code == -7 and word in str(e.value)


@pytest.mark.gpu
@pytest.mark.parametrize("k,piece", [(21, 257), (31, 1000), (41, 4096), (21, 1 << 25)])
def test_plain_text_database_is_parsed_in_pieces_by_the_host_threads(tmp_path, k, piece, monkeypatch):
    """`meryl print` text, uncompressed: cut into pieces, every piece parsed by the thread that draws it (lines belong to the
    piece they START in).  Same table as the flat form for every piece size (down to pieces shorter than ten lines), with
    blank lines, CR LF endings, a last line without newline; malformed lines and a k-mer of another length are reported with
    their byte offset wherever they sit; .gz text still goes through the serial reader."""
    import gzip
    import merfin_amd as m
    from oracle import plain
    monkeypatch.setenv("MFX_TEXT_PIECE", str(piece))
    r = np.random.default_rng(k + piece)
    n = 6000
    if k <= 31:
        kms = np.unique(r.integers(0, 1 << (2 * k), size=n, dtype=np.uint64))
        vals = r.integers(1, 2000000, size=len(kms)).astype(np.uint32)
        text = [plain.dec(int(x), k) for x in kms.tolist()]
    else:
        ints = sorted({int.from_bytes(r.bytes(16), "little") & ((1 << (2 * k)) - 1) for _ in range(n)})
        kms = np.array([[x & (2**64 - 1), x >> 64] for x in ints], dtype=np.uint64)
        vals = r.integers(1, 2000000, size=len(ints)).astype(np.uint32)
        text = [plain.dec(x, k) for x in ints]
    # the oracle's decoder spells A C G T by the reference's 2-bit codes; the text form is what `meryl print` writes
    lines = []
    for i, (t, v) in enumerate(zip(text, vals.tolist())):
        lines.append("%s\t%d" % (t, v))
        if i % 97 == 0:
            lines.append("")                                   # blank lines are skipped
    body = "\r\n".join(lines[:50]) + "\r\n" + "\n".join(lines[50:])          # CR LF for a while; no newline after the last line
    path = str(tmp_path / "db.txt")
    open(path, "w", newline="").write(body)
    info = m.db_probe(path)
    assert (info["format"], info["k"], info["n_kmers"]) == ("text", k, len(vals))
    flat = str(tmp_path / "db.mfxk")
    m.db_write_flat(flat, k, kms, vals)
    a, b = m.Index(k, len(vals) + 16), m.Index(k, len(vals) + 16)
    a.load_db(path, 0)
    b.load_db(flat, 0)
    ea, eb = a.export(), b.export()
    assert len(ea[0]) == len(vals) and all(np.array_equal(x, y) for x, y in zip(ea, eb))
    with gzip.open(path + ".gz", "wt", newline="") as f:
        f.write(body)
    c = m.Index(k, len(vals) + 16)
    c.load_db(path + ".gz", 0)
    assert all(np.array_equal(x, y) for x, y in zip(c.export(), eb))
    # errors, in the first piece and in a late one
    for at in (3, len(lines) - 5):
        for bad, what in (("ACGT" + "x" * 5 + "\t7", "expected '<kmer>"), (text[0][:-1] + "\t9", "differs from")):
            broken = list(lines)
            broken[at] = bad
            bp = str(tmp_path / "broken.txt")
            open(bp, "w").write("\n".join(broken) + "\n")
            off = len("\n".join(broken[:at])) + (1 if at else 0)
            with pytest.raises(m.MfxError, match="at byte %d" % off):
                m.Index(k, len(vals) + 16).load_db(bp, 0)
            with pytest.raises(m.MfxError, match=what.replace("(", r"\(")):
                m.db_probe(bp)
    with pytest.raises(m.MfxError, match="built for k="):
        m.Index(k - 2, 100).load_db(path, 0)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", list(range(int(__import__("os").environ.get("MFX_RANDOM_SEEDS", "8")))))
def test_randomized_database_forms_agree(tmp_path, seed, monkeypatch):
    """seeded sweep: k (narrow and 128-bit), number of k-mers (also 0 in most of the 64 pieces), values up to 2^32-1, the
    meryl layout's prefix width, the text reader's piece size, one table or the shards of one process -- flat, text and
    meryl-layout forms of one database always give the same table"""
    import merfin_amd as m
    from oracle import plain
    r = np.random.default_rng(600 + seed)
    k = int(r.choice([7, 11, 16, 21, 27, 31, 33, 48, 64]))
    n = int(r.choice([1, 40, 3000, 20000]))
    space = 1 << (2 * k)
    ints = sorted({int.from_bytes(r.bytes(16), "little") % space for _ in range(n)})
    vals = r.choice([1, 2, 3, 17, 255, 256, 65535, 65536, 2**32 - 1], size=len(ints)).astype(np.uint32)
    wide = k > 31
    keys = (np.array([[x & (2**64 - 1), x >> 64] for x in ints], dtype=np.uint64) if wide else np.array(ints, dtype=np.uint64))
    flat, text, mdir = str(tmp_path / "d.mfxk"), str(tmp_path / "d.txt"), str(tmp_path / "d.meryl")
    m.db_write_flat(flat, k, keys, vals)
    with open(text, "w") as f:
        for x, v in zip(ints, vals.tolist()):
            f.write("%s\t%d\n" % (plain.dec(x, k), v))
    prefix_bits = int(r.integers(6, min(2 * k - 4, 16) + 1))
    meryl_layout.write_db(mdir, k, keys, vals, prefix_bits=prefix_bits)
    monkeypatch.setenv("MFX_TEXT_PIECE", str(int(r.choice([64, 999, 1 << 25]))))
    lo, hi = int(r.choice([0, 2, 300])), int(r.choice([250, 70000, 2**32 - 1]))
    world = 1 if wide else int(r.choice([1, 1, 3]))
    tables = []
    for path in (flat, text, mdir):
        assert m.db_probe(path)["n_kmers"] == len(ints) and m.db_probe(path)["k"] == k
        shards = []
        for rank in range(world):
            ix = m.Index(k, len(ints) + 16)
            if world > 1:
                ix.set_shard(rank, world)
            shards.append(ix)
        if world > 1:
            m.load_db_multi(shards, path, 0, lo, hi)
        else:
            shards[0].load_db(path, 0, lo, hi)
        parts = [s.export() for s in shards]
        assert sum(len(p[0]) for p in parts) == len(ints)
        ks = sum(((meryl_layout.to_ints(p[0]) if wide else p[0].tolist()) for p in parts), [])
        vs = sum((p[1].tolist() for p in parts), [])
        merged = sorted(zip(ks, vs))
        tables.append(merged)
    assert tables[0] == tables[1] == tables[2] == sorted(zip(ints, vals.tolist()))
    if world == 1:
        ix = m.Index(k, len(ints) + 16)
        ix.load_db(flat, 0, lo, hi)
        got, _ = ix.value(keys)
        assert got.tolist() == [v if lo <= v <= hi else 0 for v in vals.tolist()]


@pytest.mark.gpu
@pytest.mark.parametrize("k", [21, 15, 31])
def test_flat_forms_give_the_same_tables(tmp_path, k, monkeypatch):
    """the flat form of one database in its three encodings -- plain (8 + 4 bytes per k-mer), packed records {k-mer << 22 |
    count} for k <= 21 (counts >= 2^22 - 1 escape to a side list), delta-coded blocks of the sorted k-mers (decoded by the
    inserting kernel; counts beyond a block's field escape) -- gives the same full table and the same sequence-only table,
    escapes and zero counts included"""
    import os
    import merfin_amd as m
    from tests import synth
    contigs, read, asm = synth.world(k=k, peak=9.0, seed=77)
    rv = read[1].astype(np.uint64)
    rv[::37] = np.array([2**22 - 2, 2**22 - 1, 2**22, 2**31, 2**32 - 1] * (len(rv[::37]) // 5 + 1), dtype=np.uint64)[:len(rv[::37])]
    rv[5::41] = 0                                                    # a zero count: never stored
    read = (read[0], rv.astype(np.uint32))
    assert (np.diff(read[0].astype(np.uint64)) > 0).all()            # sorted: the writer's condition for the delta form
    n = len(read[0])
    forms = {"plain": ("0", "0"), "packed": ("0", "1"), "delta": ("1", "1")}
    if k > 21:
        del forms["packed"]
    paths = {}
    for name, (delta, packed) in forms.items():
        monkeypatch.setenv("MFX_FLAT_DELTA", delta)
        monkeypatch.setenv("MFX_FLAT_PACKED", packed)
        paths[name] = str(tmp_path / ("r_%s.mfxk" % name))
        m.db_write_flat(paths[name], k, *read)
    assert os.path.getsize(paths["plain"]) == 32 + 12 * n
    n_esc = int((read[1] >= 2**22 - 1).sum())
    if "packed" in paths:
        assert os.path.getsize(paths["packed"]) == 32 + 8 * n + 12 * n_esc and n_esc > 10
    assert os.path.getsize(paths["delta"]) < 0.75 * os.path.getsize(paths["plain"])
    seqs = m.Sequences(contigs)
    tables = []
    for name in forms:
        assert m.db_probe(paths[name]) == {"k": k, "format": "flat", "n_kmers": n}
        full = m.Index(k, n + 16)
        full.load_db(paths[name], 0)
        so = m.Index.for_seq(k, sum(len(c) for c in contigs) + 16)
        so.count_asm(seqs)
        so.load_db(paths[name], 0)
        tables.append((full.export(), so.export()))
    for other in tables[1:]:
        for a, b in zip(tables[0], other):
            for x, y in zip(a, b):
                np.testing.assert_array_equal(x, y)
    ek, er, _ = tables[-1][0]
    keep = read[1] > 0
    np.testing.assert_array_equal(ek, read[0][keep])
    np.testing.assert_array_equal(er, read[1][keep])


@pytest.mark.gpu
def test_delta_form_many_blocks_and_a_sharded_load(tmp_path):
    """a delta-coded database of many blocks (a partial last block, dense and sparse stretches, blocks whose counts all
    escape) loads as its arrays do -- into a whole table and, in one pass, into the three shards of one process"""
    import merfin_amd as m
    rng = np.random.default_rng(5)
    k = 21
    n = 3 * 4096 * 7 + 1234
    keys = np.unique(rng.integers(0, 1 << 42, size=n + 5000, dtype=np.uint64))[:n]
    keys[4096:8192] = keys[4096] + np.arange(4096, dtype=np.uint64)          # a block of consecutive k-mers: 1-bit differences
    keys = np.unique(keys)
    # canonical only (the tables hold canonical k-mers): fold every k-mer onto min(fwd, rc) and re-sort
    x, rc = keys.copy(), np.zeros_like(keys)
    for _ in range(k):                                                        # complement of a 2-bit code: code ^ 2
        rc = (rc << np.uint64(2)) | ((x & np.uint64(3)) ^ np.uint64(2))
        x = x >> np.uint64(2)
    keys = np.unique(np.minimum(keys, rc))
    n = len(keys)
    vals = rng.integers(1, 60, size=n).astype(np.uint32)
    vals[8192:12288] = 2**31 + np.arange(4096, dtype=np.uint32)               # a whole block of escapes
    vals[::1000] = 2**32 - 1
    path = str(tmp_path / "d.mfxk")
    m.db_write_flat(path, k, keys, vals)
    whole = m.Index(k, n + 16)
    whole.load_db(path, 0)
    ek, er, _ = whole.export()
    np.testing.assert_array_equal(ek, keys)
    np.testing.assert_array_equal(er, vals)
    shards = [m.Index(k, n + 16) for _ in range(3)]
    for r, ix in enumerate(shards):
        ix.set_shard(r, 3)
    m.load_db_multi(shards, path, 0)
    got = sorted((int(a), int(b)) for ix in shards for a, b in zip(*ix.export()[:2]))
    assert got == list(zip(keys.tolist(), vals.tolist()))


@pytest.mark.gpu
def test_delta_form_with_a_damaged_directory_is_refused(tmp_path):
    """the block directory of a delta-coded file is checked as a whole before any block is read by it (offsets chained, 8-byte
    aligned, inside the file; widths in range; every block as long as its widths say): a damaged file is a format error, never
    a read outside the staging lanes or the device buffers"""
    import struct
    import merfin_amd as m
    rng = np.random.default_rng(12)
    k = 21
    keys = np.unique(rng.integers(0, 1 << 42, size=30000, dtype=np.uint64))
    vals = rng.integers(1, 50, size=len(keys)).astype(np.uint32)
    good = str(tmp_path / "good.mfxk")
    m.db_write_flat(good, k, keys, vals)
    raw = bytearray(open(good, "rb").read())
    (nblocks,) = struct.unpack_from("<Q", raw, 32)
    assert nblocks == (len(keys) + 4095) // 4096 and nblocks >= 3
    d1 = 40 + 16 * 1 + 8                                             # the info word of block 1

    def damaged(name, edit, cut=None):
        b = bytearray(raw)
        edit(b)
        p = str(tmp_path / name)
        open(p, "wb").write(bytes(b[:cut] if cut else b))
        ix = m.Index(k, len(keys) + 16)
        with pytest.raises(m.MfxError) as e:
            ix.load_db(p, 0)
        assert e.value.code in (-7, -6), (name, str(e.value))         # FORMAT, IO
        assert ix.info()["distinct"] == 0                            # nothing was inserted on the way

    def put(off, fmt, v):
        return lambda b: struct.pack_into(fmt, b, off, v)

    info1 = struct.unpack_from("<Q", raw, d1)[0]
    damaged("nblocks.mfxk", put(32, "<Q", nblocks + 1))
    damaged("offset.mfxk", put(d1, "<Q", info1 + 8))                 # block 1 starts 8 bytes late
    damaged("kbits.mfxk", put(d1, "<Q", (info1 & ~(0xff << 48)) | (63 << 48)))
    damaged("vbits.mfxk", put(d1, "<Q", (info1 & ~(0xff << 56)) | (1 << 56)))
    damaged("vbits_wide.mfxk", put(d1, "<Q", (info1 & ~(0xff << 56)) | (23 << 56)))
    damaged("short.mfxk", lambda b: None, cut=len(raw) - 64)
    damaged("dir_cut.mfxk", lambda b: None, cut=40 + 16 * 2)
    damaged("n.mfxk", put(16, "<Q", len(keys) + 5000))               # more k-mers than the blocks hold
    ix = m.Index(k, len(keys) + 16)
    ix.load_db(good, 0)
    assert ix.info()["distinct"] == len(keys)


@pytest.mark.gpu
def test_records_wider_than_2k_bits_are_refused_by_the_device_loaders(tmp_path, monkeypatch):
    """a k-mer has 2k bits.  Every flat encoding with one wider record is a format error of the load (plain and packed: the
    kernels that unpack the records count them, index meta[4]; delta-coded: the k-mers of a block only ever exist in the kernel
    that decodes it) and none of the wide records reaches the table.  Damaged header counts are format errors, not aborts."""
    import struct
    import merfin_amd as m
    from tests.test_db_convert import _write_plain
    k = 15
    rng = np.random.default_rng(3)
    keys = np.unique(rng.integers(0, 1 << 30, size=9000, dtype=np.uint64))
    x, rc = keys.copy(), np.zeros_like(keys)
    for _ in range(k):
        rc = (rc << np.uint64(2)) | ((x & np.uint64(3)) ^ np.uint64(2))
        x = x >> np.uint64(2)
    keys = np.unique(np.minimum(keys, rc))
    vals = rng.integers(1, 50, size=len(keys)).astype(np.uint32)
    # plain: one key with a bit above 2k
    bad = keys.copy()
    bad[1234] |= np.uint64(1 << 33)
    p = str(tmp_path / "plain.mfxk")
    _write_plain(p, k, bad, vals)
    for make in (lambda: m.Index(k, len(keys) + 16), ):
        ix = make()
        with pytest.raises(m.MfxError) as e:
            ix.load_db(p, 0)
        assert e.value.code == -7 and "wider than 2k" in str(e.value)
        assert ix.info()["distinct"] == len(keys) - 1
    # packed records {k-mer << 22 | count}
    p = str(tmp_path / "packed.mfxk")
    with open(p, "wb") as f:
        f.write(b"MFXKMER1" + struct.pack("<IIQQ", k, 3, len(keys), 0))
        f.write(((bad << np.uint64(22)) | vals.astype(np.uint64)).astype("<u8").tobytes())
    ix = m.Index(k, len(keys) + 16)
    with pytest.raises(m.MfxError) as e:
        ix.load_db(p, 0)
    assert e.value.code == -7 and ix.info()["distinct"] == len(keys) - 1
    # delta-coded: widen the last difference of the last block until the last k-mer leaves 2k bits
    good = str(tmp_path / "good.mfxk")
    m.db_write_flat(good, k, keys, vals)
    raw = bytearray(open(good, "rb").read())
    assert struct.unpack_from("<I", raw, 12)[0] & 4
    (nblocks,) = struct.unpack_from("<Q", raw, 32)
    first, info = struct.unpack_from("<QQ", raw, 40 + 16 * (nblocks - 1))
    struct.pack_into("<Q", raw, 40 + 16 * (nblocks - 1), first | (1 << 31))       # the block's first k-mer: 32 bits > 2k = 30
    p = str(tmp_path / "delta.mfxk")
    open(p, "wb").write(bytes(raw))
    ix = m.Index(k, len(keys) + 16)
    with pytest.raises(m.MfxError) as e:
        ix.load_db(p, 0)
    assert e.value.code == -7                                                     # seen by the host's directory check
    # ... and a hand-made block whose DIFFERENCES leave 2k bits: only the decoding kernel can see that
    kb, vb = 30, 2
    d = [1, (1 << 30) - 1]
    blk = struct.pack("<QQ", d[0] | (d[1] << kb), 1 | (1 << vb) | (1 << 2 * vb))
    off = 32 + 8 + 2 * 16
    p = str(tmp_path / "delta2.mfxk")
    with open(p, "wb") as f:
        f.write(b"MFXKMER1" + struct.pack("<IIQQ", k, 5, 3, 0) + struct.pack("<Q", 1))
        f.write(struct.pack("<QQ", 1 << 29, off | (kb << 48) | (vb << 56)) + struct.pack("<QQ", 0, off + 16))
        f.write(blk)
    ix = m.Index(k, 64)
    with pytest.raises(m.MfxError) as e:
        ix.load_db(p, 0)
    assert e.value.code == -7 and "wider than 2k" in str(e.value)
    assert ix.info()["distinct"] == 2                                             # the two k-mers inside 2k bits
    # header counts that would wrap the size arithmetic
    for n, nesc, flags in ((1 << 62, 0, 1), (3, 1 << 62, 3), (3, 1 << 61, 5)):
        p = str(tmp_path / "hdr.mfxk")
        _write_plain(p, k, keys[:3], vals[:3], n=n, n_escape=nesc, flags=flags)
        ix = m.Index(k, 64)
        with pytest.raises(m.MfxError) as e:
            ix.load_db(p, 0)
        assert e.value.code == -7, str(e.value)
