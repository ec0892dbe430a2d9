"""CPU-side checks of the C-ABI library: it loads, exports every symbol the
header declares, its host-side K* helpers agree bit-for-bit with the oracle,
and it refuses to run without a GPU (no CPU fallback)."""
import os
import re

import numpy as np
import pytest

from oracle import pyoracle as po

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _lib():
    import merfin_amd as m
    from merfin_amd import binding
    return m, binding, m.load_library()


def test_library_exports_every_declared_symbol():
    m, binding, L = _lib()
    hdr = open(os.path.join(ROOT, "include", "merfin_amd.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(mfx_[a-z_A-Z0-9]+)\s*\(", hdr))
    assert declared == set(binding.SYMBOLS), declared ^ set(binding.SYMBOLS)
    for s in sorted(declared):
        assert hasattr(L, s), s
    assert b"gfx950" in L.mfx_version()


def test_host_kstar_helpers_match_oracle(golden_dir):
    m, _, _ = _lib()
    K, P = po.load_kmetric(os.path.join(golden_dir, "example_lookup_table.txt"))
    for kp, op in ((m.KParams(30.0), po.Params(21, 30.0)), (m.KParams(26.0, K, P), po.Params(21, 26.0, K, P)),
                   (m.KParams(10.5), po.Params(21, 10.5))):
        for rv in list(range(0, 260)) + [1000, 65535, 2**32 - 1]:
            for av in (0, 1, 2, 3, 7, 1000):
                a = m.getK(kp, rv, av)
                assert a == po.getK_values(op, rv, av)
                assert m.getKmetric(a[0], a[1]) == po.getKmetric(a[0], a[1])
    assert m.histoQV(100, 1e6, 21) == po.histoQV(100, 1e6, 21)
    kp2 = m.KParams.from_file(26.0, os.path.join(golden_dir, "example_lookup_table.txt"))
    assert list(kp2.probK) == list(K) and list(kp2.probP) == list(P)


def test_no_cpu_fallback_without_gpu():
    m, _, _ = _lib()
    if m.device_count() > 0:
        pytest.skip("a GPU is visible here")
    with pytest.raises(m.MfxError) as e:
        m.Index(21, 1000)
    assert e.value.code == -8 and "no CPU path" in str(e.value)
    with pytest.raises(m.MfxError):
        m.Sequences([b"ACGT"])


def test_k_range_is_the_references():
    """k-mers hold 2k <= 128 bits (meryl-utility's kmer): k = 64 is accepted (and then needs a GPU), k = 65 is not"""
    m, _, _ = _lib()
    with pytest.raises(m.MfxError) as e:
        m.Index(65, 10)
    assert e.value.code == -1 and "1 <= k <= 64" in str(e.value)
    with pytest.raises(m.MfxError) as e:
        m.Index(0, 10)
    assert e.value.code == -1
    assert m.load_library().mfx_index_estimate_gb(41, 7 * 10**8) == pytest.approx(2 * m.load_library().mfx_index_estimate_gb(31, 7 * 10**8), rel=1e-6)


def test_product_never_touches_the_oracle():
    """Nothing under merfin_amd/ may import, link or execute oracle/."""
    for d, _, files in os.walk(os.path.join(ROOT, "merfin_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h", ".hpp", "Makefile", ".txt")):
                s = open(os.path.join(d, f), errors="ignore").read()
                assert "oracle" not in s.replace("no CPU fallback", ""), os.path.join(d, f)


def test_compressed_files_with_shell_metacharacters_in_the_name(tmp_path):
    """compressedFileWriter / compressedFileReader (merfin-histogram.C:151, merfin.C:195) start the compressor with an
    argv array (csrc/mfx_pipe.h): a quote, a space or a `;` in a path is part of the name, never shell syntax.  Host-only
    entry points, so this runs without a GPU."""
    import gzip
    m, _, _ = _lib()
    from merfin_amd import distributed as D
    nb = 2048
    undr = np.zeros(8, dtype=np.uint64)
    over = np.zeros(8, dtype=np.uint64)
    undr[[0, 3]] = (5, 2)
    over[[0, 7]] = (9, 1)
    h = D.pack_counts(nb, 1, undr, over, 17, 0, [17], [0])
    res = m.result_from_counts(nb, h, 0.25, 1)
    d = tmp_path / "it's a; dir"
    d.mkdir()
    weird = str(d / "o'u t$(touch pwned).hist.gz")
    res.report(21, weird, None)
    res.report(21, str(tmp_path / "plain.hist"), None)
    assert gzip.open(weird, "rb").read() == (tmp_path / "plain.hist").read_bytes() == b"-0.6\t2\n0.0\t14\n1.4\t1\n"
    assert not os.path.exists("pwned") and not (tmp_path / "pwned").exists()
    # read side: a `meryl print` text database behind gzip, same kind of name
    db = str(d / "re'ad \"db\".txt.gz")
    with gzip.open(db, "wb") as f:
        f.write(b"ACGTA\t3\nCCCCC\t9\n")
    assert m.db_probe(db) == {"k": 5, "format": "text", "n_kmers": 2}
    # a compressor that fails is an I/O error, not a silent success
    bad = str(tmp_path / "truncated.txt.gz")
    open(bad, "wb").write(open(db, "rb").read()[:-6])
    with pytest.raises(m.MfxError):
        m.db_probe(bad)
    # an output directory that does not exist
    with pytest.raises(m.MfxError):
        res.report(21, str(tmp_path / "no such dir" / "x.hist.gz"), None)


def test_sequence_packer_bodies_agree(monkeypatch):
    """mfx_pack_bases (host side of the packed sequence transport): the vector bodies the CPU offers equal the scalar
    definition on every byte value, every length mod 64, and the documented bit layout (first base in the highest bits;
    code = (c >> 1) & 3; valid iff ACGTacgt -- the device's own tile encoding, csrc/mfx_device.h)"""
    import merfin_amd as m
    r = np.random.default_rng(5)
    allbytes = bytes(range(256))
    text = allbytes + bytes(r.choice(np.frombuffer(b"ACGTacgtNn\x00 -*", dtype=np.uint8), size=5000)) + allbytes[::-1] + b"ACGT" * 40

    def definition(b):
        nw = (len(b) + 31) // 32
        codes, valid = [0] * nw, [0] * nw
        for i, c in enumerate(b):
            codes[i // 32] |= ((c >> 1) & 3) << (62 - 2 * (i % 32))
            valid[i // 32] |= (1 if chr(c) in "ACGTacgt" else 0) << (31 - (i % 32))
        return np.array(codes, dtype=np.uint64), np.array(valid, dtype=np.uint32)

    for isa in ("scalar", "avx2", None):               # None: the best body this CPU has (AVX-512 VBMI where present)
        if isa:
            monkeypatch.setenv("MFX_PACK_ISA", isa)
        else:
            monkeypatch.delenv("MFX_PACK_ISA", raising=False)
        for cut in list(range(0, 70)) + [len(text) - 3, len(text)]:
            c, v = m.pack_bases(text[:cut])
            dc, dv = definition(text[:cut])
            assert np.array_equal(c, dc) and np.array_equal(v, dv), (isa, cut)
        c, v = m.pack_bases(text[7:])                      # unaligned source address
        dc, dv = definition(text[7:])
        assert np.array_equal(c, dc) and np.array_equal(v, dv)
    c, v = m.pack_bases(b"ACGTNacgt")
    assert int(c[0]) == 0x1EC7800000000000 and int(v[0]) == 0xF7800000


@pytest.mark.parametrize("piece", [64, 300, 1 << 25])
def test_text_database_probe_needs_no_gpu(tmp_path, piece, monkeypatch):
    """mfx_db_probe of `meryl print` text (host only): k and the number of k-mers for every piece size of the parallel
    reader, blank and CR LF lines skipped, malformed lines reported with their byte offset; .gz goes through the serial reader"""
    import gzip
    import merfin_amd as m
    monkeypatch.setenv("MFX_TEXT_PIECE", str(piece))
    r = np.random.default_rng(piece)
    k = 21
    lines = ["".join(r.choice(list("ACGT"), size=k)) + "\t%d" % v for v in r.integers(1, 10**6, size=2000)]
    body = "\r\n".join(lines[:100]) + "\r\n\r\n" + "\n".join(lines[100:1500]) + "\n\n\n" + "\n".join(lines[1500:])   # no newline at the end
    p = str(tmp_path / "db.txt")
    open(p, "w", newline="").write(body)
    assert m.db_probe(p) == {"k": k, "format": "text", "n_kmers": 2000}
    with gzip.open(p + ".gz", "wt", newline="") as f:
        f.write(body)
    assert m.db_probe(p + ".gz") == {"k": k, "format": "text", "n_kmers": 2000}
    for at, bad, what in ((0, "ACGTN\t3", "expected '<kmer>"), (1999, lines[5][:10] + "\t7", "differs from"), (700, lines[3].replace("\t", "\tx"), "expected '<kmer>")):
        broken = list(lines)
        broken[at] = bad
        bp = str(tmp_path / "broken.txt")
        open(bp, "w").write("\n".join(broken) + "\n")
        off = len("\n".join(broken[:at])) + (1 if at else 0)
        with pytest.raises(m.MfxError) as e:
            m.db_probe(bp)
        assert what in str(e.value) and ("at byte %d" % off) in str(e.value)


def test_vcf_loads_without_a_device(tmp_path):
    """mfx_vcf_load is host work only (vcfFile::loadFile, vcf.C:93-149): it succeeds with no GPU in the machine, fails with a
    message on a missing file, and the handle is released without ever meeting an evaluator"""
    import merfin_amd as m
    g = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "case1.vcf")
    v = m.LoadedVcf(g)
    assert v.h
    v.close()
    with pytest.raises(m.MfxError):
        m.LoadedVcf(str(tmp_path / "missing.vcf"))
