"""mfx_db_convert / `merfin -convert`: every accepted form of a k-mer database (meryl-shaped directory, `meryl print` text plain
and .gz, the three flat encodings) rewritten as the flat form on the host -- sorted, so delta-coded -- and read back by the
independent decoder of tests/test_flat_delta.py.  No device is involved."""
import gzip
import os
import subprocess

import numpy as np
import pytest

from tests import meryl_layout
from tests.test_flat_delta import decode_delta, read_flat

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "merfin_amd", "bin", "merfin")


def kmer_text(k, km):
    return "".join("ACTG"[(km >> (2 * (k - 1 - i))) & 3] for i in range(k))


def database(k, n, seed):
    r = np.random.default_rng(seed)
    keys = np.unique(r.integers(0, 1 << min(62, 2 * k), size=n * 2, dtype=np.uint64))[:n]
    vals = r.integers(1, 80, size=len(keys)).astype(np.uint32)
    vals[::53] = r.integers(2**22 - 2, 2**32 - 1, size=len(vals[::53]), dtype=np.uint64).astype(np.uint32)
    return keys, vals


@pytest.mark.parametrize("k", [21, 31, 12])
def test_every_form_converts_to_the_same_flat_file(tmp_path, k, monkeypatch):
    import merfin_amd as m
    keys, vals = database(k, 30000, 100 + k)
    n = len(keys)
    src = {}
    # `meryl print` text, plain (parsed by all host threads: small pieces) and compressed
    text = "".join("%s\t%d\n" % (kmer_text(k, int(a)), int(b)) for a, b in zip(keys.tolist(), vals.tolist()))
    src["text"] = str(tmp_path / "db.txt")
    open(src["text"], "w").write(text)
    src["text.gz"] = str(tmp_path / "db.txt.gz")
    gzip.open(src["text.gz"], "wt").write(text)
    # the flat encodings
    for name, (delta, packed) in {"plain": ("0", "0"), "packed": ("0", "1"), "delta": ("1", "1")}.items():
        if name == "packed" and k > 21:
            continue
        monkeypatch.setenv("MFX_FLAT_DELTA", delta)
        monkeypatch.setenv("MFX_FLAT_PACKED", packed)
        src[name] = str(tmp_path / ("db_%s.mfxk" % name))
        m.db_write_flat(src[name], k, keys, vals)
    monkeypatch.delenv("MFX_FLAT_DELTA")
    monkeypatch.delenv("MFX_FLAT_PACKED")
    # a meryl-shaped directory (tests/meryl_layout.py: the recalled layout)
    src["meryl"] = str(tmp_path / "db.meryl")
    meryl_layout.write_db(src["meryl"], k, keys, vals)
    monkeypatch.setenv("MFX_TEXT_PIECE", "4096")
    monkeypatch.setenv("MFX_HOST_THREADS", "5")
    ref = None
    for name, path in src.items():
        out = str(tmp_path / ("out_%s.mfxk" % name.replace(".", "_")))
        assert m.db_convert(path, out) == n, name
        kk, dk, dv = decode_delta(out)
        assert kk == k and dk == keys.tolist() and dv == vals.tolist(), name
        data = open(out, "rb").read()
        ref = ref or data
        assert data == ref, name                                     # byte-identical whatever the source was


def test_unsorted_text_is_sorted_and_duplicates_are_refused(tmp_path, monkeypatch):
    import merfin_amd as m
    k = 17
    keys, vals = database(k, 20000, 7)
    order = np.random.default_rng(1).permutation(len(keys))
    p = str(tmp_path / "shuffled.txt")
    open(p, "w").write("".join("%s\t%d\n" % (kmer_text(k, int(keys[i])), int(vals[i])) for i in order))
    monkeypatch.setenv("MFX_TEXT_PIECE", "8192")
    out = str(tmp_path / "o.mfxk")
    assert m.db_convert(p, out) == len(keys)
    kk, dk, dv = decode_delta(out)
    assert dk == keys.tolist() and dv == vals.tolist()
    open(p, "a").write("%s\t3\n" % kmer_text(k, int(keys[5])))
    with pytest.raises(m.MfxError) as e:
        m.db_convert(p, out)
    assert "twice" in str(e.value)
    with pytest.raises(m.MfxError):
        m.db_convert(str(tmp_path / "missing.txt"), out)


def test_wide_k_mers_keep_their_order_in_the_plain_form(tmp_path):
    import merfin_amd as m
    k = 40
    r = np.random.default_rng(3)
    kms = sorted(set(int(r.integers(0, 1 << 62)) << 18 | int(r.integers(0, 1 << 18)) for _ in range(500)))
    p = str(tmp_path / "w.txt")
    open(p, "w").write("".join("%s\t%d\n" % (kmer_text(k, km), i + 1) for i, km in enumerate(kms)))
    out = str(tmp_path / "w.mfxk")
    assert m.db_convert(p, out) == len(kms)
    raw, kk, flags, n, n_esc = read_flat(out)
    assert kk == k and flags & 6 == 0 and n == len(kms)
    words = np.frombuffer(raw, dtype="<u8", count=2 * n, offset=32).reshape(n, 2)
    assert [int(w[0]) | (int(w[1]) << 64) for w in words] == kms
    assert np.frombuffer(raw, dtype="<u4", count=n, offset=32 + 16 * n).tolist() == list(range(1, n + 1))


def test_cli_convert_needs_no_device(tmp_path):
    import merfin_amd as m
    k = 21
    keys, vals = database(k, 5000, 11)
    p = str(tmp_path / "db.txt")
    open(p, "w").write("".join("%s\t%d\n" % (kmer_text(k, int(a)), int(b)) for a, b in zip(keys.tolist(), vals.tolist())))
    out = str(tmp_path / "db.mfxk")
    r = subprocess.run([EXE, "-convert", p, "-output", out], capture_output=True, text=True, env=dict(os.environ, HIP_VISIBLE_DEVICES="-1"))
    assert r.returncode == 0, r.stderr
    assert "Wrote %d k-mers" % len(keys) in r.stderr and "bytes per k-mer" in r.stderr
    kk, dk, dv = decode_delta(out)
    assert dk == keys.tolist() and dv == vals.tolist()
    assert m.db_probe(out) == {"k": k, "format": "flat", "n_kmers": len(keys)}
    r = subprocess.run([EXE, "-convert", p], capture_output=True, text=True)
    assert r.returncode == 1 and "No output (-output) supplied." in r.stderr


def _write_plain(path, k, keys, vals, n=None, n_escape=0, flags=1):
    import struct
    keys = np.asarray(keys, dtype="<u8")
    vals = np.asarray(vals, dtype="<u4")
    with open(path, "wb") as f:
        f.write(b"MFXKMER1" + struct.pack("<IIQQ", k, flags, len(keys) if n is None else n, n_escape))
        f.write(keys.tobytes())
        f.write(vals.tobytes())


def test_a_flat_file_with_a_k_mer_wider_than_2k_bits_is_refused(tmp_path):
    """a k-mer with bits at or above 2k once indexed past sort_pairs' bucket histogram (advisor, round 3): now MFX_E_FORMAT"""
    import merfin_amd as m
    k = 9
    keys = [5, 3, (1 << 18) | 7, 1 << 40, 2]           # not ascending (so the converter sorts) and two keys beyond 18 bits
    p = str(tmp_path / "wide.mfxk")
    _write_plain(p, k, keys, [1] * len(keys))
    with pytest.raises(m.MfxError) as e:
        m.db_convert(p, str(tmp_path / "o.mfxk"))
    assert "wider than 2k bits" in str(e.value)
    _write_plain(p, k, [5, 3, 2], [1, 2, 3])            # the same file without them converts
    assert m.db_convert(p, str(tmp_path / "o.mfxk")) == 3


@pytest.mark.parametrize("n, n_escape, flags", [(1 << 62, 0, 1), (3, 1 << 62, 3), (3, 1 << 61, 5), ((1 << 64) - 1, (1 << 64) - 1, 1),
                                                (1 << 40, 0, 3), (1 << 50, 0, 5), (3, 4, 3)])
def test_damaged_flat_headers_are_format_errors_not_aborts(tmp_path, n, n_escape, flags):
    """header counts that wrap the size checks (n_escape = 2^62: 12 * n_escape == 0 mod 2^64) used to reach std::vector's
    length_error through the C ABI (advisor, round 3)"""
    import merfin_amd as m
    p = str(tmp_path / "bad.mfxk")
    _write_plain(p, 21, [1, 2, 3], [1, 1, 1], n=n, n_escape=n_escape, flags=flags)
    with pytest.raises(m.MfxError) as e:
        m.db_convert(p, str(tmp_path / "o.mfxk"))
    assert "beyond the file's size" in str(e.value) or "more escapes" in str(e.value)


def test_damaged_meryl_blocks_are_errors_not_aborts(tmp_path):
    """every byte of the head of a data file (chunk framing, block header: prefix, number of k-mers, code widths) damaged in turn,
    then truncations: the decoder answers with an error or -- where the damage is a valid database -- a result, never with an
    exception through the C ABI (a block header claiming 2^60 k-mers used to size a vector: std::length_error, terminate)"""
    import merfin_amd as m
    k = 21
    keys, vals = database(k, 3000, 99)
    d = str(tmp_path / "db.meryl")
    meryl_layout.write_db(d, k, keys, vals)
    assert m.db_convert(d, str(tmp_path / "ok.mfxk")) == len(keys)
    files = sorted(f for f in os.listdir(d) if f.endswith(".merylData"))
    target = next(os.path.join(d, f) for f in files if os.path.getsize(os.path.join(d, f)) > 400)
    img = open(target, "rb").read()
    refused = accepted = 0
    for at in range(0, 200):
        for mask in (0x80, 0x01, 0xff):
            dmg = bytearray(img)
            dmg[at] ^= mask
            open(target, "wb").write(bytes(dmg))
            try:
                m.db_convert(d, str(tmp_path / "o.mfxk"))
                accepted += 1
            except m.MfxError:
                refused += 1
    for cut in (0, 1, 8, 63, 64, 65, 200, len(img) // 2, len(img) - 1):
        open(target, "wb").write(img[:cut])
        with pytest.raises(m.MfxError):
            m.db_convert(d, str(tmp_path / "o.mfxk"))
    open(target, "wb").write(img)
    assert m.db_convert(d, str(tmp_path / "ok2.mfxk")) == len(keys)
    assert refused > 300                                      # (most damage is caught: framing, magic, monotony, the index's statistics)
