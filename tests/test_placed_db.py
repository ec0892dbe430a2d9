"""The PLACED form of a k-mer database (csrc/mfx_place.h, mfx_db.cpp FLAT_PLACED): records sorted by where the compact table of a
sequence-only index puts them.  CPU side: the placement number P of a k-mer against an independent plain-Python model of the
arithmetic, its inverse, the order property (ascending P = ascending line for every table size), and the file form through
mfx_db_convert_placed / mfx_db_write_flat_placed / mfx_db_convert back.  (The device side: tests/test_gpu_placed_db.py.)"""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
M32 = 0xffffffff


def _m():
    import merfin_amd as m
    m.load_library()
    return m


# ---- plain-Python model (written from the description in mfx_place.h, not from its code) ----
def revcomp(x, k):
    r = 0
    for _ in range(k):
        r = (r << 2) | ((x & 3) ^ 2)
        x >>= 2
    return r


def tmer_order(c):
    return (((c * 0x9E3779B1) & M32) >> 7) & 511


def mix(lo, hi):
    u = (lo * 0x9E3779B1) & M32
    u ^= u >> 15
    u = (u * 0x85EBCA77) & M32
    u ^= u >> 13
    return u ^ ((hi * 0xC2B2AE3D) & M32)


def model_encode(k, key):
    """P of a canonical k-mer: its smallest t-mer (ties: leftmost) samples one of four windows of k - 3 bases"""
    t = ((k + 1) & 3) + 4
    m = k - 3
    best, x = None, 0
    for p in range(k - t + 1):
        a = (key >> (2 * (k - t - p))) & ((1 << (2 * t)) - 1)
        o = tmer_order(min(a, revcomp(a, t)))
        if best is None or o < best:
            best, x = o, p
    j = x % 4
    a = (key >> (2 * (3 - j))) & ((1 << (2 * m)) - 1)
    b = revcomp(a, m)
    c, s = (a, 0) if a <= b else (b, 1)
    left = key >> (2 * (m + 3 - j))                       # the j bases left of the window
    right = key & ((1 << (2 * (3 - j))) - 1)              # the 3 - j bases right of it
    e = (left << (2 * (3 - j))) | right
    R = max(0, 2 * m - 32)
    hi, lo = c >> 32, c & M32
    top = mix(lo, hi)
    return (top << (R + 9)) | (hi << 9) | s | (j << 1) | (e << 3), top


def canon(r, k, n):
    x = r.integers(0, 1 << (2 * k), size=n, dtype=np.uint64)
    out = np.array([min(int(v), revcomp(int(v), k)) for v in x], dtype=np.uint64)
    return np.unique(out)


@pytest.mark.parametrize("k", [13, 14, 16, 19, 20, 21, 22, 25, 27, 30])
def test_placement_number_matches_the_model_and_inverts(k, tmp_path):
    m = _m()
    r = np.random.default_rng(900 + k)
    km = canon(r, k, 3000)
    # low-complexity k-mers: every t-mer equal (all ties), both strands
    extra = [0, int("1" * 0, 2) if False else 0]
    for unit in ("A", "AC", "ACG", "TTG"):
        s = (unit * 40)[:k]
        v = 0
        for ch in s:
            v = (v << 2) | {"A": 0, "C": 1, "T": 2, "G": 3}[ch]
        extra.append(min(v, revcomp(v, k)))
    km = np.unique(np.concatenate([km, np.array(extra, dtype=np.uint64)]))
    P = m.db_place_keys(k, km)
    assert len(np.unique(P)) == len(km)                       # one to one
    assert int(P.max()) >> max(2 * k + 3, 41) == 0
    for key, p in list(zip(km.tolist(), P.tolist()))[:400] + list(zip(km.tolist(), P.tolist()))[-8:]:
        want, _top = model_encode(k, key)
        assert p == want, (k, key)
    # ascending P = ascending line, whatever the table's size
    o = np.argsort(P)
    tops = np.array([model_encode(k, int(x))[1] for x in km[o][:500]], dtype=np.uint64)
    for nlines in (1024, 12345, (1 << 31) + 7):
        lines = (tops * np.uint64(nlines)) >> np.uint64(32)
        assert (np.diff(lines.astype(np.int64)) >= 0).all()
    # a non-canonical k-mer is placed as its canonical form
    rc = np.array([revcomp(int(x), k) for x in km[:50]], dtype=np.uint64)
    np.testing.assert_array_equal(m.db_place_keys(k, rc), P[:50])
    # the file form and back: placed -> sorted flat -> the same k-mers and counts
    vals = (1 + (km % np.uint64(5000))).astype(np.uint32)
    vals[::97] = 2 ** 31 + 5                                   # escapes (beyond any block's count field)
    placed = str(tmp_path / "p.mfxk")
    m.db_write_flat_placed(placed, k, P[o], vals[o])
    info = m.db_probe(placed)
    assert info.get("placed") and info["k"] == k and info["n_kmers"] == len(km)
    back = str(tmp_path / "back.mfxk")
    assert m.db_convert(placed, back) == len(km)
    assert not m.db_probe(back).get("placed")
    flat = str(tmp_path / "flat.mfxk")
    m.db_write_flat(flat, k, km, vals)
    assert open(back, "rb").read() == open(flat, "rb").read()
    # ... and the converter makes the same placed file from the k-mer-sorted one
    again = str(tmp_path / "again.mfxk")
    assert m.db_convert_placed(flat, again) == len(km)
    assert open(again, "rb").read() == open(placed, "rb").read()


def test_placed_form_refusals(tmp_path):
    m = _m()
    r = np.random.default_rng(5)
    for k in (12, 31):
        km = canon(r, k, 100)
        flat = str(tmp_path / ("f%d" % k))
        m.db_write_flat(flat, k, km, np.ones(len(km), dtype=np.uint32))
        if k == 12:
            with pytest.raises(m.MfxError, match="placed database holds"):
                m.db_convert_placed(flat, str(tmp_path / "x"))
        with pytest.raises(m.MfxError):
            m.db_place_keys(k, km)                             # (k = 31: P takes 65 bits -- the converter alone makes that file, below)
        with pytest.raises(m.MfxError):
            m.db_write_flat_placed(str(tmp_path / "w"), k, np.arange(1, 50, dtype=np.uint64), np.ones(49, dtype=np.uint32))
    k = 21
    km = canon(r, k, 200)
    nc = np.array([revcomp(int(x), k) for x in km if revcomp(int(x), k) != int(x)], dtype=np.uint64)
    flat = str(tmp_path / "nc")
    m.db_write_flat(flat, k, np.sort(nc), np.ones(len(nc), dtype=np.uint32))
    with pytest.raises(m.MfxError, match="not canonical"):
        m.db_convert_placed(flat, str(tmp_path / "x"))
    P = np.sort(m.db_place_keys(k, km))
    with pytest.raises(m.MfxError, match="not strictly ascending"):
        m.db_write_flat_placed(str(tmp_path / "y"), k, P[::-1].copy(), np.ones(len(P), dtype=np.uint32))
    # a placed file of another placement version is refused by every reader
    good = str(tmp_path / "good")
    m.db_write_flat_placed(good, k, P, np.ones(len(P), dtype=np.uint32))
    raw = bytearray(open(good, "rb").read())
    raw[13] ^= 0x02                                            # flags bits 8-15: the version
    bad = str(tmp_path / "bad")
    open(bad, "wb").write(bytes(raw))
    with pytest.raises(m.MfxError, match="another version"):
        m.db_convert(bad, str(tmp_path / "z"))


def test_placed_31mers_carry_the_strand_bit_in_the_count_field(tmp_path):
    """k = 31: P takes 65 bits; the file holds P >> 1 and bit 0 of a record's count field is the strand bit (mfx_place.h, mfx_p_encode_s).
    Two k-mers that differ in that bit alone share the stored number; an ESCAPED record's field is all ones, so its strand is what the
    pair or the escape list says.  placed -> sorted must give back the k-mers and counts, every combination of escaped / plain twins."""
    m = _m()
    k, mm = 31, 28
    r = np.random.default_rng(931)
    km = set(canon(r, k, 4000).tolist())
    twins = []
    while len(twins) < 8:
        j = int(r.integers(0, 4))
        c = int(r.integers(0, 1 << 56))
        if revcomp(c, mm) == c:
            continue
        e = int(r.integers(0, 64))
        left, right = e >> (2 * (3 - j)), e & ((1 << (2 * (3 - j))) - 1)
        a = (left << (2 * (mm + 3 - j))) | (c << (2 * (3 - j))) | right
        b = (left << (2 * (mm + 3 - j))) | (revcomp(c, mm) << (2 * (3 - j))) | right
        if a > revcomp(a, k) or b > revcomp(b, k):
            continue                                           # both must be canonical k-mers
        pa, pb = model_encode(k, a)[0], model_encode(k, b)[0]
        if pa >> 1 == pb >> 1 and pa != pb:
            twins.append((a, b) if pa < pb else (b, a))
    for a, b in twins:
        km.update((a, b))
    km = np.array(sorted(km), dtype=np.uint64)
    vals = (1 + (km % np.uint64(3000))).astype(np.uint32)
    vals[::89] = 2 ** 31 + 9                                   # escapes, beyond 31 bits
    vals[5::97] = (1 << 21) + 3                                # fits a 22-bit field as a count, not with the strand bit beside it
    big = np.uint32(3000000011)
    combos = [(7, 9), (big, 9), (7, big), (big, big), (2 ** 21 - 1, 2 ** 21), (1, 1), (big, 1), (1, big)]
    pos = {int(x): i for i, x in enumerate(km.tolist())}
    for (a, b), (va, vb) in zip(twins, combos):
        vals[pos[a]], vals[pos[b]] = va, vb
    flat, placed, back = (str(tmp_path / n) for n in ("flat.mfxk", "placed.mfxk", "back.mfxk"))
    m.db_write_flat(flat, k, km, vals)
    assert m.db_convert_placed(flat, placed) == len(km)
    info = m.db_probe(placed)
    assert info.get("placed") and info["k"] == k and info["n_kmers"] == len(km)
    assert m.db_convert(placed, back) == len(km)
    assert open(back, "rb").read() == open(flat, "rb").read()


def test_cli_convert_placed(tmp_path):
    m = _m()
    exe = os.path.join(ROOT, "merfin_amd", "bin", "merfin")
    k = 21
    km = canon(np.random.default_rng(11), k, 5000)
    vals = (1 + (km % np.uint64(40))).astype(np.uint32)
    flat = str(tmp_path / "in.mfxk")
    m.db_write_flat(flat, k, km, vals)
    out = str(tmp_path / "out.mfxk")
    r = subprocess.run([exe, "-convert", flat, "-placed", "-output", out], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert m.db_probe(out).get("placed")
    back = str(tmp_path / "back.mfxk")
    m.db_convert(out, back)
    assert open(back, "rb").read() == open(flat, "rb").read()
