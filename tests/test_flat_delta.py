"""The delta-coded flat database form (mfx_db.cpp FLAT_DELTA) as a FILE FORMAT: what mfx_db_write_flat writes for sorted
k-mers is decoded here by an independent reader (numpy / Python ints, from the format's description in the header of
mfx_db.cpp) and must give back the arrays.  The GPU side of the format -- the kernel that decodes and inserts the blocks --
is pinned by tests/test_gpu_db.py against the other forms of the same database."""
import struct

import numpy as np
import pytest

BLOCK = 4096


def read_flat(path):
    raw = open(path, "rb").read()
    magic, k, flags, n, n_esc = struct.unpack_from("<8sIIQQ", raw, 0)
    assert magic == b"MFXKMER1"
    return raw, k, flags, n, n_esc


def bits_at(words, bit, nbits):
    i, sh = bit >> 6, bit & 63
    x = int(words[i]) >> sh
    if sh + nbits > 64:
        x |= int(words[i + 1]) << (64 - sh)
    return x & ((1 << nbits) - 1)


def decode_delta(path):
    raw, k, flags, n, n_esc = read_flat(path)
    assert flags & 4
    (nblocks,) = struct.unpack_from("<Q", raw, 32)
    assert nblocks == (n + BLOCK - 1) // BLOCK
    d = np.frombuffer(raw, dtype="<u8", count=2 * (nblocks + 1), offset=40)
    keys, vals = [], []
    for b in range(nblocks):
        first, info = int(d[2 * b]), int(d[2 * b + 1])
        off, kb, vb = info & ((1 << 48) - 1), (info >> 48) & 0xff, (info >> 56) & 0xff
        nxt = int(d[2 * b + 3]) & ((1 << 48) - 1)
        cnt = min(BLOCK, n - b * BLOCK)
        assert off % 8 == 0 and nxt - off == (((cnt - 1) * kb + 63) // 64 + (cnt * vb + 63) // 64) * 8
        w = np.frombuffer(raw, dtype="<u8", count=(nxt - off) // 8, offset=off)
        vw = w[((cnt - 1) * kb + 63) // 64:]
        cur = first
        for e in range(cnt):
            if e:
                cur += bits_at(w, (e - 1) * kb, kb) if kb else 0
            keys.append(cur)
            v = bits_at(vw, e * vb, vb)
            vals.append(None if v == (1 << vb) - 1 else v)
    end = int(d[2 * nblocks + 1]) & ((1 << 48) - 1)
    assert len(raw) == end + 12 * n_esc
    ek = np.frombuffer(raw, dtype="<u8", count=n_esc, offset=end).tolist()
    ev = np.frombuffer(raw, dtype="<u4", count=n_esc, offset=end + 8 * n_esc).tolist()
    esc = dict(zip(ek, ev))
    assert len(esc) == n_esc and sum(v is None for v in vals) == n_esc
    return k, keys, [esc[kk] if v is None else v for kk, v in zip(keys, vals)]


@pytest.mark.parametrize("k,n,seed", [(21, 1, 1), (21, 4096, 2), (21, 4097, 3), (15, 20000, 4), (31, 3 * 4096 + 77, 5), (4, 200, 6)])
def test_written_blocks_decode_to_the_arrays(tmp_path, k, n, seed):
    import merfin_amd as m
    rng = np.random.default_rng(seed)
    space = 1 << (2 * k)
    keys = np.unique(rng.integers(0, space, size=min(n * 2, space), dtype=np.uint64))[:n]
    n = len(keys)
    vals = rng.integers(0, 70, size=n).astype(np.uint32)
    vals[::97] = rng.integers(2**20, 2**32 - 1, size=len(vals[::97]), dtype=np.uint64).astype(np.uint32)
    if n > 4096:
        vals[4096:4200] = 2**32 - 1                                  # a stretch of the largest count
        keys[100:200] = keys[100] + np.arange(100, dtype=np.uint64)  # consecutive k-mers
        keys = np.unique(keys)
        n = len(keys)
        vals = vals[:n]
    path = str(tmp_path / "d.mfxk")
    m.db_write_flat(path, k, keys, vals)
    kk, dk, dv = decode_delta(path)
    assert kk == k and dk == keys.tolist() and dv == vals.tolist()
    assert m.db_probe(path) == {"k": k, "format": "flat", "n_kmers": n}


def test_unsorted_k_mers_are_written_in_another_form(tmp_path, monkeypatch):
    """the delta form needs strictly ascending k-mers; anything else is written as packed (k <= 21) or plain records, and
    MFX_FLAT_DELTA=0 switches the form off"""
    import merfin_amd as m
    keys = np.array([5, 3, 9, 9], dtype=np.uint64)
    vals = np.array([1, 2, 3, 4], dtype=np.uint32)
    p = str(tmp_path / "u.mfxk")
    m.db_write_flat(p, 21, keys, vals)
    assert read_flat(p)[2] & 6 == 2
    m.db_write_flat(p, 31, keys, vals)
    assert read_flat(p)[2] & 6 == 0
    s = np.sort(np.unique(keys))
    m.db_write_flat(p, 21, s, vals[:len(s)])
    assert read_flat(p)[2] & 6 == 4
    monkeypatch.setenv("MFX_FLAT_DELTA", "0")
    m.db_write_flat(p, 21, s, vals[:len(s)])
    assert read_flat(p)[2] & 6 == 2
