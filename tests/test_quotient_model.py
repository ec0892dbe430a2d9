"""The arithmetic of the QUOTIENT form of the compact index layout (22 <= k <= 31; csrc/mfx_kernels.hip: mfx_q_mix / mfx_q_unmix,
mfx_q_parts, mfx_q_place, mfx_q_invert), restated in plain Python integers: the minimizer hash is a bijection, (home line, key
field) -> k-mer inverts k-mer -> (home line, key field) for every table size the host may choose, the key field fits its 40
bits, and two different k-mers never share (line, key field).  No device involved; the device code is held against the oracle
by tests/test_gpu_seqonly.py."""
import random

import pytest

M = 0xFFFFFFFF


def mix(lo, hi):
    u = (lo * 0x9E3779B1) & M
    u ^= u >> 15
    u = (u * 0x85EBCA77) & M
    u ^= u >> 13
    return u ^ ((hi * 0xC2B2AE3D) & M)


def unmix(top, hi):
    u = top ^ ((hi * 0xC2B2AE3D) & M)
    u ^= u >> 13
    u ^= u >> 26
    u = (u * 0xB6C92F47) & M
    u ^= u >> 15
    u ^= u >> 30
    return (u * 0x0E8B2F51) & M


def revcomp(x, k):
    r = 0
    for _ in range(k):
        r = (r << 2) | ((x & 3) ^ 2)
        x >>= 2
    return r


def tmer_order(c):
    return (((c * 0x9E3779B1) & M) >> 7) & 511


def parts(key, k, t):
    rc = revcomp(key, k)
    best, x, tm = 1 << 32, 0, (1 << (2 * t)) - 1
    for p in range(k - t + 1):
        a, b = (key >> (2 * (k - t - p))) & tm, (rc >> (2 * p)) & tm
        o = tmer_order(min(a, b))
        if o < best:
            best, x = o, p
    m, j = k - 3, x & 3
    mm = (1 << (2 * m)) - 1
    a, b = (key >> (2 * (3 - j))) & mm, (rc >> (2 * j)) & mm
    e = ((key >> (2 * (m + 3 - j))) << (2 * (3 - j))) | (key & ((1 << (2 * (3 - j))) - 1))
    return min(a, b), (1 if b < a else 0), j, e


def place(k, nl, qshift, c, sbit, j, e):
    hi = c >> 32
    top = mix(c & M, hi)
    line, fq = (top * nl) >> 32, ((top * nl) & M) >> qshift
    R, Q = 2 * (k - 3) - 32, 32 - qshift
    return line, hi | (fq << R) | ((sbit | (j << 1) | (e << 3)) << (R + Q))


def invert(k, nl, qshift, home, f0):
    m = k - 3
    R, Q = 2 * m - 32, 32 - qshift
    hi, fq, meta = f0 & ((1 << R) - 1), (f0 >> R) & ((1 << Q) - 1), f0 >> (R + Q)
    sbit, j, e = meta & 1, (meta >> 1) & 3, (meta >> 3) & 63
    top = ((home << 32) + nl - 1) // nl
    first, want = (top * nl) & M, fq << qshift
    if want > first:
        top += (want - first + nl - 1) // nl
    assert (top * nl) >> 32 == home and ((top * nl) & M) >> qshift == fq
    c = (hi << 32) | unmix(top, hi)
    mmer = revcomp(c, m) if sbit else c
    left, right = e >> (2 * (3 - j)), e & ((1 << (2 * (3 - j))) - 1)
    return (left << (2 * (m + 3 - j))) | (mmer << (2 * (3 - j))) | right


def test_the_minimizer_hash_is_a_bijection():
    r = random.Random(1)
    for _ in range(20000):
        lo, hi = r.getrandbits(32), r.getrandbits(24)
        assert unmix(mix(lo, hi), hi) == lo


@pytest.mark.parametrize("k", list(range(22, 32)))
def test_quotient_round_trip_at_the_smallest_and_at_ordinary_table_sizes(k):
    r = random.Random(k)
    t = ((k + 1) & 3) + 4
    nmin = 1 << (2 * (k - 3) - 31)                                   # quot_min_lines (mfx_api.cpp)
    for nl in (max(nmin, 1024), max(nmin, 1024) + 12345, max(nmin, 845_000_000), (1 << 32) - 17):
        qshift = nl.bit_length() - 1
        seen = {}
        for _ in range(1500):
            f = r.getrandbits(2 * k)
            key = min(f, revcomp(f, k))
            line, f0 = place(k, nl, qshift, *parts(key, k, t))
            assert f0 < (1 << 40) and line < nl
            assert invert(k, nl, qshift, line, f0) == key
            assert seen.setdefault((line, f0), key) == key
