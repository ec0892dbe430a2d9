"""Writer of a meryl-database-shaped directory following the layout RECALLED in
SURVEY.md Appendix C (stuffedBits containers, merylIndex + 64 x 0xBBBBBB.merylData,
unary/binary coded suffixes, 32-bit values).  It exists to exercise the decoder
in merfin_amd/csrc/mfx_db.cpp.  It is NOT evidence that the decoder reads real
meryl output: no meryl source or database exists in this environment, so both
sides rest on the same recollection ("parity unpinned")."""
import os
import struct

import numpy as np

IDX_MAGIC1 = 0x646e496c7972656d
DAT_MAGIC1 = 0x7461446c7972656d
DAT_MAGIC2 = 0x0a3030656c694661


def idx_magic2(version):
    return 0x30302e765f5f7865 | (version << 56)


class BitWriter:
    def __init__(self):
        self.bits = []

    def put(self, v, n):
        for i in range(n - 1, -1, -1):
            self.bits.append((v >> i) & 1)

    def unary(self, z):
        self.bits.extend([0] * z)
        self.bits.append(1)

    def image(self):
        """one stuffedBits file image holding a single block"""
        nbits = len(self.bits)
        nw = (nbits + 63) // 64
        b = self.bits + [0] * (nw * 64 - nbits)
        words = []
        for w in range(nw):
            x = 0
            for bit in b[w * 64:(w + 1) * 64]:
                x = (x << 1) | bit
            words.append(x)
        out = struct.pack("<QII", max(nbits, 64), 1, 1)       # dataBlockLenMax, dataBlocksLen, dataBlocksMax
        out += struct.pack("<Q", 0) + struct.pack("<Q", nbits)  # bgn[0], len[0]
        out += b"".join(struct.pack("<Q", w) for w in words)
        return out


def to_ints(kmers):
    """k-mers as Python integers: a uint64 array (k <= 32) or rows [low 64 bits, high bits] (k > 32)"""
    a = np.asarray(kmers)
    if a.ndim == 2:
        return [int(lo) | (int(hi) << 64) for lo, hi in a.tolist()]
    return [int(x) for x in a.tolist()]


def write_db(path, k, kmers, values, prefix_bits=12, unary_bits=None, version=3, stats=True, stats_override=None):
    """kmers: sorted unique uint64 array, values: uint32.  stats: append the statistics block (numUnique,
    numDistinct, numTotal as 64-bit fields) to the master index; stats_override = (unique, distinct, total) writes
    other numbers there (corrupt-field tests)."""
    os.makedirs(path, exist_ok=True)
    suffix_bits = 2 * k - prefix_bits
    blocks_bits = prefix_bits - 6
    if unary_bits is None:
        unary_bits = max(1, min(suffix_bits - 1, 8))
    binary_bits = suffix_bits - unary_bits
    w = BitWriter()
    w.put(IDX_MAGIC1, 64)
    w.put(idx_magic2(version), 64)
    for v in (prefix_bits, suffix_bits, 6, blocks_bits):
        w.put(v, 32)
    if version >= 2:
        w.put(0, 32)
    values = np.asarray(values, dtype=np.uint32)
    ks = to_ints(kmers)
    assert all(a < b for a, b in zip(ks, ks[1:])), "k-mers must be sorted and unique"
    if stats:
        st = stats_override or (int((values == 1).sum()), len(ks), int(values.astype(np.uint64).sum()))
        for v in st:
            w.put(int(v), 64)
    open(os.path.join(path, "merylIndex"), "wb").write(w.image())
    vals_all = values.tolist()
    per_file = {}
    for km, v in zip(ks, vals_all):                            # k-mer = [prefix : prefix_bits][suffix : suffix_bits]
        pfx = km >> suffix_bits
        per_file.setdefault(pfx >> blocks_bits, {}).setdefault(pfx, []).append((km & ((1 << suffix_bits) - 1), v))
    for fl in range(64):
        name = "0x" + format(fl, "06b") + ".merylData"
        if fl not in per_file:
            open(os.path.join(path, name), "wb").close()       # all 64 data files exist, empty pieces included
            continue
        with open(os.path.join(path, name), "wb") as f:
            for pfx in sorted(per_file[fl]):
                sfx = [x[0] for x in per_file[fl][pfx]]
                vals = [x[1] for x in per_file[fl][pfx]]
                b = BitWriter()
                b.put(DAT_MAGIC1, 64)
                b.put(DAT_MAGIC2, 64)
                b.put(pfx, 64)
                b.put(len(sfx), 64)
                b.put(1, 8)
                b.put(unary_bits, 32)
                b.put(binary_bits, 32)
                b.put(0, 64)
                b.put(1, 8)
                b.put(0, 64)
                b.put(0, 64)
                hi_prev = 0
                for s_ in sfx:
                    hi = s_ >> binary_bits
                    b.unary(hi - hi_prev)
                    hi_prev = hi
                    b.put(s_ & ((1 << binary_bits) - 1), binary_bits)       # wider than 64 bits for large k: one bit stream
                for v in vals:
                    b.put(v, 32)
                f.write(b.image())
