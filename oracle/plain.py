"""plain.py -- the reference formulas restated a SECOND time, in plain Python: arbitrary-precision integers for the
k-mers (so any k up to 64 works unchanged), dict-based k-mer counts, string reverse complement, IEEE doubles.
TEST INFRASTRUCTURE ONLY (same rules as merfin_oracle.h: only tests/, smoke() and bench.py's cpu_baseline may use
anything under oracle/).  Written independently of merfin_oracle.c and checked against it bit for bit at k <= 31
(tests/test_oracle_vs_numpy.py); it is the oracle of the 32 <= k <= 64 path, where the C oracle's 64-bit k-mers end.
Pure-Python loops: small cases only.  Citations are relative to /root/reference."""
import math

COMP = {"A": "T", "C": "G", "G": "C", "T": "A"}
CODE = {"A": 0, "C": 1, "T": 2, "G": 3}          # meryl kmerTiny encoding


def enc(s):
    v = 0
    for ch in s:
        v = (v << 2) | CODE[ch]
    return v


def dec(v, k):
    return "".join("ACTG"[(v >> (2 * (k - 1 - i))) & 3] for i in range(k))


def revcomp(s):
    return "".join(COMP[c] for c in reversed(s))


def c_round(x):          # C round(): half away from zero (Python's round() is half-to-even)
    return math.floor(x + 0.5) if x >= 0 else -math.floor(-x + 0.5)


def valid_kmers(contig, k):
    """(start position, forward k-mer string) of every valid k-mer: all k bases ACGT, either case (kmerIterator)"""
    s = contig.upper()
    for i in range(len(s) - k + 1):
        w = s[i:i + k]
        if all(c in "ACGT" for c in w):
            yield i, w


def count_kmers(k, contigs):
    """`meryl count`: canonical k-mer -> occurrences (merfin-globals.C:182-186)"""
    d = {}
    for c in contigs:
        for _, w in valid_kmers(c, k):
            x = min(enc(w), enc(revcomp(w)))
            d[x] = d.get(x, 0) + 1
    return d


def getK(peak, probK, probP, readV, asmV):           # merfin-globals.C:66-98
    readK, prob = 0.0, 1.0
    if readV == 0:
        readK = 0.0
    elif readV < peak:
        readK = 1.0
    else:
        readK = float(c_round(readV / peak))
    if 0 < readV <= len(probK):
        readK, prob = float(probK[readV - 1]), float(probP[readV - 1])
    return readK, float(asmV), prob


def kmetric(readK, asmK):                            # merfin-globals.H:248-261
    if readK == 0:
        return 0.0
    if asmK > readK:
        return (asmK / readK - 1) * -1
    if asmK < readK:
        return readK / asmK - 1
    return 0.0


def values(k, w, R, A, minV=0, maxV=2**64 - 1):
    """getK(kmer,kmer): value(fmer) + value(rmer) in uint32 arithmetic (merfin-globals.C:107-108); R holds only the
    read k-mers with minV <= count <= maxV (merylExactLookup::load, merfin.C:199-200)"""
    f, r = enc(w), enc(revcomp(w))
    rd = lambda x: R.get(x, 0) if minV <= R.get(x, 0) <= maxV else 0
    return (rd(f) + rd(r)) & 0xffffffff, (A.get(f, 0) + A.get(r, 0)) & 0xffffffff


def py_hist(k, peak, probK, probP, contig, R, A, minV=0, maxV=2**64 - 1):    # merfin-histogram.C:54-91
    undr, over = {}, {}
    kasm = kmissing = 0
    kover = 0.0
    for _, w in valid_kmers(contig, k):
        kasm += 1
        readV, asmV = values(k, w, R, A, minV, maxV)
        readK, asmK, prob = getK(peak, probK, probP, readV, asmV)
        if readK == 0:
            kmissing += 1
            continue
        if asmK > readK:
            idx = int(((asmK / readK - 1) + 0.1) / 0.2)
            undr[idx] = undr.get(idx, 0) + 1
            kover += (1.0 - readK / asmK) * prob
        else:
            idx = int(((readK / asmK - 1) + 0.1) / 0.2)
            over[idx] = over.get(idx, 0) + 1
    return undr, over, kasm, kmissing, kover


def py_dump(k, peak, probK, probP, contig, R, A):    # merfin-dump.C:44-67: per start position (readV, asmV, readK, asmK, K*)
    out = {}
    for i, w in valid_kmers(contig, k):
        readV, asmV = values(k, w, R, A)
        readK, asmK, _ = getK(peak, probK, probP, readV, asmV)
        out[i] = (readV, asmV, readK, asmK, kmetric(readK, asmK))
    return out


def py_completeness(k, peak, probK, probP, R, A):    # merfin-completeness.C:70-123: 64 per-piece sums, then the totals
    tot, und = [0.0] * 64, [0.0] * 64
    for x, rv in R.items():
        if rv == 0:
            continue
        readK, asmK, _ = getK(peak, probK, probP, rv, A.get(x, 0))
        piece = (x >> (2 * k - 6)) & 63 if 2 * k >= 6 else 0
        tot[piece] += readK
        if readK > asmK:
            und[piece] += readK - asmK
    return tot, und
