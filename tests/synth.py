"""Seeded synthetic assemblies + k-mer count sets for the parity tests
(SURVEY.md section 8d recipe, scaled down).  Pure numpy; no reference code."""
import numpy as np

from oracle import pyoracle as po

SEED = 20260928
BASES = np.frombuffer(b"ACGT", dtype=np.uint8)


def rng(seed=SEED):
    return np.random.default_rng(seed)


def random_contig(r, n):
    return BASES[r.integers(0, 4, size=n)].copy()


def make_truth(r, sizes, repeat_frac=(0.03, 0.01, 0.005), tandem=None):
    """Contigs of the given sizes with 2-/3-/10-copy dispersed repeats, N runs
    and lower-case stretches injected."""
    contigs = []
    for n in sizes:
        c = random_contig(r, n)
        if n >= 2000:
            for copies, frac in zip((2, 3, 10), repeat_frac):
                unit = max(50, int(n * frac / copies))
                src = r.integers(0, n - unit)
                u = c[src:src + unit].copy()
                for _ in range(copies - 1):
                    d = r.integers(0, n - unit)
                    c[d:d + unit] = u
            if tandem:
                ulen, ncopy = tandem
                if n > ulen * ncopy + 10:
                    d = r.integers(0, n - ulen * ncopy)
                    c[d:d + ulen * ncopy] = np.tile(c[d:d + ulen], ncopy)
        contigs.append(c)
    return contigs


def decorate(r, contigs, n_run=200, n_single=5, lower_frac=0.005):
    out = []
    for c in contigs:
        c = c.copy()
        n = len(c)
        if n > 4 * n_run:
            s = r.integers(0, n - n_run)
            c[s:s + n_run] = ord("N")
        for _ in range(n_single if n > 100 else 0):
            c[r.integers(0, n)] = ord("N")
        m = r.random(n) < lower_frac
        c[m] = c[m] | 0x20                      # lower-case (N -> n too, still invalid)
        out.append(c)
    return out


def mutate(r, contigs, sub_rate=1e-3):
    """assembly = truth + substitution errors"""
    out = []
    for c in contigs:
        c = c.copy()
        m = np.nonzero(r.random(len(c)) < sub_rate)[0]
        for i in m:
            if c[i] in b"ACGT":
                c[i] = BASES[(np.searchsorted(BASES, c[i]) + r.integers(1, 4)) % 4]
        out.append(c)
    return out


def as_bytes(contigs):
    return [c.tobytes() for c in contigs]


def read_counts(r, k, truth_bytes, peak, err_kmers=0):
    """read index: Poisson(peak * copies) per distinct canonical truth k-mer
    (0 -> absent) plus `err_kmers` random low-count error k-mers."""
    tk, tc = po.count_kmers(k, truth_bytes)
    v = r.poisson(peak * tc.astype(np.float64)).astype(np.uint32)
    keep = v > 0
    tk, v = tk[keep], v[keep]
    if err_kmers:
        e = r.integers(0, 4 ** k, size=err_kmers, dtype=np.uint64)
        e = np.unique(np.array([po.lib().orc_canonical(int(x), k) for x in e], dtype=np.uint64))
        e = e[~np.isin(e, tk)]
        ev = r.choice(np.array([1, 2, 3], dtype=np.uint32), size=len(e), p=[0.8, 0.15, 0.05])
        tk = np.concatenate([tk, e])
        v = np.concatenate([v, ev])
        o = np.argsort(tk)
        tk, v = tk[o], v[o]
    return tk, v


def world(k=21, sizes=(30000, 9000, 4096, 4097, 500, 20, 0, 8191), peak=17.3, seed=SEED, err_kmers=2000,
          tandem=(37, 60)):
    """A complete small test world: assembly contigs (bytes), read (kmers, values), asm (kmers, values)."""
    r = rng(seed)
    truth = make_truth(r, sizes, tandem=tandem)
    asm = decorate(r, mutate(r, truth))
    tb, ab = as_bytes(truth), as_bytes(asm)
    rk, rv = read_counts(r, k, tb, peak, err_kmers)
    ak, av = po.count_kmers(k, ab)
    return ab, (rk, rv), (ak, av)


# ---------------------------------------------------------------------------
# variant worlds: truth genome, an assembly carrying errors, and a VCF that
# proposes corrections (true ones, decoys and malformed/edge-case records)
# ---------------------------------------------------------------------------
def variant_world(k=21, sizes=(12000, 5000, 300), peak=17.3, seed=SEED, burst=0.08, base_rate=0.002, decoys=60, tables=True):
    """tables=False: (names, asm, vcf, truth contigs) -- the caller counts the k-mers itself (k > 31: oracle/plain.py)"""
    r = rng(seed)
    truth = [random_contig(r, n) for n in sizes]
    names = ["ctg%d" % i for i in range(len(sizes))]
    asm, lines = [], []
    for ci, t in enumerate(truth):
        t = t.tobytes()
        a = bytearray()
        i = 0
        in_burst = 0
        while i < len(t):
            if in_burst == 0 and r.random() < 0.004:
                in_burst = int(r.integers(20, 90))
            rate = burst if in_burst else base_rate
            in_burst = max(0, in_burst - 1)
            if len(a) > 2 and i + 4 < len(t) and r.random() < rate:
                kind = r.integers(0, 3)
                gt = "1/1" if r.random() < 0.8 else ("1|1" if r.random() < 0.5 else "0/1")
                q = "%.1f" % (r.random() * 60) if r.random() < 0.9 else "."
                if kind == 0:                       # substitution error in the assembly
                    wrong = bytes([BASES[(list(b"ACGT").index(t[i]) + int(r.integers(1, 4))) % 4]])
                    lines.append((names[ci], len(a) + 1, wrong.decode(), chr(t[i]), q, gt))
                    a += wrong
                    i += 1
                elif kind == 1:                     # assembly lacks 1-3 truth bases -> insertion fixes it
                    m = int(r.integers(1, 4))
                    anchor = chr(a[-1])
                    lines.append((names[ci], len(a), anchor, anchor + t[i:i + m].decode(), q, gt))
                    i += m
                else:                               # assembly has 1-3 extra bases -> deletion fixes it
                    m = int(r.integers(1, 4))
                    extra = random_contig(r, m).tobytes()
                    anchor = chr(a[-1])
                    lines.append((names[ci], len(a), anchor + extra.decode(), anchor, q, gt))
                    a += extra
            else:
                a.append(t[i])
                i += 1
        asm.append(bytes(a))
    # decoys and special records
    extra = []
    for _ in range(decoys):
        ci = int(r.integers(0, len(asm)))
        a = asm[ci]
        if len(a) < 50:
            continue
        p = int(r.integers(0, len(a) - 5))
        ref = chr(a[p])
        alts = [b for b in "ACGT" if b != ref]
        roll = r.random()
        if roll < 0.35:
            extra.append((names[ci], p + 1, ref, alts[0], "30.0", "1/1"))
        elif roll < 0.5:
            extra.append((names[ci], p + 1, ref, alts[0] + "," + alts[1], "12.5", "1/2"))
        elif roll < 0.6:
            extra.append((names[ci], p + 1, ref, alts[0], "9", "0/0"))
        elif roll < 0.7:
            extra.append((names[ci], p + 1, ref, alts[0], "9", "./."))
        elif roll < 0.8:                            # multi-base REF spanning the following variants
            L = int(r.integers(3, 9))
            extra.append((names[ci], p + 1, a[p:p + L].decode(), ref, "22.2", "1/1"))
        elif roll < 0.9:                            # ALT equal to REF / GT index past the ALT list
            extra.append((names[ci], p + 1, ref, ref + "," + alts[2], "5", "1/2" if r.random() < 0.5 else "3/1"))
        else:
            extra.append((names[ci], p + 1, ref, alts[1], "7.77", "1/1"))
            extra.append((names[ci], p + 1, ref, alts[2], "8.88", "1/1"))          # same POS twice (sort ties)
    # a dense pile exceeding -comb
    if len(asm[0]) > 3000:
        for j in range(22):
            p = 2000 + 3 * j
            ref = chr(asm[0][p])
            extra.append((names[0], p + 1, ref, [b for b in "ACGT" if b != ref][j % 3], "40", "1/1"))
    # near the contig edges (inside the first / last k-1 bases)
    for ci in (0, 1):
        a = asm[ci]
        for p in (0, 3, len(a) - 2):
            ref = chr(a[p])
            extra.append((names[ci], p + 1, ref, [b for b in "ACGT" if b != ref][0], "33", "1/1"))
    allv = lines + extra
    order = r.permutation(len(allv))
    # keep it mostly position-sorted but not entirely (the reference sorts per chromosome anyway)
    allv = [allv[i] for i in sorted(range(len(allv)), key=lambda i: (allv[i][0], allv[i][1] // 7, order[i]))]
    vcf = ["##fileformat=VCFv4.2"] + ["##contig=<ID=%s,length=%d>" % (n, len(a)) for n, a in zip(names, asm)]
    vcf += ["##FORMAT=<ID=GT,Number=1,Type=String,Description=\"Genotype\">", "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tSAMPLE"]
    for chrom, pos, ref, alt, q, gt in allv:
        vcf.append("%s\t%d\t.\t%s\t%s\t%s\tPASS\t.\tGT:DP\t%s:%d" % (chrom, pos, ref, alt, q, gt, int(r.integers(3, 60))))
    vcf.insert(8, "ctg0\t100\t.\tA\tC\t3\tq40\t.\tGT")           # only 9 columns -> excluded
    vcf.insert(12, "ghost\t10\t.\tA\tC\t3\tPASS\t.\tGT\t1/1")      # chromosome absent from the FASTA
    tb = [t.tobytes() for t in truth]
    if not tables:
        return names, asm, "\n".join(vcf) + "\n", tb
    rk, rv = read_counts(r, k, tb, peak, err_kmers=500)
    ak, av = po.count_kmers(k, asm)
    return names, asm, "\n".join(vcf) + "\n", (rk, rv), (ak, av)
