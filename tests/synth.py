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
