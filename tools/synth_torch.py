"""Device-side synthetic workload generator for bench.py and the full-size GPU
tests (SURVEY.md section 8d recipe).  Bench/test infrastructure: torch is used
here only to fabricate inputs in HBM; the evaluated path is the HIP library.

Everything is a pure function of (seed, position) through splitmix64, so any
rank / chunking reproduces the same genome, assembly and read counts.
"""
import math

import torch

SEED = 20260928
_M64 = (1 << 64) - 1


def _s64(x):
    x &= _M64
    return x - (1 << 64) if x >= (1 << 63) else x


def _lsr(x, s):
    return (x >> s) & ((1 << (64 - s)) - 1)


def splitmix64(x):
    """exact splitmix64 finaliser on an int64 tensor (wrapping arithmetic)"""
    z = x + _s64(0x9E3779B97F4A7C15)
    z = (z ^ _lsr(z, 30)) * _s64(0xBF58476D1CE4E5B9)
    z = (z ^ _lsr(z, 27)) * _s64(0x94D049BB133111EB)
    return z ^ _lsr(z, 31)


def _hash_range(seed, start, n, device):
    i = torch.arange(start, start + n, dtype=torch.int64, device=device)
    return splitmix64(i * _s64(0xD1342543DE82EF95) + seed)


_ACGT = None


def random_bases(n, seed, device, chunk=1 << 27):
    """n ASCII bases, i.i.d. uniform over ACGT"""
    global _ACGT
    if _ACGT is None or _ACGT.device != torch.device(device):
        _ACGT = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=device)
    out = torch.empty(n, dtype=torch.uint8, device=device)
    for o in range(0, n, chunk):
        m = min(chunk, n - o)
        h = _hash_range(seed, o, m, device)
        out[o:o + m] = _ACGT[_lsr(h, 61) & 3]
    return out


class _Rng:
    """tiny host-side splitmix64 stream for structural decisions (segment positions)"""

    def __init__(self, seed):
        self.s = seed & _M64

    def next(self):
        self.s = (self.s + 0x9E3779B97F4A7C15) & _M64
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _M64
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _M64
        return z ^ (z >> 31)

    def below(self, n):
        return self.next() % n


HUMAN_PROPORTIONS = [248, 242, 198, 190, 182, 171, 159, 145, 138, 134, 135, 133, 114, 107, 102, 90, 83, 80, 59, 64, 47, 51, 156, 57]


def contig_sizes(total, ncontigs=24):
    props = HUMAN_PROPORTIONS[:ncontigs] if ncontigs <= 24 else [1] * ncontigs
    s = sum(props)
    sizes = [total * p // s for p in props]
    sizes[0] += total - sum(sizes)
    return sizes


def make_truth(sizes, seed=SEED, device="cuda", unit=10000):
    """Truth genome: random contigs with dispersed 2-/3-/10-copy repeats (3 %, 1 %,
    0.5 % of bases) and 1000-copy tandem repeats (0.05 %)."""
    contigs = []
    layout = []          # (contig, [(copies, [positions], unit_len)])
    for ci, n in enumerate(sizes):
        c = random_bases(n, seed + 1000003 * (ci + 1), device)
        r = _Rng(seed * 31 + ci)
        reps = []
        if n >= 20 * unit:
            for copies, frac in ((2, 0.03), (3, 0.01), (10, 0.005)):
                nunits = max(1, int(n * frac / (copies * unit)))
                for _ in range(nunits):
                    pos = [r.below(n - unit) for _ in range(copies)]
                    src = c[pos[0]:pos[0] + unit].clone()
                    for p in pos[1:]:
                        c[p:p + unit] = src
                    reps.append((copies, pos, unit))
            # tandem: 1000 copies of a 37-mer unit, 0.05 % of bases
            tlen = 37 * 1000
            for _ in range(max(1, int(n * 0.0005 / tlen))):
                p = r.below(n - tlen)
                c[p:p + tlen] = c[p:p + 37].repeat(1000)
        contigs.append(c)
        layout.append(reps)
    return contigs, layout


# ---------------------------------------------------------------------------
# The `repeats` world: what a real (T2T-style human) assembly adds to the i.i.d. genome of SURVEY 8(d) -- repeat families whose
# k-mers have copy numbers in the hundreds to hundreds of thousands, i.e. read counts far beyond the 11-bit count fields of the
# compact index (readV >= 2047 <=> copy number >~ 79 at 26x) and assembly counts beyond them as well:
#   alu      : dispersed families, 300 bp consensus x 100 000 copies, every copy diverged from the consensus by its own 5-15 %
#   sat      : tandem arrays of a 171 bp monomer x 10 000 copies, the array's divergence one of 2 / 5 / 10 / 20 %
#   tandem5  : an EXACT 5-mer tandem array (>= 2 Mb: five distinct 21-mers, 400 000 copies each)
#   rdna     : a 45 kb unit x 400 copies (five arrays of 80), 0.1 % diverged
# `level` (percent of positions meant to carry a saturated read count: 1, 3, 10) picks how many of each; the copy numbers are
# scaled with the genome (x total / 3 Gb, never below `min_copies`) so that a small sample of the world still saturates.
# Everything is a pure function of (seed, level, sizes).
# ---------------------------------------------------------------------------
REPEAT_LEVELS = {
    #        alu families, satellite arrays, tandem5 arrays, rdna copies per array (x5 arrays)
    1:  dict(alu=1,  sat=4,   tandem5=1, rdna=80),
    3:  dict(alu=5,  sat=24,  tandem5=2, rdna=160),
    10: dict(alu=18, sat=100, tandem5=4, rdna=480),
}


def _mutated_copies(cons_codes, ncopies, div, seed, device):
    """[ncopies, L] uint8 ASCII: the consensus (codes 0..3) with i.i.d. substitutions at rate div[copy] (a tensor or a float)"""
    L = cons_codes.numel()
    h = _hash_range(seed, 0, ncopies * L, device).view(ncopies, L)
    u = _lsr(h, 11).double() * (1.0 / (1 << 53))
    d = div.view(-1, 1) if torch.is_tensor(div) else div
    hit = u < d
    code = cons_codes.view(1, L).expand(ncopies, L).long()
    code = torch.where(hit, (code + 1 + (_lsr(h, 3) % 3)) & 3, code)
    return torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=device)[code]


def inject_repeats(contigs, level, seed=SEED, min_copies=400):
    """writes the repeat families of REPEAT_LEVELS[level] into the truth contigs (in place); returns a description"""
    spec = REPEAT_LEVELS[level]
    total = sum(int(c.numel()) for c in contigs)
    scale = min(1.0, total / 3e9)
    dev = contigs[0].device
    big = [i for i, c in enumerate(contigs) if c.numel() >= 200000]
    if not big:
        return {"level": level, "bases": 0}
    r = _Rng(seed * 131 + level)
    placed = {"alu": 0, "sat": 0, "tandem5": 0, "rdna": 0}

    def rand_codes(n, s):
        return (_lsr(_hash_range(s, 0, n, dev), 61) & 3).to(torch.uint8)

    # tandem arrays first (the dispersed copies then fall into them here and there, as they do in a genome)
    for a in range(spec["tandem5"]):
        ci = big[r.below(len(big))]
        c = contigs[ci]
        n = min(int(2_000_000 * max(scale, 0.05)), c.numel() // 8) // 5 * 5
        p = r.below(c.numel() - n)
        unit = torch.tensor(list(b"GGAAT" if a % 2 == 0 else b"CTTAC"), dtype=torch.uint8, device=dev)
        if a >= 2:
            unit = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)[rand_codes(5, seed + 17 * a).long()]
        c[p:p + n] = unit.repeat(n // 5)
        placed["tandem5"] += n
    for a in range(5):
        ci = big[r.below(len(big))]
        c = contigs[ci]
        ncop = max(int(spec["rdna"] * max(scale, 0.1)), 8)
        L = 45000
        while ncop * L > c.numel() // 4 and L > 2000:
            L //= 2
        cons = rand_codes(L, seed + 0x4D4A)                      # ONE unit for the five arrays
        cp = _mutated_copies(cons, ncop, 0.001, seed + 911 * (a + 1), dev)
        p = r.below(c.numel() - ncop * L)
        c[p:p + ncop * L] = cp.view(-1)
        placed["rdna"] += ncop * L
    for a in range(spec["sat"]):
        ci = big[r.below(len(big))]
        c = contigs[ci]
        ncop = max(int(10000 * scale), min_copies)
        while ncop * 171 > c.numel() // 8:
            ncop //= 2
        cons = rand_codes(171, seed + 0x5A7 + 31 * a)
        div = (0.02, 0.05, 0.10, 0.20)[a % 4]
        cp = _mutated_copies(cons, ncop, div, seed + 7717 * (a + 1), dev)
        p = r.below(c.numel() - ncop * 171)
        c[p:p + ncop * 171] = cp.view(-1)
        placed["sat"] += ncop * 171
    # dispersed families: the copies of one family are dealt to the contigs by size, on a grid of 512-base cells picked by hash
    L = 300
    for a in range(spec["alu"]):
        ncop_total = max(int(100000 * scale), min_copies)
        cons = rand_codes(L, seed + 0xA1 + 53 * a)
        for ci in big:
            c = contigs[ci]
            n = c.numel()
            ncells = n // 512 - 1
            want = ncop_total * n / float(sum(contigs[i].numel() for i in big))
            if ncells < 4 or want < 0.5:
                continue
            hc = _hash_range(seed + 0xCE11 + 97 * a + 1009 * ci, 0, ncells, dev)
            pick = (_lsr(hc, 11).double() * (1.0 / (1 << 53))) < min(0.5, want / ncells)
            cells = pick.nonzero().squeeze(1)
            if cells.numel() == 0:
                continue
            ncop = int(cells.numel())
            pos = cells * 512 + (_lsr(hc[cells], 3) % (512 - L))
            dv = 0.05 + 0.10 * (_lsr(_hash_range(seed + 0xD1 + a + 13 * ci, 0, ncop, dev), 11).double() * (1.0 / (1 << 53)))
            cp = _mutated_copies(cons, ncop, dv, seed + 333 * (a + 1) + 7 * ci, dev)
            idx = (pos.view(-1, 1) + torch.arange(L, device=dev).view(1, L)).view(-1)
            c[idx] = cp.view(-1)
            placed["alu"] += ncop * L
    placed.update(level=level, bases=sum(placed.values()), scale=scale)
    return placed


def make_assembly(truth, layout, seed=SEED, sub_rate=1e-4, unit=10000):
    """assembly = truth + substitutions (1e-4/base) + some 2-copy repeats collapsed
    (second copy replaced by novel sequence) + some unique segments duplicated,
    then N runs (one 50 kb run per contig, one N per ~1 Mb) and 0.5 % lower case."""
    out = []
    for ci, (t, reps) in enumerate(zip(truth, layout)):
        n = t.numel()
        dev = t.device
        a = t.clone()
        r = _Rng(seed * 77 + ci)
        two = [x for x in reps if x[0] == 2]
        for j, (_, pos, ul) in enumerate(two):
            if j % 5 == 0:                                   # collapse -> readK 2, asmK 1 (`over` bins)
                a[pos[1]:pos[1] + ul] = random_bases(ul, seed + 7777 * (ci + 1) + j, dev)
        if n >= 20 * unit:
            for _ in range(max(1, int(n * 0.002 / unit))):   # duplication -> asmK 2, readK 1 (`undr` bins)
                s, d = r.below(n - unit), r.below(n - unit)
                a[d:d + unit] = a[s:s + unit].clone()
        # substitutions
        chunk = 1 << 27
        for o in range(0, n, chunk):
            m = min(chunk, n - o)
            h = _hash_range(seed + 99991 * (ci + 1), o, m, dev)
            hit = (_lsr(h, 11).double() * (1.0 / (1 << 53))) < sub_rate
            idx = hit.nonzero().squeeze(1)
            if idx.numel():
                old = a[o + idx]
                code = ((old >> 1) & 3).long()
                new = (code + 1 + (_lsr(h[idx], 3) % 3)) & 3
                a[o + idx] = torch.tensor(list(b"ACTG"), dtype=torch.uint8, device=dev)[new]
        # N runs / single Ns / lower case
        if n > 200000:
            p = r.below(n - 50000)
            a[p:p + 50000] = ord("N")
        for _ in range(n // 1000000):
            a[r.below(n)] = ord("N")
        for o in range(0, n, chunk):
            m = min(chunk, n - o)
            h = _hash_range(seed + 424243 * (ci + 1), o, m, dev)
            low = (_lsr(h, 11).double() * (1.0 / (1 << 53))) < 0.005
            a[o:o + m] |= (low.to(torch.uint8) << 5)
        out.append(a)
    return out


_LUT = None


def _codes(seq):
    global _LUT
    if _LUT is None or _LUT.device != seq.device:
        lut = torch.full((256,), 4, dtype=torch.int64)
        for ch, c in zip(b"ACTGactg", [0, 1, 2, 3, 0, 1, 2, 3]):
            lut[ch] = c
        _LUT = lut.to(seq.device)
    return _LUT[seq.long()]


def strand_kmers(seq, k):
    """(forward k-mer, reverse-complement k-mer, valid) int64/int64/bool [n-k+1] of an ASCII uint8 tensor"""
    n = seq.numel()
    if n < k:
        z = torch.zeros(0, dtype=torch.int64, device=seq.device)
        return z, z, z.bool()
    c = _codes(seq)
    inv = (c > 3)
    c = c & 3
    m = n - k + 1
    f = torch.zeros(m, dtype=torch.int64, device=seq.device)
    r = torch.zeros(m, dtype=torch.int64, device=seq.device)
    for j in range(k):
        w = c[j:j + m]
        f = (f << 2) | w
        r = r | ((w ^ 2) << (2 * j))
    cs = torch.cumsum(inv.to(torch.int32), 0)
    win = cs[k - 1:] - torch.cat([torch.zeros(1, dtype=cs.dtype, device=cs.device), cs[:m - 1]])
    return f, r, win == 0


def canonical_kmers(seq, k):
    """(canonical k-mer int64 [n-k+1], valid bool [n-k+1]) of an ASCII uint8 tensor"""
    n = seq.numel()
    if n < k:
        z = torch.zeros(0, dtype=torch.int64, device=seq.device)
        return z, z.bool()
    c = _codes(seq)
    inv = (c > 3)
    c = c & 3
    m = n - k + 1
    f = torch.zeros(m, dtype=torch.int64, device=seq.device)
    r = torch.zeros(m, dtype=torch.int64, device=seq.device)
    for j in range(k):
        w = c[j:j + m]
        f = (f << 2) | w
        r = r | ((w ^ 2) << (2 * j))
    cs = torch.cumsum(inv.to(torch.int32), 0)
    win = cs[k - 1:] - torch.cat([torch.zeros(1, dtype=cs.dtype, device=cs.device), cs[:m - 1]])
    return torch.minimum(f, r), win == 0


def poisson_cdf(lam, nmax=None):
    nmax = nmax or int(lam + 12 * math.sqrt(lam) + 20)
    logp = [-lam + i * math.log(lam) - math.lgamma(i + 1) for i in range(nmax)]
    cdf, s = [], 0.0
    for lp in logp:
        s += math.exp(lp)
        cdf.append(min(s, 1.0))
    return torch.tensor(cdf, dtype=torch.float64)


def add_reads_from_truth(ix, truth, k, lam, seed=SEED, chunk=1 << 25):
    """Every k-mer OCCURRENCE of the truth genome contributes an independent
    Poisson(lam) count; mfx_index_add_read sums duplicates, so a k-mer with copy
    number c ends with Poisson(lam*c) (0 -> absent).  Uses only the public C ABI."""
    total = 0
    for ci, t in enumerate(truth):
        dev = t.device
        cdf = poisson_cdf(lam).to(dev)
        n = t.numel()
        for o in range(0, max(n - k + 1, 0), chunk):
            m = min(chunk, n - k + 1 - o)
            km, ok = canonical_kmers(t[o:o + m + k - 1], k)
            h = _hash_range(seed + 5550001 * (ci + 1), o, m, dev)
            u = _lsr(h, 11).double() * (1.0 / (1 << 53))
            v = torch.searchsorted(cdf, u, right=True).to(torch.int32)
            v = torch.where(ok, v, torch.zeros_like(v))
            ix.add_read(km.contiguous(), v.contiguous())
            total += int(m)
    return total


def add_error_kmers(ix, n_err, k, seed=SEED, chunk=1 << 25):
    """n_err random canonical k-mers with counts {1: 80 %, 2: 15 %, 3: 5 %} (sequencing errors)"""
    dev = "cuda"
    mask = (1 << (2 * k)) - 1
    for o in range(0, n_err, chunk):
        m = min(chunk, n_err - o)
        h = _hash_range(seed + 0xE44, o, m, dev)
        f = h & mask
        # reverse complement by bit tricks on int64 tensors
        r = torch.zeros_like(f)
        x = f
        for _ in range(k):
            r = (r << 2) | ((x & 3) ^ 2)
            x = x >> 2
        km = torch.minimum(f, r)
        u = _lsr(splitmix64(h), 11).double() * (1.0 / (1 << 53))
        v = (1 + (u >= 0.8).to(torch.int32) + (u >= 0.95).to(torch.int32))
        ix.add_read(km.contiguous(), v.contiguous())


def add_neighbor_error_kmers(ix, truth, k, seed=SEED, per_position=1.0, chunk=1 << 25):
    """Error k-mers the way sequencing makes them: single-base substitutions of TRUE k-mers (about `per_position`
    per genome position, counts {1: 80 %, 2: 15 %, 3: 5 %}).  Unlike uniformly random k-mers they share (k-1)- and
    (k-2)-mers with their parent, i.e. they land in the parent's minimizer bucket -- the harder case for the
    minimizer-keyed placement.  Not the bench workload (SURVEY 8d specifies random error k-mers); robustness check."""
    for ci, t in enumerate(truth):
        dev = t.device
        n = t.numel()
        for o in range(0, max(n - k + 1, 0), chunk):
            m = min(chunk, n - k + 1 - o)
            f, r, ok = strand_kmers(t[o:o + m + k - 1], k)
            h = _hash_range(seed + 7770001 * (ci + 1), o, m, dev)
            keep = ok & ((_lsr(h, 40).double() * (1.0 / (1 << 24))) < per_position)
            j = _lsr(h, 3) % k                                  # substituted base, 0 = leftmost
            d = 1 + (_lsr(h, 17) % 3)                           # code xor 1..3: always a different base
            f2 = f ^ (d << (2 * (k - 1 - j)))
            r2 = r ^ (d << (2 * j))
            km = torch.minimum(f2, r2)
            u = _lsr(splitmix64(h), 11).double() * (1.0 / (1 << 53))
            v = (1 + (u >= 0.8).to(torch.int32) + (u >= 0.95).to(torch.int32))
            v = torch.where(keep, v, torch.zeros_like(v))
            ix.add_read(km.contiguous(), v.contiguous())


def build_world(m, total_bases, k=21, lam=26.0, ncontigs=24, seed=SEED, device=0, err_factor=1.0, verbose=None,
                err_mode="random", index_factory=None, seq_only=False, repeats=0):
    """Full synthetic -hist workload resident on `device`: returns (index, sequences, info).
    index_factory(k, capacity, device=) may supply the index (e.g. a fan-out over the shards of a sharded index).
    seq_only: the SEQUENCE-ONLY index the CLI builds for -hist / -dump (mfx_index_create_for_seq): the assembly's k-mers
    are claimed and counted first, the read database -- the very same k-mers and counts as for the full index -- then
    only updates them.
    repeats: 0 = SURVEY 8(d)'s i.i.d. genome; 1 / 3 / 10 = the `repeats` world at that level (inject_repeats): info["repeats"]."""
    import time
    torch.cuda.set_device(device)
    dev = "cuda:%d" % device
    say = verbose or (lambda *a: None)
    t0 = time.time()
    sizes = contig_sizes(total_bases, ncontigs)
    truth, layout = make_truth(sizes, seed, dev)
    rep_info = inject_repeats(truth, repeats, seed) if repeats else None
    asm = make_assembly(truth, layout, seed)
    torch.cuda.synchronize()
    say("genome+assembly generated: %.1fs%s" % (time.time() - t0, (" (repeat families: %r)" % (rep_info,)) if rep_info else ""))
    n_err = int(total_bases * err_factor)
    if seq_only:
        seqs = m.Sequences.from_device([a.data_ptr() for a in asm], [a.numel() for a in asm], device=device)
        ix = m.Index.for_seq(k, int(total_bases) + 1024, device=device)
        ix.count_asm(seqs)
        say("assembly k-mers claimed + counted: %.1fs" % (time.time() - t0))
        add_reads_from_truth(ix, truth, k, lam, seed)
        del truth
        add_error_kmers(ix, n_err, k, seed)
        torch.cuda.synchronize()
        say("read database (truth counts + error k-mers) applied, update-only: %.1fs" % (time.time() - t0))
        info = ix.info()
        info["build_s"] = time.time() - t0
        info["sizes"] = sizes
        info["repeats"] = rep_info
        return ix, seqs, asm, info
    cap = int(total_bases * 1.03) + n_err + 1024
    ix = (index_factory or m.Index)(k, cap, device=device)
    add_reads_from_truth(ix, truth, k, lam, seed)
    torch.cuda.synchronize()
    say("read counts from truth added: %.1fs" % (time.time() - t0))
    if err_mode == "neighbor":
        add_neighbor_error_kmers(ix, truth, k, seed, per_position=err_factor)
    del truth
    if err_mode != "neighbor":
        add_error_kmers(ix, n_err, k, seed)
    torch.cuda.synchronize()
    say("error k-mers added: %.1fs" % (time.time() - t0))
    seqs = m.Sequences.from_device([a.data_ptr() for a in asm], [a.numel() for a in asm], device=device)
    ix.count_asm(seqs)
    say("assembly k-mers counted: %.1fs" % (time.time() - t0))
    info = ix.info()
    info["build_s"] = time.time() - t0
    info["sizes"] = sizes
    info["repeats"] = rep_info
    return ix, seqs, asm, info
