"""BASELINE config 1 at its named size (5 Mb, 20x reads; MFX_TEST_CFG1_MB scales it): a synthetic genome, ACTUAL simulated reads (20x, 150 bp, both strands, 0.5 %
substitution errors), their canonical 21-mers counted by a sort-based counter in the harness (numpy) -- the
read database a `meryl count` would produce -- then `-hist` with -peak 17.3 on the CPU oracle and on the GPU.
Also checks the GPU's own counting route: every k-mer OCCURRENCE inserted with value 1 must sum to the same
counts (mfx_index_add_read adds duplicates), i.e. the index build is exact at a few 10^7 inserts."""
import os

import numpy as np
import pytest

from oracle import pyoracle as po
from tests import synth
from tests.test_gpu_parity import assert_hist_equal, build_index, oracle_hist

pytestmark = pytest.mark.gpu


def canonical_occurrences(k, reads):
    """every canonical k-mer occurrence of a [n_reads, L] uint8 ACGT matrix (vectorised rolling encode)"""
    code = np.zeros(256, dtype=np.uint64)
    for ch, c in zip(b"ACTG", (0, 1, 2, 3)):
        code[ch] = c
    c = code[reads]                                        # [n, L]
    n, L = c.shape
    m = L - k + 1
    f = np.zeros((n, m), dtype=np.uint64)
    r = np.zeros((n, m), dtype=np.uint64)
    for j in range(k):
        w = c[:, j:j + m]
        f = (f << np.uint64(2)) | w
        r = r | ((w ^ np.uint64(2)) << np.uint64(2 * j))
    return np.minimum(f, r).ravel()


def test_simulated_reads_counted_in_the_harness():
    import merfin_amd as m
    k, peak, cov, L = 21, 17.3, 20, 150
    r = synth.rng(20260928)
    scale = float(os.environ.get("MFX_TEST_CFG1_MB", "5"))   # 5 Mb = the size BASELINE config 1 names (1 = quick run)
    truth = synth.make_truth(r, tuple(int(x * scale) for x in (400000, 250000, 150000, 100000, 60000, 30000, 9000, 1000)))
    asm = synth.mutate(r, truth, sub_rate=2e-4)
    genome = np.concatenate(truth)
    bounds = np.cumsum([0] + [len(t) for t in truth])
    n_reads = cov * len(genome) // L
    start = r.integers(0, len(genome) - L, size=n_reads)
    ctg = np.searchsorted(bounds, start, side="right") - 1
    start = np.minimum(start, bounds[ctg + 1] - L)         # keep each read inside its contig
    reads = genome[start[:, None] + np.arange(L)[None, :]]
    err = r.random(reads.shape) < 0.005
    sub = synth.BASES[(np.searchsorted(synth.BASES, reads) + r.integers(1, 4, size=reads.shape)) % 4]
    reads = np.where(err, sub, reads)
    rc = r.random(n_reads) < 0.5                           # half of the reads come from the other strand
    comp = np.zeros(256, dtype=np.uint8)
    for a, b in zip(b"ACGT", b"TGCA"):
        comp[a] = b
    reads[rc] = comp[reads[rc]][:, ::-1]
    occ = canonical_occurrences(k, reads)
    rk, rv = np.unique(occ, return_counts=True)            # the sort-based counter
    rv = rv.astype(np.uint32)
    assert len(occ) > 15_000_000 * scale and 2.0 < len(rk) / len(genome) < 6.0      # true k-mers + error k-mers
    contigs = [c.tobytes() for c in asm]
    ak, av = po.count_kmers(k, contigs)
    p, g, ka, km = oracle_hist(k, peak, contigs, (rk, rv), (ak, av))
    assert g.kmissing > 0 and g.kasm > 0.99 * len(genome)
    # (a) the harness's counts loaded as a database
    ix = build_index(m, k, (rk, rv), (ak, av))
    seqs = m.Sequences(contigs)
    assert_hist_equal(m.Evaluator(ix, m.KParams(peak)).hist(seqs), g, ka, km, k)
    # (b) the GPU as the counter: one insert per occurrence
    ix2 = m.Index(k, len(rk) + len(ak) + 16)
    ones = np.ones(1 << 22, dtype=np.uint32)
    for o in range(0, len(occ), 1 << 22):
        chunk = occ[o:o + (1 << 22)]
        ix2.add_read(chunk, ones[:len(chunk)])
    ix2.count_asm(seqs)
    ek, er, ea = ix2.export()
    keep = er > 0
    np.testing.assert_array_equal(ek[keep], rk)
    np.testing.assert_array_equal(er[keep], rv)
    assert_hist_equal(m.Evaluator(ix2, m.KParams(peak)).hist(seqs), g, ka, km, k)
