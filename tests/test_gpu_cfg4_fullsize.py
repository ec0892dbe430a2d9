"""BASELINE config 4 at its named size on one MI355X: -polish of a 3 Gb assembly (24 human-proportioned contigs, k=21)
against a ~4 M-call VCF (the 8-GPU form of the same run gives every GPU a contiguous run of contigs, see
`merfin -devices`; its pieces are exercised here one after the other).  The CPU oracle cannot reach this size -- it is
compared byte for byte at fixture size in tests/test_gpu_variants.py -- so the checks are properties of the domain
(merfin-variants.C:131-310, varMer.C:48-145):
  - the assembly differs from the truth genome at the positions of 80 % of the calls (real errors, the call restores
    the truth base) and the other 20 % are decoys (the call would break a correct base); the read k-mers come from
    the truth.  -polish must select (nearly) all corrections and (nearly) no decoy, and every selected 1/1 record's
    ALT must be the truth base;
  - cutting the contigs into contiguous runs (what the slots of -devices get) and concatenating the outputs gives the
    whole run's records, byte for byte;
  - every input record is accounted for: clusters evaluated + skipped, records selected <= records read.
MFX_TEST_CFG4_BASES / MFX_TEST_CFG4_CALLS scale it down for a smaller device (the test then says so)."""
import os
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

BASES = int(float(os.environ.get("MFX_TEST_CFG4_BASES", "3e9")))
CALLS = int(float(os.environ.get("MFX_TEST_CFG4_CALLS", "4e6")))
K, LAM, NCONTIG = 21, 26.0, 24


def test_cfg4_polish_at_named_size(tmp_path):
    torch = pytest.importorskip("torch")
    import gc
    import merfin_amd as m
    from tools import synth_torch as st
    gc.collect()
    torch.cuda.empty_cache()
    free, _tot = torch.cuda.mem_get_info()
    need = BASES * 2.2 / 0.7 * 16 + 3 * BASES + 8e9
    if need > free:
        pytest.skip("config 4 at %d bases needs %.0f GB of free HBM, %.0f GB available" % (BASES, need / 1e9, free / 1e9))
    r = np.random.default_rng(11)
    sizes = st.contig_sizes(BASES, NCONTIG)
    names = ["chr%d" % (i + 1) for i in range(NCONTIG)]
    truth = [st.random_bases(n, st.SEED + 17 * (i + 1), "cuda") for i, n in enumerate(sizes)]
    lines = ["##fileformat=VCFv4.2"] + ["##contig=<ID=%s,length=%d>" % (n, s) for n, s in zip(names, sizes)]
    lines.append("#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tSAMPLE")
    ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
    asm, calls = [], {}                                       # calls[(contig, pos1)] = (is_err, truth base)
    for ci, t in enumerate(truth):
        a = t.cpu().numpy().copy()
        n = len(a)
        nv = max(8, CALLS * n // BASES)
        # half of the calls come in tight groups (clusters of several calls within 2k bases), as DeepVariant's do
        pos = np.unique(np.concatenate([r.integers(30, n - 30, size=nv // 2),
                                        (r.integers(30, n - 200, size=nv // 8)[:, None] + r.integers(0, 40, size=(nv // 8, 4))).ravel()]))
        tru = a[pos].copy()
        wrong = ACGT[(np.searchsorted(ACGT, tru) + r.integers(1, 4, size=len(pos))) % 4]
        is_err = r.random(len(pos)) < 0.8
        a[pos[is_err]] = wrong[is_err]
        nm = names[ci]
        for p, te, w, e in zip(pos.tolist(), tru.tolist(), wrong.tolist(), is_err.tolist()):
            ref, alt = (chr(w), chr(te)) if e else (chr(te), chr(w))
            lines.append("%s\t%d\t.\t%s\t%s\t30\tPASS\t.\tGT\t1/1" % (nm, p + 1, ref, alt))
            calls[(nm, p + 1)] = (e, chr(te))
        asm.append(a.tobytes())
        del a
    vcf = str(tmp_path / "in.vcf")
    with open(vcf, "w") as f:
        f.write("\n".join(lines) + "\n")
    ncalls = len(calls)
    del lines
    ix = m.Index(K, int(BASES * 2.2))
    st.add_reads_from_truth(ix, truth, K, LAM)
    del truth
    torch.cuda.empty_cache()
    seqs = m.Sequences(asm)
    ix.count_asm(seqs)
    del seqs
    ev = m.Evaluator(ix, m.KParams(LAM))

    whole = str(tmp_path / "whole.vcf")
    t0 = time.time()
    ncl = ev.variants("polish", vcf, names, asm, whole, log_path=str(tmp_path / "log.txt"))
    dt = time.time() - t0
    body = [l for l in open(whole).read().splitlines() if not l.startswith("#")]
    print("\nconfig 4: %d bases, %d calls, %d clusters in %.1f s (%.0f clusters/s), %d records selected"
          % (BASES, ncalls, ncl, dt, ncl / dt, len(body)))
    assert 0.4 * ncalls < ncl <= ncalls and len(body) <= ncalls

    # the selection restores the truth
    fixed = decoys = wrong_alt = 0
    for l in body:
        w = l.split("\t")
        e, tb = calls[(w[0], int(w[1]))]
        if w[9].startswith("1/1"):
            if e:
                fixed += 1
                wrong_alt += w[4] != tb
            else:
                decoys += 1
    n_err = sum(1 for e, _ in calls.values() if e)
    print("          corrections selected %d of %d (%.2f %%), decoys selected %d of %d" % (fixed, n_err, 100.0 * fixed / n_err, decoys, ncalls - n_err))
    assert wrong_alt == 0
    assert fixed >= 0.97 * n_err, (fixed, n_err)
    assert decoys <= 0.01 * (ncalls - n_err) + 5, (decoys, ncalls - n_err)

    # contiguous runs of contigs, one after the other, as the slots of -devices get them: same records
    parts = []
    cuts = [0, 5, 11, 17, NCONTIG]
    nparts = 0
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        pth = str(tmp_path / ("part%d.vcf" % lo))
        nparts += ev.variants("polish", vcf, names[lo:hi], asm[lo:hi], pth, log_path=str(tmp_path / "plog.txt"))
        parts += [l for l in open(pth).read().splitlines() if not l.startswith("#")]
    assert nparts == ncl and parts == body
