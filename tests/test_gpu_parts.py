"""PARTS of one assembly, one per slot (mfx_index_claim_seq on the slot's contigs + mfx_index_count_claimed over the whole
assembly + update-only read database, mfx_hist_run_parts): what config 5's `-hist` runs on the 8-GPU node -- every device
evaluates its contigs on a sequence-only index of THEIR k-mers, nothing is exchanged -- against the oracle on the whole
assembly (merfin-histogram.C:54-91; the reference's own way to spread a run is contigs over processes,
scripts/parallel1/merfin.sh:68-85).  All slots on device 0 here; tests/test_gpu_multidevice.py has the >= 2-GPU twin."""
import os

import numpy as np
import pytest

from oracle import pyoracle as po
from tests import synth
from tests.test_gpu_parity import assert_hist_equal, oracle_hist

pytestmark = pytest.mark.gpu


def split_contigs(contigs, nslots):
    """contig numbers per slot, balanced by bases (largest first), every slot's list ascending"""
    order = sorted(range(len(contigs)), key=lambda i: -len(contigs[i]))
    load, parts = [0] * nslots, [[] for _ in range(nslots)]
    for i in order:
        d = load.index(min(load))
        parts[d].append(i)
        load[d] += len(contigs[i])
    return [sorted(p) for p in parts]


def build_parts(m, k, contigs, read, nslots, asm_db=None, lo=0, hi=2**64 - 1, device_of=lambda d: 0):
    whole = {}
    ids = split_contigs(contigs, nslots)
    ixs, seqs = [], []
    for d, mine in enumerate(ids):
        dev = device_of(d)
        if dev not in whole:
            whole[dev] = m.Sequences(contigs, device=dev)
        own = m.Sequences([contigs[i] for i in mine], device=dev)
        ix = m.Index.for_seq(k, sum(len(contigs[i]) for i in mine) + 16, device=dev)
        ix.claim_seq(own)
        if asm_db is None:
            ix.count_claimed(whole[dev])                       # the assembly counts of the claimed k-mers, from every contig
        else:
            ix.add_asm(*asm_db)                                # ... or from -seqmers: update-only as well
        ix.add_read(read[0], read[1], lo, hi)
        ixs.append(ix)
        seqs.append(own)
    return ixs, seqs, ids


@pytest.mark.parametrize("k,peak,use_prob,nslots,from_db", [(21, 17.3, False, 3, False), (21, 26.0, True, 5, True), (31, 17.3, False, 4, False),
                                                            (25, 9.0, False, 2, False), (15, 9.0, False, 8, False), (22, 9.0, False, 3, True)])
def test_parts_of_an_assembly_equal_the_oracle_on_the_whole(k, peak, use_prob, nslots, from_db, golden_dir):
    import merfin_amd as m
    contigs, read, asm = synth.world(k=k, peak=peak, seed=1300 + k + nslots, sizes=(30000, 9000, 4097, 4096, 300, 20000, 0, 12000, k, 5000))
    probK = probP = None
    if use_prob:
        probK, probP = po.load_kmetric(os.path.join(golden_dir, "example_lookup_table.txt"))
    p, g, ka, km = oracle_hist(k, peak, contigs, read, asm, probK, probP)
    ixs, seqs, ids = build_parts(m, k, contigs, read, nslots, asm_db=asm if from_db else None)
    kp = m.KParams(peak, probK, probP)
    evs = [m.Evaluator(ix, kp) for ix in ixs]
    res = m.hist_parts(evs, seqs, ids, len(contigs))
    assert_hist_equal(res, g, ka, km, k)
    # a slot's table answers value() of ITS k-mers as the full tables of the whole run do: the assembly count is the WHOLE assembly's
    ad, rd = dict(zip(asm[0].tolist(), asm[1].tolist())), dict(zip(read[0].tolist(), read[1].tolist()))
    total = 0
    for d, ix in enumerate(ixs):
        ek, er, ea = ix.export()
        total += len(ek)
        np.testing.assert_array_equal(ea, np.array([ad[x] for x in ek.tolist()], dtype=np.uint32))
        np.testing.assert_array_equal(er, np.array([rd.get(x, 0) for x in ek.tolist()], dtype=np.uint32))
        # and the slot's own contigs alone give the oracle's per-contig counters
        one = m.Evaluator(ix, kp).hist(seqs[d])
        np.testing.assert_array_equal(one.contig_kasm(), ka[ids[d]])
        np.testing.assert_array_equal(one.contig_kmissing(), km[ids[d]])
    assert total >= len(asm[0])                                  # every k-mer of the assembly has a slot in at least one part
    # another part's sequence on a slot's table is refused (the table does not hold its k-mers)
    if nslots > 1 and len(ids[0]) and len(ids[1]) and sum(len(contigs[i]) for i in ids[1]) >= k:
        with pytest.raises(m.MfxError, match="ANOTHER sequence"):
            evs[0].hist(seqs[1])


def test_parts_rules():
    import merfin_amd as m
    k = 21
    contigs, read, asm = synth.world(k=k, peak=9.0, seed=77)
    whole = m.Sequences(contigs)
    with pytest.raises(m.MfxError, match="sequence-only"):
        m.Index(k, 1000).count_claimed(whole)                     # a full index counts by mfx_index_count_asm
    ix = m.Index.for_seq(k, sum(len(c) for c in contigs) + 16)
    own = m.Sequences(contigs[:1])
    ix.claim_seq(own)
    ix.count_claimed(whole)
    ix.count_claimed(m.Sequences([b""]))                          # (an empty sequence counts nothing)
    with pytest.raises(m.MfxError, match="claimed before"):
        ix.claim_seq(own)                                         # counts arrived: no more claims
    evs = [m.Evaluator(ix, m.KParams(9.0))]
    with pytest.raises(m.MfxError, match="two slots|out of range"):
        m.hist_parts(evs, [own], [[len(contigs)]], len(contigs))
