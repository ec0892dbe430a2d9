"""BASELINE config 5's shape on one MI355X: k = 31, the index SHARDED over 8 slots (all on device 0 here -- the 15 Gb /
2e10-k-mer wheat case needs the 8-GPU node for its 460 GB of table; what one GPU holds is a 3 Gb genome, i.e. the same
200 GB-class table cut into 8 shards), `-hist` routed to the owners (mfx_hist_run_sharded) and `-completeness` summed
over the shards.  The oracle cannot reach this size, so the sharded results are compared with the UNSHARDED run of the
same library on the same world (built first, read out, freed): integers bit-exact, koverCpy to 1e-12, the 64 per-piece
completeness sums exactly (they are integer-valued, merfin-completeness.C:117-123); plus the domain's own invariant
binned + kmissing == kasm.  Oracle parity of both paths at k = 31 is in tests/test_gpu_sharded.py / test_gpu_parity.py.
MFX_TEST_CFG5_BASES scales it down for a smaller device (the test then says so)."""
import os
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

BASES = int(float(os.environ.get("MFX_TEST_CFG5_BASES", "3e9")))
K, LAM, WORLD = 31, 26.0, 8


class _Fan:
    """what synth_torch.build_world needs of an index, fanned out to the shards (each keeps the k-mers it owns)"""

    def __init__(self, shards):
        self.shards = shards

    def add_read(self, k, v):
        for s in self.shards:
            s.add_read(k, v)

    def count_asm(self, seqs):
        for s in self.shards:
            s.count_asm(seqs)

    def info(self):
        infos = [s.info() for s in self.shards]
        return {"distinct": sum(i["distinct"] for i in infos)}


def test_cfg5_sharded_equals_whole_at_scale(monkeypatch):
    torch = pytest.importorskip("torch")
    import gc
    import merfin_amd as m
    from tools import synth_torch as st
    gc.collect()
    torch.cuda.empty_cache()
    monkeypatch.setenv("MFX_LOAD_FACTOR", "0.6")              # 8 tables + 1 assembly must fit next to each other
    free, _tot = torch.cuda.mem_get_info()
    need = BASES * 2.03 * 1.15 / 0.6 * 16 + 6 * BASES + 12e9
    if need > free:
        pytest.skip("config-5 shape at %d bases needs %.0f GB of free HBM, %.0f GB available" % (BASES, need / 1e9, free / 1e9))
    kp = m.KParams(LAM)

    t0 = time.time()
    ix, seqs, asm, info = st.build_world(m, BASES, k=K, lam=LAM, ncontigs=24)
    ev = m.Evaluator(ix, kp)
    whole = ev.hist(seqs)
    W = dict(undr=whole.undr().copy(), over=whole.over().copy(), kasm=whole.kasm, kmissing=whole.kmissing, kover=whole.koverCpy,
             ckasm=whole.contig_kasm().copy(), ckmis=whole.contig_kmissing().copy(), distinct=info["distinct"])
    wt, wu = ev.completeness_pieces()
    assert int(W["undr"].sum() + W["over"].sum()) + W["kmissing"] == W["kasm"] and W["kasm"] > 0.97 * BASES
    t_whole = time.time() - t0
    del whole
    ev.close(); seqs.close(); ix.close()
    del ev, seqs, ix, asm
    gc.collect()
    torch.cuda.empty_cache()

    t0 = time.time()
    shards = []

    def factory(k, cap, device=0):
        for r in range(WORLD):
            s = m.Index(k, int(cap / WORLD * 1.15) + 1024, device=device)
            s.set_shard(r, WORLD)
            shards.append(s)
        return _Fan(shards)

    fan, seqs, asm, info = st.build_world(m, BASES, k=K, lam=LAM, ncontigs=24, index_factory=factory)
    assert info["distinct"] == W["distinct"]                   # every k-mer has exactly one owner
    sizes = [s.info()["distinct"] for s in shards]
    assert max(sizes) < 1.1 * min(sizes)                       # the owner hash balances the shards
    evs = [m.Evaluator(s, kp) for s in shards]
    routers = [m.Router(s, WORLD, min(16384, seqs.ntiles)) for s in shards]     # tiles per round, as the CLI
    t1 = time.time()
    res = m.hist_sharded(evs, routers, [seqs] * WORLD)
    t_first = time.time() - t1                                 # the first run also makes the routers' group buffers
    t2 = time.time()
    res2 = m.hist_sharded(evs, routers, [seqs] * WORLD)
    t_hist = time.time() - t2
    assert (res2.kasm, res2.kmissing, res2.koverCpy) == (res.kasm, res.kmissing, res.koverCpy)     # bit-stable run to run (fixed-point koverCpy)
    assert np.array_equal(res2.undr(), res.undr()) and np.array_equal(res2.over(), res.over())
    assert np.array_equal(res.undr(), W["undr"]) and np.array_equal(res.over(), W["over"])
    assert (res.kasm, res.kmissing) == (W["kasm"], W["kmissing"])
    assert np.array_equal(res.contig_kasm(), W["ckasm"]) and np.array_equal(res.contig_kmissing(), W["ckmis"])
    assert abs(res.koverCpy - W["kover"]) <= 1e-12 * max(abs(W["kover"]), 1.0)
    st_, su_ = np.zeros(64), np.zeros(64)
    for e in evs:
        t, u = e.completeness_pieces()
        st_ += t
        su_ += u
    assert np.array_equal(st_, wt) and np.array_equal(su_, wu)
    # ---- the same -hist by PARTS (what `merfin -hist -devices 0-7 -sharded` runs since round 4): every slot holds a part of the
    # contigs and the sequence-only index of THEIR k-mers (claimed from its contigs, counted over the whole assembly, the read
    # database update-only) -- no routing, no exchange; the hash-sharded tables above stay what -completeness runs on
    for e in evs:
        e.close()
    for r_ in routers:
        r_.close() if hasattr(r_, "close") else None
    for s_ in shards:
        s_.close()
    del evs, routers, shards, fan
    gc.collect()
    torch.cuda.empty_cache()
    monkeypatch.delenv("MFX_LOAD_FACTOR")
    from tests.test_gpu_parts import split_contigs
    t0p = time.time()
    lens = [int(a.numel()) for a in asm]
    ids = split_contigs([range(n) for n in lens], WORLD)
    pix, pseq = [], []
    for mine in ids:
        own = m.Sequences.from_device([asm[i].data_ptr() for i in mine], [lens[i] for i in mine])
        ix = m.Index.for_seq(K, sum(lens[i] for i in mine) + 1024)
        ix.claim_seq(own)
        ix.count_claimed(seqs)
        pix.append(ix)
        pseq.append(own)

    class _Parts:
        def add_read(self, k_, v_):
            for ix in pix:
                ix.add_read(k_, v_)
    truth, layout = st.make_truth(st.contig_sizes(BASES, 24), st.SEED, "cuda:0")
    st.add_reads_from_truth(_Parts(), truth, K, LAM, st.SEED)
    del truth
    st.add_error_kmers(_Parts(), int(BASES * 1.0), K, st.SEED)
    torch.cuda.synchronize()
    t_pbuild = time.time() - t0p
    pevs = [m.Evaluator(ix, kp) for ix in pix]
    pres = m.hist_parts(pevs, pseq, ids, len(lens))
    tp = []
    for _ in range(3):
        t3 = time.time()
        pres = m.hist_parts(pevs, pseq, ids, len(lens))
        tp.append(time.time() - t3)
    assert np.array_equal(pres.undr(), W["undr"]) and np.array_equal(pres.over(), W["over"])
    assert (pres.kasm, pres.kmissing) == (W["kasm"], W["kmissing"])
    assert np.array_equal(pres.contig_kasm(), W["ckasm"]) and np.array_equal(pres.contig_kmissing(), W["ckmis"])
    assert abs(pres.koverCpy - W["kover"]) <= 1e-12 * max(abs(W["kover"]), 1.0)
    psz = [ix.info() for ix in pix]
    print("\nconfig-5 shape by PARTS: %d slots on one GPU, tables %.1f GB in all (%d k-mers claimed), build %.1f s; -hist %.3f s = %.1f G k-mers/s, no exchange"
          % (WORLD, sum(i["bytes"] for i in psz) / 1e9, sum(i["distinct"] for i in psz), t_pbuild, min(tp), pres.kasm / min(tp) / 1e9))
    print("config-5 shape: %d bases, k=%d, %d k-mers in %d shards (%.1f-%.1f M each); whole build+hist %.1f s, sharded build %.1f s, "
          "sharded -hist %.3f s = %.1f G k-mers/s through the route->owner loop on one GPU (first run, which makes the group buffers: %.3f s)"
          % (BASES, K, W["distinct"], WORLD, min(sizes) / 1e6, max(sizes) / 1e6, t_whole, t1 - t0, t_hist, res.kasm / t_hist / 1e9, t_first))
