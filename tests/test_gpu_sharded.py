"""Sharded index (BASELINE config 5): every rank owns a hash-slice of the k-mer
table; -hist routes each k-mer to its owner.  The GPU box has one GPU, so the
ranks are (a) simulated in one process with a virtual exchange and (b) run as
two real processes sharing GPU 0, exchanging through gloo (host memory) --
RCCL refuses two ranks on one device.  Results must equal the oracle's."""
import os
import sys

import numpy as np
import pytest

from oracle import pyoracle as po
from tests import synth
from tests.test_gpu_parity import assert_hist_equal, oracle_hist

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build_shards(m, k, read, asm, world, contigs=None, count_from_seq=False):
    shards = []
    for r in range(world):
        ix = m.Index(k, len(read[0]) + len(asm[0]) + 16)      # generous: every shard sized for the whole set
        ix.set_shard(r, world)
        ix.add_read(*read)
        if count_from_seq:
            ix.count_asm(m.Sequences(contigs))
        else:
            ix.add_asm(*asm)
        shards.append(ix)
    return shards


# k = 31 is BASELINE config 5's k-mer size (hexaploid wheat, index sharded over 8 GPUs): the full 15 Gb / 2e10-k-mer
# size needs the 8-GPU node; here the same code path runs at k = 31 with 2...8 ranks at oracle-checkable size.
@pytest.mark.parametrize("world,mode,k", [(2, "mz", 21), (3, "mz", 21), (4, "plain", 21), (2, "mz", 31), (8, "mz", 31), (3, "plain", 31)])
def test_sharded_hist_virtual_exchange(world, mode, k, monkeypatch):
    torch = pytest.importorskip("torch")
    import merfin_amd as m
    from merfin_amd import distributed as D
    monkeypatch.setenv("MFX_HOME_MODE", mode)
    peak = 17.3
    contigs, read, asm = synth.world(k=k, peak=peak, seed=81)
    p, g, ka, km = oracle_hist(k, peak, contigs, read, asm)
    shards = _build_shards(m, k, read, asm, world, contigs, count_from_seq=True)
    # the shards partition the k-mer set
    union = np.union1d(read[0], asm[0])
    sizes = [s.info()["distinct"] for s in shards]
    assert sum(sizes) == len(union) and min(sizes) > 0.5 * len(union) / world
    evs = [m.Evaluator(s, m.KParams(peak)) for s in shards]
    seqs = m.Sequences(contigs)
    T = seqs.ntiles
    words = m.hist_words(evs[0].nbins, seqs.ncontigs)
    counts = torch.zeros(words, dtype=torch.int64, device="cuda")
    kover = torch.zeros(1, dtype=torch.float64, device="cuda")
    per = 3                                                   # several routing rounds per rank
    keys = torch.empty(per * m.TILE, dtype=torch.int64, device="cuda")
    ctg = torch.empty(per * m.TILE, dtype=torch.int32, device="cuda")
    for src in range(world):
        router = m.Router(shards[src], world, per)
        lo, hi = D.shard(T, src, world)
        for tb in range(lo, hi, per):
            send = router.route(seqs, tb, min(hi, tb + per), evs[0].nbins, counts, keys, ctg)
            off = 0
            for dst in range(world):
                n = int(send[dst])
                if n:
                    evs[dst].hist_keys_launch(keys[off:off + n], ctg[off:off + n], n, seqs.ncontigs, counts, kover)
                off += n
            torch.cuda.synchronize()
    res = m.result_from_counts(evs[0].nbins, counts.cpu().numpy().view(np.uint64), float(kover.item()), seqs.ncontigs)
    assert_hist_equal(res, g, ka, km, k)


@pytest.mark.parametrize("world,k", [(2, 21), (5, 21), (8, 31)])
def test_sharded_completeness_sums_to_whole(world, k):
    """every shard evaluates the k-mers it owns; the per-piece sums add up to the unsharded ones, bit for bit,
    and the total equals the oracle's computeCompleteness (merfin-completeness.C:70-123)"""
    import merfin_amd as m
    peak = 17.3
    contigs, read, asm = synth.world(k=k, peak=peak, seed=85)
    whole = m.Index(k, len(read[0]) + len(asm[0]) + 16)
    whole.add_read(*read)
    whole.add_asm(*asm)
    wt, wu = m.Evaluator(whole, m.KParams(peak)).completeness_pieces()
    st, su = np.zeros(64), np.zeros(64)
    for ix in _build_shards(m, k, read, asm, world):
        t, u = m.Evaluator(ix, m.KParams(peak)).completeness_pieces()
        st += t
        su += u
    assert (st == wt).all() and (su == wu).all() and wt.sum() > 0 and wu.sum() > 0
    p = po.Params(k, peak)
    for piece in range(64):                                   # oracle: the reference's per-piece merge loop
        lo, hi = piece << (2 * k - 6), (piece + 1) << (2 * k - 6)
        rs, as_ = (read[0] >= lo) & (read[0] < hi), (asm[0] >= lo) & (asm[0] < hi)
        assert (st[piece], su[piece]) == po.completeness_piece(p, read[0][rs], read[1][rs], asm[0][as_], asm[1][as_])


@pytest.mark.parametrize("world", [2, 8, 13])
def test_split_router_equals_sort_router(world, monkeypatch):
    """the sort-free counting split must emit exactly what the stable radix sort emits: k-mers grouped by owner,
    sequence order inside an owner, the contig id of every k-mer, the per-owner counts and the kasm counters"""
    torch = pytest.importorskip("torch")
    import merfin_amd as m
    k, peak = 21, 17.3
    contigs, read, asm = synth.world(k=k, peak=peak, seed=87)
    r = synth.rng(9)
    contigs = list(contigs) + [synth.random_contig(r, int(n)).tobytes() for n in r.integers(0, 6000, size=40)] + [b"N" * 9000]
    ix = m.Index(k, len(read[0]) + 16)
    ix.set_shard(0, world)
    ix.add_read(*read)
    seqs = m.Sequences(contigs)
    T = seqs.ntiles
    nb = 2048
    words = m.hist_words(nb, seqs.ncontigs)
    outs = []
    for force_sort in ("1", "0"):
        monkeypatch.setenv("MFX_ROUTE_SORT", force_sort)
        router = m.Router(ix, world, T)
        counts = torch.zeros(words, dtype=torch.int64, device="cuda")
        keys = torch.full((T * m.TILE,), -1, dtype=torch.int64, device="cuda")
        ctg = torch.full((T * m.TILE,), -1, dtype=torch.int32, device="cuda")
        got = []
        for tb, te in ((0, T // 3), (T // 3, T // 3), (T // 3, T)):      # includes an empty range
            send = router.route(seqs, tb, te, nb, counts, keys, ctg)
            n = int(send.sum())
            got.append((send.copy(), keys[:n].cpu().numpy().copy(), ctg[:n].cpu().numpy().copy()))
        outs.append((got, counts.cpu().numpy().copy()))
    (a, ca), (b, cb) = outs
    np.testing.assert_array_equal(ca, cb)
    assert int(ca[2 * nb]) > 0
    for (sa, ka, ga), (sb, kb, gb) in zip(a, b):
        np.testing.assert_array_equal(sa, sb)
        np.testing.assert_array_equal(ka, kb)
        np.testing.assert_array_equal(ga, gb)
    assert sum(int(x[0].sum()) for x in a) == int(ca[2 * nb])


def _worker(rank, world, port, tmp):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    import merfin_amd as m
    from merfin_amd import distributed as D
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    k, peak = 21, 17.3
    contigs, read, asm = synth.world(k=k, peak=peak, seed=83)
    ix = m.Index(k, len(read[0]) + len(asm[0]) + 16)
    ix.set_shard(rank, world)
    ix.add_read(*read)
    ix.add_asm(*asm)
    ev = m.Evaluator(ix, m.KParams(peak))
    seqs = m.Sequences(contigs)
    router = m.Router(ix, world, 4)
    counts = torch.zeros(m.hist_words(ev.nbins, seqs.ncontigs), dtype=torch.int64, device="cuda")
    kover = torch.zeros(1, dtype=torch.float64, device="cuda")

    def exchange_host(keys, contigs_, send):                 # gloo: through host memory
        sc = torch.tensor([int(x) for x in send], dtype=torch.int64)
        rc = torch.empty(world, dtype=torch.int64)
        dist.all_to_all_single(rc, sc)
        rcl, scl = [int(x) for x in rc.tolist()], [int(x) for x in send]
        n_in = sum(scl)
        rk = torch.empty(sum(rcl), dtype=torch.int64)
        rg = torch.empty(sum(rcl), dtype=torch.int32)
        dist.all_to_all_single(rk, keys[:n_in].cpu(), output_split_sizes=rcl, input_split_sizes=scl)
        dist.all_to_all_single(rg, contigs_[:n_in].cpu(), output_split_sizes=rcl, input_split_sizes=scl)
        return rk.cuda(), rg.cuda()

    T = seqs.ntiles
    lo, hi = D.shard(T, rank, world)
    per = router.max_tiles
    rounds = (-(-T // world) + per - 1) // per
    keys = torch.empty(per * m.TILE, dtype=torch.int64, device="cuda")
    ctg = torch.empty(per * m.TILE, dtype=torch.int32, device="cuda")
    for r in range(rounds):
        tb = min(hi, lo + r * per)
        te = min(hi, tb + per)
        send = router.route(seqs, tb, te, ev.nbins, counts, keys, ctg)
        rk, rg = exchange_host(keys, ctg, send)
        if rk.numel():
            ev.hist_keys_launch(rk, rg, rk.numel(), seqs.ncontigs, counts, kover)
        torch.cuda.synchronize()
    c, kv = counts.cpu(), kover.cpu()
    D.all_reduce_hist(c, kv)
    if rank == 0:
        res = m.result_from_counts(ev.nbins, c.numpy().view(np.uint64), float(kv.item()), seqs.ncontigs)
        p, g, ka, km = oracle_hist(k, peak, contigs, read, asm)
        try:
            assert_hist_equal(res, g, ka, km, k)
            open(os.path.join(tmp, "ok"), "w").write("1")
        except AssertionError as e:
            open(os.path.join(tmp, "ok"), "w").write("0 %r" % (e,))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_hist_two_processes_one_gpu(tmp_path):
    torch = pytest.importorskip("torch")
    import torch.multiprocessing as mp
    port = 29700 + os.getpid() % 1500
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "ok").read_text() == "1"


@pytest.mark.parametrize("world,k,per,sort", [(2, 21, 3, "0"), (4, 31, 2, "0"), (8, 31, 1000, "0"), (3, 21, 2, "1"), (8, 31, 1000, "1")])
def test_sharded_hist_one_process(world, k, per, sort, monkeypatch):
    """mfx_hist_run_sharded: the whole sharded -hist (route -> peer copies to the owners -> evaluate -> sum) driven by one
    process; the slots are shards of one table, here all on device 0.  `per` = tiles routed per round (several rounds / one).
    sort = "1": the radix-sort router (what more than 16 owners use), whose last kernel -- the gather that writes the
    groups -- must be complete before the owners' streams copy them."""
    import merfin_amd as m
    monkeypatch.setenv("MFX_ROUTE_SORT", sort)
    peak = 17.3
    contigs, read, asm = synth.world(k=k, peak=peak, seed=91)
    p, g, ka, km = oracle_hist(k, peak, contigs, read, asm)
    shards = _build_shards(m, k, read, asm, world, contigs, count_from_seq=True)
    evs = [m.Evaluator(s, m.KParams(peak)) for s in shards]
    seqs = m.Sequences(contigs)
    routers = [m.Router(s, world, min(per, seqs.ntiles)) for s in shards]
    res = m.hist_sharded(evs, routers, [seqs] * world)
    assert_hist_equal(res, g, ka, km, k)
    again = m.hist_sharded(evs, routers, [seqs] * world)
    assert again.koverCpy == res.koverCpy and again.kmissing == res.kmissing     # repeatable, buffers re-armed
    with pytest.raises(m.MfxError):                          # the slots must be shard 0..N-1 in order
        m.hist_sharded(evs[::-1], routers[::-1], [seqs] * world)


def test_cli_sharded_hist_and_completeness(tmp_path, golden_dir):
    """`merfin -hist -devices 0,0,0 -sharded` and `-completeness ... -sharded`: byte-identical to the unsharded CLI"""
    import subprocess
    exe = os.path.join(ROOT, "merfin_amd", "bin", "merfin")
    g = lambda n: os.path.join(golden_dir, n)
    common = ["-sequence", g("case1.fasta"), "-readmers", g("case1.read.kmers.txt"), "-peak", "17.3", "-prob", g("example_lookup_table.txt")]
    r = subprocess.run([exe, "-hist"] + common + ["-output", str(tmp_path / "s.hist"), "-devices", "0,0,0", "-sharded"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert (tmp_path / "s.hist").read_bytes() == open(g("case1.hist"), "rb").read()
    # -hist asks for the k-mers of -sequence only: every slot holds a PART of the sequences and the sequence-only index of its k-mers
    assert open(g("case1.summary")).read() in r.stderr and "Part 2 of 3" in r.stderr
    # ... and the hash-sharded full tables with the routed -hist (what -completeness and the variant modes run on) when asked for
    r = subprocess.run([exe, "-hist"] + common + ["-output", str(tmp_path / "s2.hist"), "-devices", "0,0,0", "-sharded"], capture_output=True, text=True,
                       env=dict(os.environ, MFX_CLI_FULL_INDEX="1"))
    assert r.returncode == 0, r.stderr
    assert (tmp_path / "s2.hist").read_bytes() == open(g("case1.hist"), "rb").read()
    assert open(g("case1.summary")).read() in r.stderr and "Shard 2 of 3" in r.stderr
    cargs = ["-completeness", "-readmers", g("case1.read.kmers.txt"), "-seqmers", g("case1.asm.kmers.txt"), "-peak", "17.3"]
    a = subprocess.run([exe] + cargs, capture_output=True, text=True)
    b = subprocess.run([exe] + cargs + ["-devices", "0,0", "-sharded"], capture_output=True, text=True)
    assert a.returncode == 0 and b.returncode == 0, b.stderr
    tail = lambda s: s[s.index("thread  0 total"):]
    assert tail(a.stderr) == tail(b.stderr)
    bad = subprocess.run([exe, "-dump"] + common + ["-output", str(tmp_path / "x"), "-devices", "0", "-sharded"], capture_output=True, text=True)
    assert bad.returncode == 1 and "-sharded needs at least two -devices" in bad.stderr


@pytest.mark.parametrize("k", [21, 31])
def test_one_pass_feeds_every_shard(tmp_path, k):
    """mfx_index_load_db_multi: ONE decode of a database (flat binary, `meryl print` text, meryl-layout directory) feeds
    all shards of a process, each keeping the k-mers it owns -- the same tables as N separate loads"""
    import merfin_amd as m
    from tests import meryl_layout
    from tests.test_cli import _write_text_db
    world = 4
    contigs, read, asm = synth.world(k=k, peak=17.3, seed=95, sizes=(9000, 3000, 400), err_kmers=500)
    flat, text, mdir = str(tmp_path / "r.mfxk"), str(tmp_path / "r.txt"), str(tmp_path / "r.meryl")
    m.db_write_flat(flat, k, *read)
    _write_text_db(text, k, *read)
    meryl_layout.write_db(mdir, k, read[0], read[1], prefix_bits=12)
    m.db_write_flat(str(tmp_path / "a.mfxk"), k, *asm)
    separately = []
    for r in range(world):
        ix = m.Index(k, len(read[0]) + len(asm[0]) + 16)
        ix.set_shard(r, world)
        ix.load_db(flat, 0, 3, 60)
        ix.load_db(str(tmp_path / "a.mfxk"), 1)
        separately.append(ix.export())
    assert sum(len(e[0]) for e in separately) == len(np.union1d(read[0], asm[0]))
    for path in (flat, text, mdir):
        shards = []
        for r in range(world):
            ix = m.Index(k, len(read[0]) + len(asm[0]) + 16)
            ix.set_shard(r, world)
            shards.append(ix)
        m.load_db_multi(shards, path, 0, 3, 60)
        m.load_db_multi(shards, str(tmp_path / "a.mfxk"), 1)
        for ix, want in zip(shards, separately):
            got = ix.export()
            for a, b in zip(got, want):
                np.testing.assert_array_equal(a, b)
            assert ix.origin()[1:] == (3, 60)
    # tables of different k cannot share a load
    with pytest.raises(m.MfxError):
        m.load_db_multi([m.Index(k, 100), m.Index(k - 2, 100)], flat, 0)


@pytest.mark.parametrize("world,k,use_prob", [(2, 21, False), (3, 21, True), (4, 31, False), (8, 31, True)])
def test_sharded_dump_equals_whole_index(world, k, use_prob, tmp_path, golden_dir):
    """mfx_dump_values_sharded / mfx_dump_contig_sharded: every shard answers for the k-mers it owns and 0 otherwise, the value
    arrays are added, kmissing is recounted from the sums -- same arrays as the oracle's processDump (merfin-dump.C:44-67),
    same text as the unsharded -dump, for ranges that start inside a tile too"""
    import merfin_amd as m
    peak = 17.3
    contigs, read, asm = synth.world(k=k, peak=peak, seed=97, sizes=(21000, 9000, 700, 10))
    K = P = None
    if use_prob:
        K, P = po.load_kmetric(os.path.join(golden_dir, "example_lookup_table.txt"))
        read = (read[0], (read[1] * 3).astype(np.uint32))            # reach the table's rows that are not 0
    p = po.Params(k, peak, K, P)
    RL, AL = po.Lookup(k, *read), po.Lookup(k, *asm)
    shards = _build_shards(m, k, read, asm, world)
    assert all(len(s.export()[0]) > 0 for s in shards)
    evs = [m.Evaluator(s, m.KParams(peak, K, P)) for s in shards]
    seqs = m.Sequences(contigs)
    whole = m.Index(k, len(read[0]) + len(asm[0]) + 16)
    whole.add_read(*read)
    whole.add_asm(*asm)
    wev = m.Evaluator(whole, m.KParams(peak, K, P))
    for c, ctg in enumerate(contigs):
        n = len(ctg)
        rv, av, ka, km = m.dump_values_sharded(evs, [seqs] * world, c, 0, n)
        wr, wa, wka, wkm = wev.dump_values(seqs, c, 0, n)
        assert np.array_equal(rv, wr) and np.array_equal(av, wa) and (ka, km) == (wka, wkm)
        rk, ak, kmet, dk, dm = po.process_dump(p, RL, AL, ctg)
        assert (ka, km) == (dk, dm)
        npos = max(n - k + 1, 0)
        kp = m.KParams(peak, K, P)
        for i in list(range(0, npos, 97))[:200]:
            a, b, _ = m.getK(kp, int(rv[i]), int(av[i]))
            assert (a, b) == (rk[i], ak[i])
        if n > 5000:                                                     # a range that begins and ends inside tiles
            r2, a2, ka2, km2 = m.dump_values_sharded(evs, [seqs] * world, c, 4099, n - 1001)
            assert np.array_equal(r2, wr[4099:n - 1001]) and np.array_equal(a2, wa[4099:n - 1001])
            w2 = wev.dump_values(seqs, c, 4099, n - 1001)
            assert (ka2, km2) == (w2[2], w2[3])
        sp, wp = str(tmp_path / "s.dump"), str(tmp_path / "w.dump")
        assert m.dump_contig_sharded(evs, [seqs] * world, c, "ctg%d" % c, sp, append=c > 0) == (ka, km)
        wev.dump_contig(seqs, c, "ctg%d" % c, wp, append=c > 0)
    assert open(sp, "rb").read() == open(wp, "rb").read() and os.path.getsize(sp) > 100000
    with pytest.raises(m.MfxError):                          # the slots must be shard 0..N-1 in order
        m.dump_values_sharded(evs[::-1], [seqs] * world, 0, 0, 10)
    with pytest.raises(m.MfxError):
        m.dump_values_sharded([wev], [seqs], 0, 0, len(contigs[0]) + 1)


@pytest.mark.parametrize("world,mode,k,seed", [(2, "polish", 21, 71), (3, "filter", 21, 72), (4, "loose", 31, 73), (2, "strict", 15, 74), (5, "better", 21, 75)])
def test_sharded_variant_modes(world, mode, k, seed, tmp_path):
    """mfx_variants_run_sharded: the variant modes with the path k-mers looked up in all shards of the index -- VCF and
    -debug text byte-identical to the oracle's restatement of merfin-variants.C / varMer.C"""
    import merfin_amd as m
    peak = 17.3
    names, asm, vcf, read, amers = synth.variant_world(k=k, peak=peak, seed=seed)
    vp = str(tmp_path / "in.vcf")
    open(vp, "w").write(vcf)
    p = po.Params(k, peak)
    R, A = po.Lookup(k, *read), po.Lookup(k, *amers)
    n_o = po.variants_run(p, R, A, mode, vp, names, asm, str(tmp_path / "o.vcf"), comb=9, debug_path=str(tmp_path / "o.dbg"), log_path=str(tmp_path / "o.log"))
    shards = _build_shards(m, k, read, amers, world)
    evs = [m.Evaluator(s, m.KParams(peak)) for s in shards]
    n_g = m.variants_sharded(evs, mode, vp, names, asm, str(tmp_path / "g.vcf"), comb=9, debug_path=str(tmp_path / "g.dbg"), log_path=str(tmp_path / "g.log"))
    assert n_g == n_o and n_o > 20
    assert open(tmp_path / "g.vcf").read() == open(tmp_path / "o.vcf").read()
    assert open(tmp_path / "g.dbg").read() == open(tmp_path / "o.dbg").read()
    assert len([l for l in open(tmp_path / "g.vcf") if not l.startswith("#")]) > 20


def test_cli_sharded_dump_and_variants(tmp_path, golden_dir):
    """`merfin -dump|-polish|-filter|-loose ... -devices 0,0,0 -sharded`: the committed golden outputs of the unsharded run"""
    import subprocess
    exe = os.path.join(ROOT, "merfin_amd", "bin", "merfin")
    g = lambda n: os.path.join(golden_dir, n)
    base = ["-sequence", g("case1.fasta"), "-readmers", g("case1.read.kmers.txt"), "-peak", "17.3", "-prob", g("example_lookup_table.txt")]
    common = base + ["-devices", "0,0,0", "-sharded"]
    r = subprocess.run([exe, "-dump"] + common + ["-output", str(tmp_path / "s.dump")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert (tmp_path / "s.dump").read_bytes() == open(g("case1.dump"), "rb").read()
    one = subprocess.run([exe, "-dump", "-skipMissing"] + base + ["-output", str(tmp_path / "n1.dump")], capture_output=True, text=True)
    r2 = subprocess.run([exe, "-dump", "-skipMissing"] + common + ["-output", str(tmp_path / "n.dump")], capture_output=True, text=True)
    assert one.returncode == 0 and r2.returncode == 0 and not (tmp_path / "n.dump").exists(), r2.stderr
    counts = lambda t: [l for l in t.splitlines() if l.count("\t") == 3 and not l.startswith("--")]
    assert counts(r2.stderr) == counts(one.stderr) and len(counts(one.stderr)) > 1
    assert counts(r.stderr) == counts(one.stderr)                        # the per-contig missing / cumulative columns of -dump
    assert "one part of the sequences each" in r.stderr
    rf = subprocess.run([exe, "-dump"] + common + ["-output", str(tmp_path / "sf.dump")], capture_output=True, text=True, env=dict(os.environ, MFX_CLI_FULL_INDEX="1"))
    assert rf.returncode == 0 and "sharded index" in rf.stderr, rf.stderr     # the same from the hash-sharded full tables
    assert (tmp_path / "sf.dump").read_bytes() == open(g("case1.dump"), "rb").read()
    for mode, suffix in (("polish", ".polish.vcf"), ("filter", ".filter.vcf"), ("loose", ".filter.vcf")):
        out = str(tmp_path / ("v_" + mode))
        rv = subprocess.run([exe, "-" + mode] + common + ["-vcf", g("case1.vcf"), "-comb", "8", "-output", out], capture_output=True, text=True)
        assert rv.returncode == 0, rv.stderr
        assert open(out + suffix, "rb").read() == open(g("case1.%s.vcf" % mode), "rb").read()
