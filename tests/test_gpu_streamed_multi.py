"""The two C++ host paths added around the -hist kernel:
  * streamed evaluation (mfx_hist_run_streamed): host buffers -> chunked H2D overlapped with the kernel -> histogram on
    the host, SURVEY 8(d)'s timed region; must equal upload-then-evaluate bit for bit (koverCpy included);
  * several devices driven by ONE process (mfx_index_replicate / mfx_seq_replicate / mfx_hist_run_multi) -- the GPU
    box has one GPU, so the slots are several contexts on device 0 ("-devices 0,0"); results vs the CPU oracle,
    including K* bins beyond the dense device image (the reference's arrays are unbounded, merfin-histogram.C:74,87);
  * the RCCL communicator of the one-process-per-GPU path (csrc/mfx_comm.cpp) with a world of ONE rank (RCCL refuses
    two ranks per device): the collective calls run, the result is the identity."""
import numpy as np
import pytest

from oracle import pyoracle as po
from tests import synth
from tests.test_gpu_parity import assert_hist_equal, build_index, oracle_hist

pytestmark = pytest.mark.gpu


def _same(a, b):
    assert (a.kasm, a.kmissing) == (b.kasm, b.kmissing)
    np.testing.assert_array_equal(a.undr(), b.undr())
    np.testing.assert_array_equal(a.over(), b.over())
    np.testing.assert_array_equal(a.contig_kasm(), b.contig_kasm())
    np.testing.assert_array_equal(a.contig_kmissing(), b.contig_kmissing())
    assert a.koverCpy == b.koverCpy                      # bit-identical: same values, same summation tree


@pytest.mark.parametrize("transport", ["packed", "ascii", "link_bytes"])
@pytest.mark.parametrize("shape", ["few_large", "many_small", "edges"])
def test_streamed_equals_resident_and_oracle(shape, transport, monkeypatch):
    """transport: "packed" = host-packed planes (2-bit codes + validity bits over PCIe, the kernel reads packed tiles) forced,
    "ascii" = MFX_STREAM_ASCII=1 (one byte per base, the kernel encodes its tiles; what k > 31 uses), "link_bytes" = what a rank of
    many takes when its share of the host's threads encodes slower than its link moves plain bytes (MFX_STREAM_TRANSPORT=ascii forces
    it): pinned sources cross by DMA as they are and mfx_pack_kernel makes the planes on the device (pageable sources stay host-packed)"""
    import merfin_amd as m
    if transport == "ascii":
        monkeypatch.setenv("MFX_STREAM_ASCII", "1")
    monkeypatch.setenv("MFX_STREAM_TRANSPORT", "ascii" if transport == "link_bytes" else "pack")
    k, peak = 21, 17.3
    r = synth.rng(311)
    if shape == "few_large":
        contigs, read, asm = synth.world(k=k, peak=peak, seed=312, sizes=(300000, 70000, 4096, 4097, 20, 0, 8191))
    elif shape == "many_small":                               # > 64 pieces per chunk: the assembled-copy path
        contigs = [synth.random_contig(r, int(n)).tobytes() for n in r.integers(0, 900, size=3000)]
        contigs[5] = b""
        ak, av = po.count_kmers(k, contigs)
        read, asm = (ak, (2 + (ak % 30)).astype(np.uint32)), (ak, av)
    else:
        base = synth.random_contig(r, 3 * 4096 + 50)
        base[4095] = ord("N"); base[4096 + 20] = ord("n")
        contigs = [b"", b"ACGT", base.tobytes(), synth.random_contig(r, 4096).tobytes(), b"N" * 5000,
                   synth.random_contig(r, 21).tobytes(), synth.random_contig(r, 4096 + 20).tobytes(), b""]
        ak, av = po.count_kmers(k, contigs)
        read, asm = (ak, (3 + (ak % 40)).astype(np.uint32)), (ak, av)
    p, g, ka, km = oracle_hist(k, peak, contigs, read, asm)
    ix = build_index(m, k, read, asm)
    ev = m.Evaluator(ix, m.KParams(peak))
    resident = ev.hist(m.Sequences(contigs))
    assert_hist_equal(resident, g, ka, km, k)
    lens = [len(c) for c in contigs]
    # (a) pageable host buffers: staged through pinned memory by the library
    s1 = m.Sequences.create(lens)
    _same(ev.hist_streamed(s1, contigs), resident)
    _same(ev.hist(s1), resident)                              # afterwards the sequence object is a normal resident one
    # ... also for the kernels that read one byte per base (after a packed upload they unpack the planes first)
    s0 = m.Sequences(contigs)
    for c in [i for i, n in enumerate(lens) if n > 0][:4]:
        a, b = ev.dump_values(s1, c, 0, lens[c]), ev.dump_values(s0, c, 0, lens[c])
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[2:] == b[2:]
    c1, c0 = m.Index(k, len(asm[0]) + 16), m.Index(k, len(asm[0]) + 16)
    c1.count_asm(s1)
    c0.count_asm(s0)
    e1, e0 = c1.export(), c0.export()
    assert all(np.array_equal(x, y) for x, y in zip(e1, e0)) and np.array_equal(e0[0], asm[0]) and np.array_equal(e0[2], asm[1])
    # (b) pinned buffers (mfx_host_alloc), DMA'd in place
    pins = [m.PinnedBuffer(n) for n in lens]
    for pb, c in zip(pins, contigs):
        pb.array[:] = np.frombuffer(c, dtype=np.uint8)
    s2 = m.Sequences.create(lens)
    _same(ev.hist_streamed(s2, [pb.array for pb in pins]), resident)
    # (c) a second run on the same objects (buffers re-armed)
    _same(ev.hist_streamed(s2, [pb.array for pb in pins]), resident)


def test_streamed_spans_several_chunks():
    """more tiles than one 16384-tile chunk: chunk boundaries inside and between contigs, halo bytes re-sent"""
    import merfin_amd as m
    k, peak = 21, 9.0
    r = synth.rng(313)
    sizes = (16384 * 4096 + 777, 3 * 4096 * 4096 + 5, 1000)   # 67 M + 50 M bases: 3 chunks
    contigs = [synth.random_contig(r, n) for n in sizes]
    contigs[0][16384 * 4096 - 10:16384 * 4096 + 10] |= 0x20   # lower case across the first chunk boundary
    contigs[1][12345] = ord("N")
    seqs = m.Sequences([c.tobytes() for c in contigs])
    ix = m.Index(k, sum(sizes) + 1024)
    ix.count_asm(seqs)
    # read counts for the k-mers around both chunk boundaries and at the contig ends (the rest stay "missing")
    b1 = 16384 * 4096
    spots = [contigs[0][:200000], contigs[0][b1 - 100000:b1 + 100000], contigs[0][-50000:], contigs[1][:50000],
             contigs[1][12000:13000], contigs[1][-50000:], contigs[2]]
    rk = po.count_kmers(k, [x.tobytes() for x in spots])[0]
    ix.add_read(rk, (1 + (rk % 50)).astype(np.uint32))
    ev = m.Evaluator(ix, m.KParams(peak))
    resident = ev.hist(seqs)
    assert resident.kasm == sum(n - 20 for n in sizes) - 21    # one N: 21 k-mers lost
    assert 0 < resident.kmissing < resident.kasm and int(resident.undr().sum() + resident.over().sum()) > 400000
    s = m.Sequences.create(list(sizes))
    _same(ev.hist_streamed(s, contigs), resident)
    pins = [m.PinnedBuffer(n) for n in sizes]
    for pb, c in zip(pins, contigs):
        pb.array[:] = c
    _same(ev.hist_streamed(m.Sequences.create(list(sizes)), [pb.array for pb in pins]), resident)


def test_streamed_invalid_bases_just_behind_a_chunk_cut():
    """An N, a lower-case run and a contig end 1..k-1 bases behind the cut between two chunks of the streamed run: the k-mers of the
    earlier chunk's last tile run over them, and that chunk's kernel may still be reading the validity words of its halo while the next
    chunk's upload arrives (the sparse validity transport fills a chunk's range with ones before it scatters the exceptions: the halo
    words are left out of that fill).  Chunks are cut at 2048, 6144, 14336 tiles (8, 16, 32 MB of bases: CH0 doubling)."""
    import merfin_amd as m
    k, peak = 21, 9.0
    r = synth.rng(317)
    T0, T1, T2 = 2048, 6144, 14336
    sizes = (T1 * 4096 + 5, (T2 - T1 - 1) * 4096 + 4096 * 700 + 11)   # contig 0 ends 5 bases behind the second cut
    contigs = [synth.random_contig(r, n) for n in sizes]
    contigs[0][T0 * 4096 + 1] = ord("N")                      # 1 base behind the first cut
    at2 = (T2 - T1 - 1) * 4096                                # the third cut, in contig 1's coordinates
    contigs[1][at2 + 19] = ord("n")                           # k-2 bases behind it
    contigs[1][at2 + 4096 * 300 + 7:at2 + 4096 * 300 + 9] = ord("-")
    seqs = m.Sequences([c.tobytes() for c in contigs])
    ix = m.Index(k, sum(sizes) + 1024)
    ix.count_asm(seqs)
    spots = [contigs[0][T0 * 4096 - 5000:T0 * 4096 + 5000], contigs[0][-5000:], contigs[1][:5000], contigs[1][at2 - 5000:at2 + 5000]]
    rk = po.count_kmers(k, [x.tobytes() for x in spots])[0]
    ix.add_read(rk, (1 + (rk % 50)).astype(np.uint32))
    ev = m.Evaluator(ix, m.KParams(peak))
    resident = ev.hist(seqs)
    assert resident.kasm == sum(n - 20 for n in sizes) - 21 - 21 - 22
    s = m.Sequences.create(list(sizes))
    for _ in range(6):                                        # the race this guards against was a matter of timing
        _same(ev.hist_streamed(s, contigs), resident)
    # the plain-bytes transport (pinned sources, planes made on the device) over the same cuts: a chunk's first words are the halo
    # words the chunk before may still be reading -- rewritten with the values they have
    pins = [m.PinnedBuffer(n) for n in sizes]
    for pb, c in zip(pins, contigs):
        pb.array[:] = c
    import os
    os.environ["MFX_STREAM_TRANSPORT"] = "ascii"
    try:
        s3 = m.Sequences.create(list(sizes))
        for _ in range(4):
            _same(ev.hist_streamed(s3, [pb.array for pb in pins]), resident)
    finally:
        os.environ.pop("MFX_STREAM_TRANSPORT", None)


@pytest.fixture(scope="module")
def five_mb_world():
    k, peak = 21, 17.3
    contigs, read, asm = synth.world(k=k, peak=peak, seed=321, sizes=(4200000, 1100000, 4097, 500, 0, 8191), err_kmers=3000)
    p, g, ka, km = oracle_hist(k, peak, contigs, read, asm)
    return k, peak, contigs, read, asm, g, ka, km


@pytest.mark.parametrize("transport", ["pack", "ascii", "auto"])
@pytest.mark.parametrize("nslots", [2, 3, 8])
def test_streamed_over_several_slots_equals_resident_bit_for_bit(nslots, transport, monkeypatch):
    """SURVEY 8(d)'s evaluate phase over N devices (mfx_hist_run_streamed_multi; the slots are N contexts on device 0 here, distinct
    devices in tests/test_gpu_multidevice.py): every slot uploads and evaluates only its contiguous share of the tiles; bins, counters,
    per-contig counters AND koverCpy equal the single resident launch to the last bit (shares are cut at multiples of 1024 tiles, the
    host adds the first-level sums in the device's order)."""
    import merfin_amd as m
    if transport != "auto":
        monkeypatch.setenv("MFX_STREAM_TRANSPORT", transport)     # both transports (auto: chosen from the measured rates), same bits
    k, peak = 21, 9.0
    r = synth.rng(331)
    sizes = (2300 * 4096 + 77, 5, 700 * 4096 + 4095, 0, 260 * 4096)      # 3263 tiles: cuts at 1024 / 2048 / 3072 for 3 slots
    contigs = [synth.random_contig(r, n) for n in sizes]
    contigs[0][1024 * 4096 + 3] = ord("N")                    # just behind a cut between two slots
    contigs[0][2048 * 4096 - 2:2048 * 4096 + 30] |= 0x20
    contigs[2][100] = ord("N")
    for c in contigs:                                          # duplicated stretches: asmK > readK somewhere, so koverCpy is a real sum
        if len(c) > 300000:
            c[200000:260000] = c[100000:160000]
    seqs = m.Sequences([c.tobytes() for c in contigs])
    ix = m.Index(k, sum(sizes) + 1024)
    ix.count_asm(seqs)
    spots = [contigs[0][:400000], contigs[0][1024 * 4096 - 50000:1024 * 4096 + 50000], contigs[0][2048 * 4096 - 50000:2048 * 4096 + 50000],
             contigs[0][-50000:], contigs[1], contigs[2][:300000], contigs[4][-100000:]]
    rk = po.count_kmers(k, [x.tobytes() for x in spots])[0]
    ix.add_read(rk, (1 + (rk % 7)).astype(np.uint32))
    ev = m.Evaluator(ix, m.KParams(peak))
    resident = ev.hist(seqs)
    assert resident.koverCpy > 0 and 0 < resident.kmissing < resident.kasm
    evs = [m.Evaluator(ix, m.KParams(peak)) for _ in range(nslots)]
    sqs = [m.Sequences.create(list(sizes)) for _ in range(nslots)]
    pins = [m.PinnedBuffer(n) for n in sizes]                  # pinned sources: the only ones the plain-bytes transport takes
    for pb, c in zip(pins, contigs):
        pb.array[:] = c
    contigs = [pb.array for pb in pins]
    for rep in range(2):                                       # the second run re-arms every slot's buffers
        _same(m.hist_streamed_multi(evs, sqs, contigs), resident)
    # a slot's sequence object holds its part only: whole-sequence calls are refused until something is uploaded whole
    with pytest.raises(m.MfxError, match="holds only the tiles"):
        evs[0].hist(sqs[0])
    with pytest.raises(m.MfxError, match="share an evaluator or a sequence"):
        m.hist_streamed_multi(evs[:2], [sqs[0], sqs[0]], contigs)
    _same(evs[0].hist_streamed(sqs[0], contigs), resident)    # whole again
    _same(evs[0].hist(sqs[0]), resident)
    # one rank's share, added into the caller's device image (what bench.py's ranks do before their all-reduce)
    import torch
    words = m.hist_words(ev.nbins, seqs.ncontigs)
    counts = torch.zeros(words, dtype=torch.int64, device="cuda")
    kover = torch.zeros(1, dtype=torch.float64, device="cuda")
    for rank in range(nslots):
        lo, hi = m.stream_share(seqs.ntiles, rank, nslots)
        evs[rank].hist_streamed_range(sqs[rank], contigs, lo, hi, counts, kover)
    torch.cuda.synchronize()
    got = ev.result_from_counts(counts.cpu().numpy().view(np.uint64), float(kover.item()), seqs.ncontigs)
    assert (got.kasm, got.kmissing) == (resident.kasm, resident.kmissing)
    np.testing.assert_array_equal(got.undr(), resident.undr())
    np.testing.assert_array_equal(got.over(), resident.over())
    np.testing.assert_array_equal(got.contig_kasm(), resident.contig_kasm())
    assert abs(got.koverCpy - resident.koverCpy) <= 1e-12 * abs(resident.koverCpy)


def test_streamed_over_several_slots_matches_oracle(five_mb_world):
    import merfin_amd as m
    k, peak, contigs, read, asm, g, ka, km = five_mb_world
    ix = build_index(m, k, read, asm)
    lens = [len(c) for c in contigs]
    for n in (2, 4):
        evs = [m.Evaluator(ix, m.KParams(peak)) for _ in range(n)]
        res = m.hist_streamed_multi(evs, [m.Sequences.create(lens) for _ in range(n)], contigs)
        assert_hist_equal(res, g, ka, km, k)


@pytest.mark.parametrize("nslots", [2, 3, 5])
def test_one_process_several_slots_matches_oracle(nslots, five_mb_world):
    import merfin_amd as m
    k, peak, contigs, read, asm, g, ka, km = five_mb_world
    ix = build_index(m, k, read, asm)
    seqs = m.Sequences(contigs)
    assert seqs.ntiles > 256 * nslots                         # every slot gets several 256-tile blocks
    # slot 0 uses the original objects, the others replicas made by device-to-device copy (no second build)
    ixs = [ix] + [ix.replicate(0) for _ in range(nslots - 1)]
    sqs = [seqs] + [seqs.replicate(0) for _ in range(nslots - 1)]
    for rep in ixs[1:]:
        assert rep.info() == ix.info()
    evs = [m.Evaluator(x, m.KParams(peak)) for x in ixs]
    res = m.hist_multi(evs, sqs)
    assert_hist_equal(res, g, ka, km, k)
    single = evs[0].hist(seqs)
    assert (res.kasm, res.kmissing) == (single.kasm, single.kmissing)
    again = m.hist_multi(evs, sqs)
    assert again.koverCpy == res.koverCpy                     # fixed-order sum: bit-stable run to run
    # slots may also share one index and one sequence object (only the evaluators must be distinct)
    shared = m.hist_multi([m.Evaluator(ix, m.KParams(peak)) for _ in range(nslots)], [seqs] * nslots)
    assert_hist_equal(shared, g, ka, km, k)
    with pytest.raises(m.MfxError):
        m.hist_multi([evs[0], evs[0]], [seqs, seqs])


def _overflow_world(m):
    """asmK/readK ratios > 13107 (K* bin >= 65536: beyond the default dense image) on both halves of the tile range"""
    k = 21
    r = synth.rng(331)
    contigs = [synth.random_contig(r, 300 * 4096).tobytes(), synth.random_contig(r, 300 * 4096 + 11).tobytes()]
    ak, av = po.count_kmers(k, contigs)
    av = av.copy()
    rv = np.full(len(ak), 5, dtype=np.uint32)
    first = po.count_kmers(k, [contigs[0][:60], contigs[0][-60:], contigs[1][:60], contigs[1][-60:]])[0]
    hot = np.isin(ak, first)
    av[hot] = 70000 + (np.arange(hot.sum()) % 7) * 100000     # undr bins 349 995 ... 3.3 M
    rv[np.isin(ak, po.count_kmers(k, [contigs[1][5000:5040]])[0])] = 900000   # over bins ~ 900 000
    return k, 5.0, contigs, (ak, rv), (ak, av)


def test_overflow_bins_survive_the_multi_slot_reduction():
    import merfin_amd as m
    k, peak, contigs, read, asm = _overflow_world(m)
    p, g, ka, km = oracle_hist(k, peak, contigs, read, asm)
    assert g.c.undrMax > 65536 and g.c.overMax > 65536
    ix = build_index(m, k, read, asm)
    seqs = m.Sequences(contigs)
    assert_hist_equal(m.Evaluator(ix, m.KParams(peak)).hist(seqs), g, ka, km, k)
    for n in (2, 3):
        evs = [m.Evaluator(ix, m.KParams(peak)) for _ in range(n)]
        assert_hist_equal(m.hist_multi(evs, [seqs] * n), g, ka, km, k)
    # the streamed path folds them too
    assert_hist_equal(m.Evaluator(ix, m.KParams(peak)).hist_streamed(m.Sequences.create([len(c) for c in contigs]), contigs), g, ka, km, k)


def test_rccl_communicator_world_of_one():
    """the C path's collective with nranks = 1: ncclCommInitRank / ncclAllReduce(uint64) / ncclAllGather run on the
    device; one rank's image is already the global one, so every word must come back unchanged"""
    torch = pytest.importorskip("torch")
    import merfin_amd as m
    k, peak, contigs, read, asm = _overflow_world(m)
    p, g, ka, km = oracle_hist(k, peak, contigs, read, asm)
    ix = build_index(m, k, read, asm)
    seqs = m.Sequences(contigs)
    ev = m.Evaluator(ix, m.KParams(peak))
    comm = m.Comm(m.Comm.unique_id(), 0, 1, device=0)
    comm.barrier()
    counts = torch.zeros(m.hist_words(ev.nbins, seqs.ncontigs), dtype=torch.int64, device="cuda")
    kover = torch.zeros(1, dtype=torch.float64, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    ev.take_overflow()                                        # start from an empty list
    ev.hist_launch_cyclic(seqs, 0, 1, counts, kover, stream=s)
    torch.cuda.synchronize()
    before, kb = counts.clone(), float(kover.item())
    comm.hist_allreduce(ev, counts, kover, seqs.ncontigs, stream=s)
    torch.cuda.synchronize()
    assert torch.equal(before, counts) and float(kover.item()) == kb
    h = counts.cpu().numpy().view(np.uint64)
    novf = int(h[2 * ev.nbins + 2])
    assert novf > 0
    rec = comm.allgather_overflow(ev, stream=s)
    assert int(rec[:, 1].sum()) == novf                        # {far bin, occurrences} pairs: the occurrences are the image's novf word
    res = ev.result_from_counts(h, kb, seqs.ncontigs).add_overflow(rec)
    assert_hist_equal(res, g, ka, km, k)
    assert len(ev.take_overflow()) == 0                       # the gather consumed the list
    # the sharded index's exchange with nranks = 1: everything goes to (and comes from) this rank
    send = torch.arange(1000, dtype=torch.int64, device="cuda")
    recv = torch.zeros(1000, dtype=torch.int64, device="cuda")
    rc = comm.exchange_counts([777], stream=s)
    assert list(rc) == [777]
    comm.alltoallv(send, [777], recv, rc, 8, stream=s)
    torch.cuda.synchronize()
    assert torch.equal(recv[:777], send[:777]) and int(recv[777:].abs().sum()) == 0
    comm.close()


@pytest.mark.parametrize("seed", list(range(int(__import__("os").environ.get("MFX_RANDOM_SEEDS", "12")))))
def test_randomized_host_paths_agree(seed, monkeypatch):
    """Every way of getting the same -hist -- one resident launch, streamed packed, streamed one byte per base, N slots of one
    process over replicas, N shards of one process with the k-mers routed to their owners -- on seeded random worlds (k, peak,
    contig shapes with empty / tiny / N-riddled contigs, big read counts beyond the dense image, table fill, chunking):
    the oracle's result from each, and bit-identical koverCpy between the paths that sum the same per-tile values."""
    import merfin_amd as m
    from tests.test_gpu_sharded import _build_shards
    r = np.random.default_rng(7000 + seed)
    k = int(r.choice([9, 15, 21, 27, 31]))
    peak = float(r.choice([2.5, 9.0, 17.3, 26.0]))
    monkeypatch.setenv("MFX_LOAD_FACTOR", str(r.choice([0.4, 0.6, 0.8])))
    sizes = tuple(int(x) for x in r.choice([0, 5, k, 37, 4095, 4096, 4097, 20000, 70000, 300000], size=int(r.integers(1, 10))))
    contigs, read, asm = synth.world(k=k, peak=peak, seed=8000 + seed, sizes=sizes, err_kmers=int(r.integers(0, 4000)) if k > 9 else 0)
    rv = read[1].astype(np.uint64)
    if len(rv):
        big = r.random(len(rv)) < 0.01
        rv[big] = r.choice([1, 1024, 65535, 300000], size=int(big.sum()))
    read = (read[0], rv.astype(np.uint32))
    p, g, ka, km = oracle_hist(k, peak, contigs, read, asm)
    ix = build_index(m, k, read, asm)
    kp = m.KParams(peak)
    ev = m.Evaluator(ix, kp)
    seqs = m.Sequences(contigs)
    resident = ev.hist(seqs)
    assert_hist_equal(resident, g, ka, km, k)
    lens = [len(c) for c in contigs]
    _same(ev.hist_streamed(m.Sequences.create(lens), contigs), resident)
    monkeypatch.setenv("MFX_STREAM_ASCII", "1")
    _same(ev.hist_streamed(m.Sequences.create(lens), contigs), resident)
    monkeypatch.delenv("MFX_STREAM_ASCII")
    n = int(r.integers(2, 6))
    multi = m.hist_multi([m.Evaluator(ix, kp) for _ in range(n)], [seqs] * n)
    assert_hist_equal(multi, g, ka, km, k)
    if k % 2 == 1 and len(asm[0]):
        w = int(r.integers(2, 9))
        shards = _build_shards(m, k, read, asm, w)
        evs = [m.Evaluator(s, kp) for s in shards]
        per = max(1, min(int(r.choice([1, 3, 1000])), seqs.ntiles))        # tiles routed per round
        routers = [m.Router(s, w, per) for s in shards]
        assert_hist_equal(m.hist_sharded(evs, routers, [seqs] * w), g, ka, km, k)
        for c in [i for i, L in enumerate(lens) if L > 0][:2]:
            a, b = m.dump_values_sharded(evs, [seqs] * w, c, 0, lens[c]), ev.dump_values(seqs, c, 0, lens[c])
            assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[2:] == b[2:]
