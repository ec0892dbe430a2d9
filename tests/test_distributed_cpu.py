"""N>1 path on CPU (gloo, world_size 2): tile sharding covers every k-mer once,
the counts image survives the all-reduce, and the reduced result equals the
single-process oracle result.  Each rank fabricates its partial image with the
ORACLE over its tile shard (the HIP kernel needs a GPU; the decomposition,
image layout, collective and result assembly are what is under test here)."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, tmp):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from oracle import pyoracle as po
    from tests import synth
    import merfin_amd as m
    from merfin_amd import distributed as D

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    k, peak, nbins = 21, 17.3, 2048
    contigs, read, asm = synth.world(k=k, peak=peak, seed=31, sizes=(30000, 9000, 4096, 4097, 500, 20, 0, 8191))
    p = po.Params(k, peak)
    R, A = po.Lookup(k, *read), po.Lookup(k, *asm)
    tiles = D.tile_table([len(c) for c in contigs])
    lo, hi = D.shard(len(tiles), rank, world)
    nc = len(contigs)
    undr = np.zeros(nbins, dtype=np.uint64)
    over = np.zeros(nbins, dtype=np.uint64)
    ka, km = np.zeros(nc, dtype=np.uint64), np.zeros(nc, dtype=np.uint64)
    kover = 0.0
    for c, p0, n in tiles[lo:hi]:
        # k-mers STARTING in [p0, p0+n): bases up to p0+n+k-2
        h = po.process_histogram(p, R, A, contigs[c][p0:p0 + n + k - 1])
        u, o = h.undr(), h.over()
        undr[:len(u)] += u
        over[:len(o)] += o
        ka[c] += h.kasm
        km[c] += h.kmissing
        kover += h.koverCpy
    img = D.pack_counts(nbins, nc, undr, over, int(ka.sum()), int(km.sum()), ka, km)
    counts = torch.from_numpy(img.view(np.int64).copy())
    kov = torch.tensor([kover], dtype=torch.float64)
    D.all_reduce_hist(counts, kov)
    res = D.reduced_result(nbins, nc, counts, kov)
    if rank == 0:
        g, gka, gkm, _ = po.hist_run(p, R, A, contigs, threads=2)
        ok = (res.kasm == g.kasm and res.kmissing == g.kmissing
              and (res.undr()[:nbins] == np.pad(g.undr(), (0, nbins))[:nbins]).all()
              and (res.over()[:nbins] == np.pad(g.over(), (0, nbins))[:nbins]).all()
              and (res.contig_kasm() == gka).all() and (res.contig_kmissing() == gkm).all()
              and abs(res.koverCpy - g.koverCpy) <= 1e-9 * max(1.0, g.koverCpy) and g.kasm > 50000)
        open(os.path.join(tmp, "ok"), "w").write("1" if ok else "0 %r %r" % ((res.kasm, res.kmissing), (g.kasm, g.kmissing)))
    dist.barrier()
    dist.destroy_process_group()


def _completeness_worker(rank, world, port, tmp):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from oracle import pyoracle as po
    from tests import synth
    from merfin_amd import distributed as D

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    k, peak = 21, 17.3
    contigs, read, asm = synth.world(k=k, peak=peak, seed=37)
    p = po.Params(k, peak)

    def pieces(rk, rv, ak, av):
        t64, u64 = np.zeros(64), np.zeros(64)
        for piece in range(64):
            lo, hi = piece << (2 * k - 6), (piece + 1) << (2 * k - 6)
            rs, as_ = (rk >= lo) & (rk < hi), (ak >= lo) & (ak < hi)
            t64[piece], u64[piece] = po.completeness_piece(p, rk[rs], rv[rs], ak[as_], av[as_])
        return t64, u64

    # any partition of the K-MER SPACE works (the sharded index uses a hash of the minimizer): here a bit mix
    own = lambda keys: ((keys * np.uint64(0x9E3779B97F4A7C15)) >> np.uint64(40)) % np.uint64(world) == np.uint64(rank)
    mr, ma = own(read[0]), own(asm[0])
    t64, u64 = pieces(read[0][mr], read[1][mr], asm[0][ma], asm[1][ma])
    total, undr, rt, ru = D.reduce_completeness(t64, u64)
    if rank == 0:
        wt, wu = pieces(read[0], read[1], asm[0], asm[1])
        tt = uu = 0.0
        for i in range(64):
            tt += wt[i]
            uu += wu[i]
        ok = (rt == wt).all() and (ru == wu).all() and (total, undr) == (tt, uu) and tt > 0 and uu > 0 and 0 < mr.sum() < len(mr)
        open(os.path.join(tmp, "okc"), "w").write("1" if ok else "0 %r %r" % ((total, undr), (tt, uu)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_completeness(tmp_path):
    """-completeness over a sharded index: per-piece sums all-reduced, then added in piece order"""
    pytest.importorskip("torch")
    import torch.multiprocessing as mp
    port = 31500 + os.getpid() % 2000
    mp.spawn(_completeness_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "okc").read_text() == "1"


def test_two_rank_gloo_reduction(tmp_path):
    torch = pytest.importorskip("torch")
    import torch.multiprocessing as mp
    port = 29500 + os.getpid() % 2000
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "ok").read_text() == "1"


def test_shard_partition_properties():
    sys.path.insert(0, ROOT)
    from merfin_amd import distributed as D
    for T in (0, 1, 7, 732436):
        for world in (1, 2, 3, 4, 8):
            r = [D.shard(T, i, world) for i in range(world)]
            assert r[0][0] == 0 and r[-1][1] == T
            assert all(a[1] == b[0] for a, b in zip(r, r[1:]))
            assert max(h - l for l, h in r) - min(h - l for l, h in r) <= 1
    tt = D.tile_table([0, 5, 4096, 4097, 10000])
    assert [t for t in tt if t[0] == 3] == [(3, 0, 4096), (3, 4096, 1)]
    assert sum(n for _, _, n in tt) == 5 + 4096 + 4097 + 10000


def _oob_worker(rank, world, port, tmp):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import struct
    import bench
    o = bench.Oob()
    assert (o.rank, o.world) == (rank, world)
    uid = o.bcast(b"\x07" * 128 if rank == 0 else b"")             # the RCCL unique id travels like this
    assert uid == b"\x07" * 128
    for _ in range(20):
        o.barrier()
    parts = o.allgather(struct.pack("<QQ", rank, rank * rank))      # the rehearsal's host-side reduction uses this
    assert [struct.unpack("<QQ", b) for b in parts] == [(r, r * r) for r in range(world)]
    assert o.max(1.0 + rank) == float(world)                         # max-over-ranks of the elapsed time
    o.barrier()
    open(os.path.join(tmp, "ok%d" % rank), "w").write("1")


def test_bench_launcher_channel_three_ranks(tmp_path):
    """bench.py's out-of-band channel for N > 1 (torchrun's rendezvous store: RCCL unique id, barriers, max of the
    elapsed time) with three real processes on CPU.  The data-path collective itself (csrc/mfx_comm.cpp) needs GPUs."""
    torch = pytest.importorskip("torch")
    import torch.multiprocessing as mp
    port = 29300 + os.getpid() % 500
    mp.spawn(_oob_worker, args=(3, port, str(tmp_path)), nprocs=3, join=True)
    assert all((tmp_path / ("ok%d" % r)).exists() for r in range(3))


def _bench(args, env=None, timeout=300):
    e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, cwd=ROOT, capture_output=True, text=True, env=e, timeout=timeout)


def test_bench_refuses_to_mislabel_its_gpu_count():
    """`python bench.py --gpus N` owns its ranks (as merfin.C:366-414 owns its workers): without a launcher it starts N of them
    itself, and where it cannot -- fewer than N devices (none here), or a launcher whose WORLD_SIZE disagrees -- it prints ONE JSON line
    with value null and exits non-zero instead of measuring another GPU count than the one it reports."""
    import json
    r = _bench(["--gpus", "2", "--steps", "1", "--warmup", "0"], env={"HIP_VISIBLE_DEVICES": "", "ROCR_VISIBLE_DEVICES": ""})
    assert r.returncode != 0
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["value"] is None and d["n_gpus"] == 2 and "HIP device(s) visible" in d["error"]
    r = _bench(["--gpus", "2"], env={"WORLD_SIZE": "4", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert d["value"] is None and "WORLD_SIZE=4" in d["error"]
