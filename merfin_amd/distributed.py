"""Multi-GPU decomposition of -hist: one process per GPU, tiles sharded, the
index replicated, ONE all-reduce of the counts image (RCCL over xGMI when the
backend is "nccl"; the CPU tests use gloo).  Contigs are independent work
items in the reference too (merfin.C:408-410; scripts/parallel1/merfin.sh:68-85
shards contigs across processes and concatenates)."""
import numpy as np

from .binding import TILE, hist_words, result_from_counts


def tile_table(lens):
    """[(contig, first start position, n positions)] in global tile order --
    the same layout mfx_seq builds (csrc/mfx_api.cpp:seq_layout)."""
    out = []
    for c, n in enumerate(lens):
        for p in range(0, n, TILE):
            out.append((c, p, min(TILE, n - p)))
    return out


def shard(ntiles, rank, world):
    """tile range [lo, hi) of `rank`: contiguous, disjoint, covering"""
    return ntiles * rank // world, ntiles * (rank + 1) // world


def contig_partition(weights, world):
    """Contiguous split of the contigs (input order) into `world` runs of about equal total weight: [(lo, hi)] per
    rank.  Used where the output is per contig and ordered (-dump text, VCF records): every rank writes its run,
    the parts are concatenated in rank order -- what the reference's SLURM array does by hand
    (scripts/parallel1/merfin.sh:68-85)."""
    w = np.asarray(weights, dtype=np.float64)
    n = len(w)
    cum = np.concatenate([[0.0], np.cumsum(w)])
    tot = cum[-1]
    cuts = [0]
    for r in range(1, world):
        # first contig boundary at or past r/world of the weight, never before the previous cut
        c = int(np.searchsorted(cum, tot * r / world, side="left")) if tot > 0 else n * r // world
        cuts.append(min(n, max(cuts[-1], c)))
    cuts.append(n)
    return [(cuts[r], cuts[r + 1]) for r in range(world)]


def cyclic_tiles(ntiles, rank, world, block_tiles=256):
    """tiles of `rank` under the block-cyclic partition of mfx_hist_launch_cyclic (blocks of `block_tiles` dealt
    round-robin), as a list of (lo, hi) runs -- every rank sees every region of the assembly"""
    out = []
    for b in range(rank, -(-ntiles // block_tiles), world):
        out.append((b * block_tiles, min(ntiles, (b + 1) * block_tiles)))
    return out


def pack_counts(nbins, ncontigs, undr, over, kasm, kmissing, contig_kasm, contig_kmissing):
    """host-side image with the device layout (include/merfin_amd.h MFX_HIST_WORDS)"""
    h = np.zeros(hist_words(nbins, ncontigs), dtype=np.uint64)
    h[:len(undr)] = undr
    h[nbins:nbins + len(over)] = over
    h[2 * nbins] = kasm
    h[2 * nbins + 1] = kmissing
    h[2 * nbins + 3:2 * nbins + 3 + ncontigs] = contig_kasm
    h[2 * nbins + 3 + ncontigs:2 * nbins + 3 + 2 * ncontigs] = contig_kmissing
    return h


def all_reduce_hist(counts, kover):
    """The only collective of the path: sum the uint64 counts image (viewed as
    int64) and the fp64 koverCpy over ranks, in place."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(counts)
        dist.all_reduce(kover)
    return counts, kover


def reduced_result(nbins, ncontigs, counts, kover):
    h = counts.detach().cpu().numpy().view(np.uint64)
    return result_from_counts(nbins, h, float(kover.item()), ncontigs)


# ---------------------------------------------------------------------------
# Sharded index (BASELINE config 5): the table does not fit one GPU, every rank
# owns the k-mers whose minimizer hashes to it.  -hist = route -> exchange ->
# evaluate at the owner -> one all-reduce.
# ---------------------------------------------------------------------------
def exchange_device(keys, contigs, send_counts):
    """all-to-all of the routed k-mers on the device (RCCL over xGMI with backend "nccl")"""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size()
    sc = torch.tensor([int(x) for x in send_counts], dtype=torch.int64, device=keys.device)
    rc = torch.empty(world, dtype=torch.int64, device=keys.device)
    dist.all_to_all_single(rc, sc)
    rcl = [int(x) for x in rc.tolist()]
    scl = [int(x) for x in send_counts]
    n_in = sum(scl)
    rk = torch.empty(sum(rcl), dtype=keys.dtype, device=keys.device)
    rcg = torch.empty(sum(rcl), dtype=contigs.dtype, device=keys.device)
    dist.all_to_all_single(rk, keys[:n_in], output_split_sizes=rcl, input_split_sizes=scl)
    dist.all_to_all_single(rcg, contigs[:n_in], output_split_sizes=rcl, input_split_sizes=scl)
    return rk, rcg


def exchange_comm(comm, stream=None):
    """the same exchange through the library's own collective (csrc/mfx_comm.cpp: one RCCL group of point-to-point
    sends / receives over xGMI per array): no torch process group touches the data path"""
    def ex(keys, contigs, send_counts):
        import torch
        rc = comm.exchange_counts(send_counts, stream=stream)
        n_out = int(rc.sum())
        rk = torch.empty(n_out, dtype=keys.dtype, device=keys.device)
        rcg = torch.empty(n_out, dtype=contigs.dtype, device=keys.device)
        comm.alltoallv(keys, send_counts, rk, rc, 8, stream=stream)
        comm.alltoallv(contigs, send_counts, rcg, rc, 4, stream=stream)
        return rk, rcg
    return ex


def sharded_hist(ev, router, seqs, rank, world, counts, kover, stream=None, exchange=exchange_device, comm=None):
    """-hist over a sharded index.  `ev`/`router` are bound to THIS rank's shard of
    the index; `seqs` is the whole assembly (every rank holds it); this rank routes
    its tile range.  Accumulates into counts/kover (device tensors) and all-reduces.
    comm (binding.Comm): exchange and reduction go through the library's RCCL path instead of torch.distributed."""
    import torch
    if comm is not None:
        exchange = exchange_comm(comm, stream)
    T = seqs.ntiles
    lo, hi = shard(T, rank, world)
    per = router.max_tiles
    rounds = (-(-T // world) + per - 1) // per            # identical on every rank (collectives inside)
    cap = per * TILE
    keys = torch.empty(cap, dtype=torch.int64, device=counts.device)
    ctg = torch.empty(cap, dtype=torch.int32, device=counts.device)
    for r in range(rounds):
        tb = min(hi, lo + r * per)
        te = min(hi, tb + per)
        send = router.route(seqs, tb, te, ev.nbins, counts, keys, ctg, stream=stream)
        rk, rc = exchange(keys, ctg, send)
        if rk.numel():
            ev.hist_keys_launch(rk, rc, rk.numel(), seqs.ncontigs, counts, kover, stream=stream)
        torch.cuda.synchronize()                           # rk/rc are released after this round
    if comm is not None:
        comm.hist_allreduce(ev, counts, kover, seqs.ncontigs, stream=stream)
        return counts, kover
    return all_reduce_hist(counts, kover)


def reduce_completeness(total64, undrcpy64, device=None):
    """-completeness over a sharded index: every rank evaluates the k-mers it owns (the per-piece sums of
    merfin-completeness.C:56-66 restricted to its shard), the 2x64 sums are all-reduced and then added in piece
    order exactly as the reference's final loop does (merfin-completeness.C:117-123).  The per-piece sums are
    integer-valued doubles as long as the K table holds integers, so the split over ranks changes nothing.
    Returns (total, undrcpy, total64, undrcpy64)."""
    import torch
    import torch.distributed as dist
    img = torch.from_numpy(np.stack([np.asarray(total64, dtype=np.float64), np.asarray(undrcpy64, dtype=np.float64)]))
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        if device is not None:
            img = img.to(device)
        dist.all_reduce(img)
    t64, u64 = img.cpu().numpy()
    total = undr = 0.0
    for piece in range(64):
        total += float(t64[piece])
        undr += float(u64[piece])
    return total, undr, t64, u64


class _DeviceBytes:
    """a raw HBM range as a zero-copy uint8 tensor (torch.as_tensor reads __cuda_array_interface__)"""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (int(ptr), False), "version": 2}


def broadcast_index(ix, src=0, device=0, max_gb=0.0, chunk_bytes=1 << 30):
    """Replicated index without N builds: rank `src` passes its built Index, every other rank passes None; the
    table then travels rank-to-rank -- `torch.distributed.broadcast` of the HBM lines in `chunk_bytes` pieces (RCCL
    over xGMI with backend "nccl"; staged through host memory with "gloo") -- and every rank returns an Index
    with identical contents.  Only `src` decodes the k-mer databases and runs the insert kernels."""
    import torch
    import torch.distributed as dist
    from .binding import INDEX_HEADER_BYTES, Index
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return ix
    rank = dist.get_rank()
    on_device = dist.get_backend() == "nccl"
    hdr = torch.from_numpy(ix.image_header().copy()) if rank == src else torch.zeros(INDEX_HEADER_BYTES, dtype=torch.uint8)
    if on_device:
        hdr = hdr.cuda()
    dist.broadcast(hdr, src)
    if rank != src:
        ix = Index.from_header(hdr.cpu().numpy(), max_gb=max_gb, device=device)
    pl, nl, pm, nm = ix.device_image()
    for ptr, nbytes in ((pl, nl), (pm, nm)):
        for o in range(0, nbytes, chunk_bytes):
            n = min(chunk_bytes, nbytes - o)
            view = torch.as_tensor(_DeviceBytes(ptr + o, n), device="cuda")
            if on_device:
                dist.broadcast(view, src)
            else:
                host = view.cpu() if rank == src else torch.empty(n, dtype=torch.uint8)
                dist.broadcast(host, src)
                if rank != src:
                    view.copy_(host)
    torch.cuda.synchronize()
    if rank != src:
        ix.commit()
    return ix
