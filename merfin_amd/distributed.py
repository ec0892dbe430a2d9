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
