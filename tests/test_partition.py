"""contig_partition: the contiguous, ordered split of contigs over ranks used by the multi-GPU -dump / variant modes"""
import numpy as np

from merfin_amd import distributed as D


def test_contig_partition_properties():
    r = np.random.default_rng(3)
    for trial in range(200):
        n = int(r.integers(0, 40))
        world = int(r.integers(1, 9))
        w = r.integers(0, 1000, size=n).astype(float) * (r.random(n) < 0.8)
        parts = D.contig_partition(w, world)
        assert len(parts) == world and parts[0][0] == 0 and parts[-1][1] == n
        assert all(lo <= hi for lo, hi in parts)
        assert all(parts[i][1] == parts[i + 1][0] for i in range(world - 1))       # contiguous, covering, ordered
        if n and w.sum() > 0:
            loads = [w[lo:hi].sum() for lo, hi in parts]
            assert max(loads) <= w.sum() / world + w.max() + 1e-9                   # never worse than ideal + one contig


def test_contig_partition_examples():
    assert D.contig_partition([5, 1, 1, 1, 8, 2, 2], 3) == [(0, 3), (3, 5), (5, 7)]
    assert D.contig_partition([1], 4) == [(0, 1), (1, 1), (1, 1), (1, 1)]
    assert D.contig_partition([], 2) == [(0, 0), (0, 0)]


def test_cyclic_tiles_cover_every_tile_once():
    r = np.random.default_rng(4)
    for trial in range(100):
        T = int(r.integers(0, 5000))
        world = int(r.integers(1, 9))
        blk = int(2 ** r.integers(0, 9))
        seen = np.zeros(T, dtype=np.int32)
        for rank in range(world):
            runs = D.cyclic_tiles(T, rank, world, blk)
            for lo, hi in runs:
                assert 0 <= lo < hi <= T and hi - lo <= blk and lo % blk == 0 and (lo // blk) % world == rank
                seen[lo:hi] += 1
        assert (seen == 1).all()
