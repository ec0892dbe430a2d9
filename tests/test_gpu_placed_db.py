"""The PLACED database on the device (csrc/mfx_place.h; mfx_table_add_placed_kernel): its records are sorted by where the compact table
of a sequence-only index puts them, so the update walks the table line after line.  Whatever the order of its records, a database
builds the same table: placed == k-mer-sorted, for the compact layout (direct form and quotient form; the records' own placement is
taken), for tables that place anew (16-byte slots, the full table), staged and unstaged, and through the CLI."""
import os
import subprocess

import numpy as np
import pytest

from oracle import pyoracle as po
from tests import synth
from tests.test_gpu_parity import assert_hist_equal, oracle_hist

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _mfx():
    import merfin_amd as m
    if m.device_count() < 1:
        pytest.fail("no HIP device visible: the GPU tests must run on the MI355X box")
    return m


def _export_sorted(ix):
    k_, r_, a_ = ix.export()
    o = np.argsort(k_, kind="stable")
    return k_[o], r_[o], a_[o]


def _padded(k, read, asm, n_extra=200000, seed=5):
    """the read database + k-mers neither it nor the sequence holds (dropped by a sequence-only index, still decoded), sorted"""
    rng = np.random.default_rng(seed)
    extra = np.unique(rng.integers(0, 1 << (2 * k - 1), n_extra, dtype=np.uint64))
    x, r = extra.copy(), np.zeros_like(extra)
    for _ in range(k):
        r = (r << np.uint64(2)) | ((x & np.uint64(3)) ^ np.uint64(2))
        x >>= np.uint64(2)
    canon = np.setdiff1d(np.setdiff1d(np.unique(np.minimum(extra, r)), read[0]), asm[0])
    keys = np.concatenate([read[0], canon])
    vals = np.concatenate([read[1], np.full(len(canon), 3, dtype=np.uint32)])
    o = np.argsort(keys, kind="stable")
    return keys[o], vals[o]


@pytest.mark.parametrize("k", [13, 17, 21, 22, 27, 30, 31])
def test_placed_database_builds_the_same_table(k, tmp_path, monkeypatch):
    import torch
    m = _mfx()
    peak = 11.0
    contigs, read, asm = synth.world(k=k, peak=peak, seed=70 + k)
    keys, vals = _padded(k, read, asm)
    vals[::501] = 5000 + (np.arange(len(vals[::501])) % 7).astype(np.uint32) * 100000      # counts beyond any block's field: the escape list
    vals[7::1009] = 3000000000                                                              # ... and beyond 31 bits (k = 31 folds its strand bit into the field)
    truth = dict(zip(keys.tolist(), vals.tolist()))
    flat, placed = str(tmp_path / "r.mfxk"), str(tmp_path / "p.mfxk")
    m.db_write_flat(flat, k, keys, vals)
    assert m.db_convert_placed(flat, placed) == len(keys)
    if k <= 30:                                               # the placement numbers: device == host (k = 31: 65 bits, made inside the converter only)
        host = m.db_place_keys(k, keys)
        dev = m.db_place_keys(k, torch.from_numpy(keys.view(np.int64)).cuda()).cpu().numpy().view(np.uint64)
        np.testing.assert_array_equal(dev, host)
    else:
        with pytest.raises(m.MfxError):
            m.db_place_keys(k, keys)
    # ... and back: the placed file converted to the sorted form holds the k-mers and counts it was made of
    back = str(tmp_path / "back.mfxk")
    assert m.db_convert(placed, back) == len(keys)
    ixb = m.Index(k, len(keys) + 16)
    ixb.load_db(back, 0)
    bk, br, _ = _export_sorted(ixb)
    np.testing.assert_array_equal(bk, keys)
    np.testing.assert_array_equal(br, vals)
    del ixb
    seqs = m.Sequences(contigs)
    nb = sum(len(c) for c in contigs)
    lo, hi = 2, 4000000

    def build(path, compact=True, staged=False):
        if not compact:
            monkeypatch.setenv("MFX_SEQ_COMPACT", "0")
        ix = m.Index.for_seq(k, nb + 16)
        monkeypatch.delenv("MFX_SEQ_COMPACT", raising=False)
        assert bool(ix.info()["compact"]) == compact
        if staged:
            st = m.DbStage(path)
            assert st.ok, st.why
            ix.build_for_hist_staged(seqs, st, lo, hi)
            st.close()
        else:
            ix.build_for_hist(seqs, path, lo, hi)
        return ix
    want = _export_sorted(build(flat))
    p, g, ka, km = oracle_hist(k, peak, contigs, (read[0], np.where((vals[np.searchsorted(keys, read[0])] >= lo) & (vals[np.searchsorted(keys, read[0])] <= hi),
                                                                   vals[np.searchsorted(keys, read[0])], 0).astype(np.uint32)), asm)
    for compact in (True, False):
        for staged in (False, True):
            ix = build(placed, compact, staged)
            got = _export_sorted(ix)
            for a_, b_ in zip(got, want):
                np.testing.assert_array_equal(a_, b_)
            np.testing.assert_array_equal(got[1], np.array([truth.get(x, 0) for x in got[0].tolist()], dtype=np.uint32))
            assert ix.info()["dropped"] > 0
            assert_hist_equal(m.Evaluator(ix, m.KParams(peak)).hist(seqs), g, ka, km, k)
    # the full table (every record claims a slot) takes the same file
    full_p, full_f = m.Index(k, len(keys) + nb + 16), m.Index(k, len(keys) + nb + 16)
    full_p.load_db(placed, 0)
    full_f.load_db(flat, 0)
    for a_, b_ in zip(_export_sorted(full_p), _export_sorted(full_f)):
        np.testing.assert_array_equal(a_, b_)


def test_cli_hist_from_a_placed_database(tmp_path, golden_dir):
    """`merfin -convert -placed` of the golden read database, then -hist from it: the golden histogram, byte for byte"""
    exe = os.path.join(ROOT, "merfin_amd", "bin", "merfin")
    g = lambda n: os.path.join(golden_dir, n)
    r = subprocess.run([exe, "-convert", g("case1.read.kmers.txt"), "-placed", "-output", str(tmp_path / "read.placed")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    for extra in ([], ["-seqmers", g("case1.asm.kmers.txt")]):
        for env in ({}, {"MFX_DB_STAGE": "0"}):
            r = subprocess.run([exe, "-hist", "-sequence", g("case1.fasta"), "-readmers", str(tmp_path / "read.placed"), "-peak", "17.3", "-prob",
                                g("example_lookup_table.txt"), "-output", str(tmp_path / "h")] + extra, capture_output=True, text=True, env=dict(os.environ, **env))
            assert r.returncode == 0, r.stderr
            assert (tmp_path / "h").read_bytes() == open(g("case1.hist"), "rb").read()
            assert open(g("case1.summary")).read() in r.stderr


@pytest.mark.parametrize("k", [21, 27])
def test_placed_records_that_are_no_placement_number_are_refused(k, tmp_path):
    """a placed file is applied with plain stores, one record per slot, at the place its own pieces name: numbers that ascend but
    are no canonical k-mer's P -- a damaged or hand-made file -- must fail the load (MFX_E_FORMAT), not land in the table"""
    m = _mfx()
    contigs, read, asm = synth.world(k=k, peak=11.0, seed=90 + k)
    good = np.sort(m.db_place_keys(k, read[0]))
    rng = np.random.default_rng(3)
    junk = rng.integers(1, 1 << 62, 256, dtype=np.uint64) >> np.uint64(62 - min(62, 2 * k + 3))
    bad = np.unique(np.concatenate([good, junk]))
    assert len(bad) > len(good)
    path = str(tmp_path / "bad.mfxk")
    m.db_write_flat_placed(path, k, bad, np.full(len(bad), 7, dtype=np.uint32))
    seqs = m.Sequences(contigs)
    ix = m.Index.for_seq(k, sum(len(c) for c in contigs) + 16)
    with pytest.raises(m.MfxError) as e:
        ix.build_for_hist(seqs, path)
    assert "damaged" in str(e.value) or "wider" in str(e.value)
    ok = str(tmp_path / "ok.mfxk")                                 # the same records without the junk load
    m.db_write_flat_placed(ok, k, good, np.full(len(good), 7, dtype=np.uint32))
    ix2 = m.Index.for_seq(k, sum(len(c) for c in contigs) + 16)
    ix2.build_for_hist(seqs, ok)
