"""Every MFX_* knob that selects ANOTHER code path against the oracle -- the reference has no such knobs
(SURVEY section 5), so whatever one of them selects must answer exactly as merfin does (merfin-histogram.C:35-176,
merfin-dump.C:20-104).  The defaults are what every other test runs; here each alternative, one at a time:
  * library knobs: a sequence-only index built FROM A DATABASE FILE through the staging pipeline
    (mfx_index_build_for_hist), evaluated resident and streamed, k = 21 (compact layout) and k = 27 (quotient form),
    counts beyond the 11-bit fields so that the side table is in play;
  * CLI knobs: the golden case through `merfin` (-hist, -dump, -polish, -filter), byte for byte."""
import os
import subprocess

import numpy as np
import pytest

from oracle import pyoracle as po
from tests import synth
from tests.test_golden import COMB, EXE, G, PEAK
from tests.test_gpu_parity import assert_hist_equal, build_index, oracle_hist

pytestmark = pytest.mark.gpu

LIB_KNOBS = [
    ("MFX_COUNT_ASCII", "1"),            # claim / count kernel reads one byte per base instead of the packed planes
    ("MFX_UPLOAD_ASCII", "1"),           # mfx_seq_upload sends bytes
    ("MFX_UPLOAD_PACKED_MIN", "1"),      # ... and packs even the smallest sequence
    ("MFX_SIDE_DIV", "4"), ("MFX_SIDE_DIV", "4096"),
    ("MFX_INGEST_LANES", "2"), ("MFX_INGEST_LANES", "3"),
    ("MFX_INGEST_CHUNK_LOG2", "16"),     # the smallest lanes: the load goes through them in several chunks
    ("MFX_INGEST_RING_MB", "0"), ("MFX_BUILD_OVERLAP", "0"),
    ("MFX_POOL_SPREAD", "0"), ("MFX_NUMA_BIND", "0"),
    ("MFX_PACK_PLACE", "os"), ("MFX_PACK_PLACE", "node"), ("MFX_PACK_PLACE", "all"),
    ("MFX_PACK_ISA", "scalar"),
    ("MFX_STREAM_SPARSE_VALID", "0"), ("MFX_STREAM_ASCII", "1"),
    ("MFX_SEQ_QUOT", "0"), ("MFX_SEQ_COMPACT", "0"), ("MFX_MZ_MOD", "0"), ("MFX_HOME_MODE", "plain"),
    ("MFX_INSERT_MODE", "0"), ("MFX_COUNT_MODE", "1"),
    ("MFX_BLOCKS_PER_CU", "1"), ("MFX_HOST_THREADS", "1"), ("MFX_HIST_GENERIC", "1"), ("MFX_FORCE_TWO_STRAND", "1"),
    ("MFX_FLAT_DELTA", "0"), ("MFX_FLAT_PACKED", "0"),
    ("MFX_LOAD_FACTOR", "0.6"),
]

_worlds = {}


def _world(k):
    """one world per k for the whole matrix: saturated read counts, N runs, a short and an empty contig"""
    if k not in _worlds:
        peak = 17.3
        contigs, read, asm = synth.world(k=k, peak=peak, seed=8800 + k, sizes=(60000, 9000, 4097, 40, 0), err_kmers=3000)
        r = np.random.default_rng(8900 + k)
        rv = read[1].astype(np.uint64)
        big = r.random(len(rv)) < 0.02
        rv[big] = r.choice([2046, 2047, 2048, 70000], size=int(big.sum()))
        read = (read[0], rv.astype(np.uint32))
        p, g, ka, km = oracle_hist(k, peak, contigs, read, asm, None, None)
        _worlds[k] = (peak, contigs, read, asm, g, ka, km)
    return _worlds[k]


@pytest.mark.parametrize("knob,value", LIB_KNOBS, ids=["%s=%s" % kv for kv in LIB_KNOBS])
@pytest.mark.parametrize("k", [21, 27])
def test_library_knob_answers_as_the_oracle(k, knob, value, tmp_path, monkeypatch):
    import merfin_amd as m
    peak, contigs, read, asm, g, ka, km = _world(k)
    monkeypatch.setenv(knob, value)
    db = str(tmp_path / "read.mfxk")
    m.db_write_flat(db, k, read[0], read[1])                        # (MFX_FLAT_*: the record form of the file)
    seqs = m.Sequences(contigs)
    ix = m.Index.for_seq(k, sum(len(c) for c in contigs) + 16)
    ix.build_for_hist(seqs, db)                                     # claim + count under the database's transfer, inserts after
    info = ix.info()
    assert info["seq_only"] and info["distinct"] == len(asm[0])
    ev = m.Evaluator(ix, m.KParams(peak))
    assert_hist_equal(ev.hist(seqs), g, ka, km, k)
    assert_hist_equal(ev.hist_streamed(m.Sequences.create([len(c) for c in contigs]), contigs), g, ka, km, k)
    # per-base values of the first contig against the full joint table built the plain way
    monkeypatch.delenv(knob)
    full = m.Evaluator(build_index(m, k, read, asm), m.KParams(peak))
    monkeypatch.setenv(knob, value)
    a = ev.dump_values(seqs, 0, 0, len(contigs[0]))
    b = full.dump_values(seqs, 0, 0, len(contigs[0]))
    np.testing.assert_array_equal(a[0], b[0])
    np.testing.assert_array_equal(a[1], b[1])
    assert a[2:] == b[2:]
    rv, av = ix.value(asm[0])
    rd = dict(zip(read[0].tolist(), read[1].tolist()))
    np.testing.assert_array_equal(av, asm[1])
    np.testing.assert_array_equal(rv, np.array([rd.get(x, 0) for x in asm[0].tolist()], dtype=np.uint32))


CLI_KNOBS = [
    ("MFX_DUMP_SERIAL", "1"), ("MFX_CLI_WARM", "0"), ("MFX_CLI_QUICK_EXIT", "1"), ("MFX_CLI_FULL_INDEX", "1"), ("MFX_CLI_OVERLAP", "1"), ("MFX_CLI_VCF_AHEAD", "0"),
    ("MFX_CLI_SEQ_THREADS", "1"), ("MFX_VARIANT_SLOTS", "3"), ("MFX_VAR_BATCH_MB", "1"), ("MFX_VAR_HOST_SCORE", "1"),
    ("MFX_HOST_THREADS", "1"), ("MFX_HOST_THREADS", "5"), ("MFX_CLI_TIMING", "3"), ("MFX_DUMP_TIMING", "1"), ("MFX_VAR_TIMING", "1"),
    ("MFX_INGEST_TIMING", "1"), ("MFX_UPLOAD_TIMING", "1"), ("MFX_CLI_SEQ_TIMING", "1"),
    ("MFX_VAR_DEVICE_TRAVERSE", "0"), ("MFX_VAR_TRAVERSE_CHECK", "1"), ("MFX_CLI_VCF_AHEAD", "2"), ("MFX_CLI_VCF_AHEAD", "1"), ("MFX_CLI_STAGE_FIRST", "0"),
    ("MFX_DB_STAGE_THREADS", "2"), ("MFX_DB_STAGE_LANES", "6"), ("MFX_PREAD_THREADS", "3"),
]


@pytest.mark.parametrize("knob,value", CLI_KNOBS, ids=["%s=%s" % kv for kv in CLI_KNOBS])
def test_cli_knob_reproduces_golden(knob, value, tmp_path):
    env = dict(os.environ)
    env[knob] = value
    common = ["-sequence", G + "/case1.fasta", "-readmers", G + "/case1.read.kmers.txt", "-peak", str(PEAK), "-prob", G + "/example_lookup_table.txt"]
    r = subprocess.run([EXE, "-hist"] + common + ["-output", str(tmp_path / "h")], capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr
    assert (tmp_path / "h").read_bytes() == open(G + "/case1.hist", "rb").read()
    assert open(G + "/case1.summary").read() in r.stderr
    r = subprocess.run([EXE, "-dump"] + common + ["-output", str(tmp_path / "d")], capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr
    assert (tmp_path / "d").read_bytes() == open(G + "/case1.dump", "rb").read()
    for mode, suffix in (("polish", ".polish.vcf"), ("filter", ".filter.vcf")):
        out = str(tmp_path / mode)
        r = subprocess.run([EXE, "-" + mode] + common + ["-seqmers", G + "/case1.asm.kmers.txt", "-vcf", G + "/case1.vcf", "-comb", str(COMB), "-output", out],
                           capture_output=True, text=True, env=env)
        assert r.returncode == 0, r.stderr
        assert open(out + suffix, "rb").read() == open(G + "/case1.%s.vcf" % mode, "rb").read()


@pytest.mark.parametrize("knob,value", [("MFX_SHARDED_ORDERED", "1"), ("MFX_ROUTE_SORT", "1"), ("MFX_MULTI_TIMING", "1")])
def test_sharded_knob_answers_as_the_oracle(knob, value, monkeypatch):
    """the one-process route -> owner loop over a full table sharded 3 ways (all shards on GPU 0)"""
    import merfin_amd as m
    k = 21
    peak, contigs, read, asm, g, ka, km = _world(k)
    monkeypatch.setenv(knob, value)
    n = 3
    ixs = []
    for r in range(n):
        ix = m.Index(k, len(read[0]) + len(asm[0]) + 16)
        ix.set_shard(r, n)
        ix.add_read(*read)
        ix.add_asm(*asm)
        ixs.append(ix)
    evs = [m.Evaluator(ix, m.KParams(peak)) for ix in ixs]
    seqs = m.Sequences(contigs)
    routers = [m.Router(ix, n, min(2, seqs.ntiles)) for ix in ixs]          # several routing rounds
    assert_hist_equal(m.hist_sharded(evs, routers, [seqs] * n), g, ka, km, k)


@pytest.mark.parametrize("threads", ["64", "3"])
def test_streamed_chunk_of_gappy_contigs(threads, monkeypatch):
    """a chunk whose validity words are nearly ALL exceptional (an N in every run of 32 bases, contigs of a few hundred
    bases): the sparse form of the validity plane steps aside (a list entry is 8 bytes, a word sent whole 4), with many
    and with few packer threads -- streamed == resident == oracle"""
    import merfin_amd as m
    monkeypatch.setenv("MFX_HOST_THREADS", threads)
    k, peak = 15, 9.0
    r = synth.rng(4242)
    contigs = []
    for i in range(300):
        c = synth.random_contig(r, 200 + int(r.integers(0, 56)))
        c[int(r.integers(0, 31))::31] = ord("N")
        contigs.append(c.tobytes())
    ak, av = po.count_kmers(k, contigs)
    rv = r.poisson(peak * av.astype(np.float64)).astype(np.uint32)
    read, asm = (ak[rv > 0], rv[rv > 0]), (ak, av)
    p, g, ka, km = oracle_hist(k, peak, contigs, read, asm, None, None)
    assert g.kasm > 1000
    ev = m.Evaluator(build_index(m, k, read, asm), m.KParams(peak))
    assert_hist_equal(ev.hist(m.Sequences(contigs)), g, ka, km, k)
    assert_hist_equal(ev.hist_streamed(m.Sequences.create([len(c) for c in contigs]), contigs), g, ka, km, k)
