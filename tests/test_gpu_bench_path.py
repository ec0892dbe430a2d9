"""The path bench.py measures -- `-hist` on the SEQUENCE-ONLY COMPACT index of a k = 21 world built by the bench's own
generator (tools/synth_torch.py: the SURVEY 8(d) recipe) -- against the ORACLE at a size where the regimes of the 3 Gb run
exist: a 256 Mb world at the default load factor (0.18 floor) has hundreds of thousands of queries that leave the one-load
fast path of the probe (mfx_lane_lookup8) -- displaced from their first mini-bucket (first cooperative pass), in a home line
full of other k-mers (second cooperative pass), behind a saturated count field (the side table: the 1000-copy tandem repeats
carry read counts of ~26,000) -- which the ~56 kb worlds of tests/synth.py never produce.  The DEBUG instance of the kernel
counts those endings (mfx_eval_debug_counters) so that the test can assert the regime was there; the measured instance must
give the same result bit for bit, and both must equal the oracle (merfin-histogram.C:54-91; integers exact, koverCpy 1e-6,
measured <= 1e-12).  This is what bench.py's cpu_baseline leg checks on its sample, as a test.

MFX_TEST_ORACLE_BASES scales the world (default 256e6: ~60 s of oracle table building + ~15 s of oracle evaluation on 16 cores)."""
import os

import numpy as np
import pytest

from oracle import pyoracle as po

pytestmark = pytest.mark.gpu

BASES = int(float(os.environ.get("MFX_TEST_ORACLE_BASES", "256e6")))
K, LAM = 21, 26.0


def _cores():
    n = len(os.sched_getaffinity(0))
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = max(1, min(n, int(float(q) / float(per))))
    except Exception:
        pass
    return n


@pytest.fixture(scope="module", params=[0, 3], ids=["iid_world", "repeats_world_3pct"])
def sample(request):
    torch = pytest.importorskip("torch")
    import gc
    import merfin_amd as m
    from oracle import pyoracle as po
    from tools import synth_torch as st
    gc.collect()
    torch.cuda.empty_cache()
    # the FULL databases (every read k-mer, as merylExactLookup::load holds them) are what the oracle looks up in
    # params: SURVEY 8(d)'s i.i.d. genome, and the `repeats` world at its 3 % level (tools/synth_torch.py: inject_repeats -- Alu-like
    # families, satellite arrays, an exact 5-mer tandem array, rDNA-like arrays: several percent of the positions carry read counts beyond
    # the compact slot's 11-bit fields and repeat-family buckets overflow two lines: the worklist of mfx_hist_rest_kernel is busy)
    ix, seqs, asm, info = st.build_world(m, BASES, k=K, lam=LAM, ncontigs=8, seed=st.SEED + 1, repeats=request.param)
    ek, er, ea = ix.export()
    ix.close()
    del ix
    contigs = [a.cpu().numpy().tobytes() for a in asm]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    probK, probP = po.load_kmetric(os.path.join(root, "tests", "golden", "example_lookup_table.txt"))
    R = po.Lookup(K, ek[er > 0], er[er > 0])
    A = po.Lookup(K, ek[ea > 0], ea[ea > 0])
    g, ka, km, _sec = po.hist_run(po.Params(K, LAM, probK, probP), R, A, contigs, threads=_cores(), mode=1, tile=1 << 18)
    del R, A
    read = (ek[er > 0], er[er > 0])
    del ek, er, ea
    yield m, torch, seqs, asm, contigs, read, (probK, probP), (g, ka, km), request.param
    del seqs, asm
    gc.collect()
    torch.cuda.empty_cache()


def _seq_index(m, seqs, read, bases):
    ix = m.Index.for_seq(K, bases + 1024)
    ix.count_asm(seqs)
    step = 1 << 26
    for o in range(0, len(read[0]), step):
        ix.add_read(read[0][o:o + step], read[1][o:o + step])
    return ix


@pytest.mark.parametrize("load_factor", [None, "0.5"])
def test_bench_path_equals_the_oracle_where_every_probe_ending_occurs(sample, load_factor, monkeypatch):
    m, torch, seqs, asm, contigs, read, (probK, probP), (g, ka, km), repeats = sample
    trim = lambda a: np.trim_zeros(np.asarray(a), "b")

    def assert_hist_equal(res, g, ka, km, k):
        # integers bit-exact; koverCpy: the bar is 1e-6 relative, only the summation order differs (1.5 M terms)
        assert (res.kasm, res.kmissing) == (g.kasm, g.kmissing)
        np.testing.assert_array_equal(trim(res.undr()), trim(g.undr()))
        np.testing.assert_array_equal(trim(res.over()), trim(g.over()))
        np.testing.assert_array_equal(res.contig_kasm(), ka)
        np.testing.assert_array_equal(res.contig_kmissing(), km)
        assert res.koverCpy == pytest.approx(g.koverCpy, rel=1e-10)
        assert m.histoQV(res.kmissing, res.kasm, k) == po.histoQV(g.kmissing, g.kasm, k)
    if load_factor:
        monkeypatch.setenv("MFX_LOAD_FACTOR", load_factor)      # the most crowded table the layout is built at: many more full home lines
    ix = _seq_index(m, seqs, read, BASES)
    info = ix.info()
    assert info["seq_only"] and info["compact"]
    slots = info["bytes"] / 8.0
    lf = info["distinct"] / slots
    assert (0.15 < lf < 0.26) if not load_factor else (0.4 < lf < 0.55), lf
    kp = m.KParams(LAM, probK, probP)
    ev = m.Evaluator(ix, kp)
    fast = ev.hist(seqs)                                       # the measured instance <true, true, 21, 4, 6>
    assert_hist_equal(fast, g, ka, km, K)
    ev.debug(True)
    dbg = ev.hist(seqs)                                        # the same code with the counters
    c = ev.debug_counters()
    ev.debug(False)
    assert_hist_equal(dbg, g, ka, km, K)
    assert dbg.koverCpy == fast.koverCpy
    # the regimes of the 3 Gb run are all here, each well beyond 10^5 queries at 256 Mb
    scale = BASES / 256e6
    assert c["first_pass"] > 2e6 * scale, c                    # ~3 % of the k-mers are not in their first mini-bucket
    assert c["side_table"] > 1e5 * scale, c                    # the tandem repeats' saturated read counts
    if repeats:
        assert c["side_table"] > 0.01 * g.kasm, c              # percent of the positions, not a handful
        assert c["line_scans"] > 1e4 * scale, c                # repeat-family buckets beyond two lines: listed for mfx_hist_rest_kernel
    if load_factor:
        assert c["second_pass"] > 1e5 * scale, c               # home lines full of other k-mers
    assert c["second_pass"] > 0, c
    print("probe endings at %d bases, load factor %.3f: %r" % (BASES, lf, c))
    # the streamed evaluation (value_8d's path) and the N-slot path on the same index
    s2 = m.Sequences.create([len(x) for x in contigs])
    assert_hist_equal(ev.hist_streamed(s2, contigs), g, ka, km, K)
    del ev, ix
