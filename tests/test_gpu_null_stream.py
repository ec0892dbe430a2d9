"""Buffers are filled on one stream and used on another.  hipMemset only QUEUES its fill on the null stream
(tools/native/memset_probe.hip), so while that stream is busy -- another slot's kernels, a caller's own work -- data put
into a fresh buffer over a non-blocking stream used to be wiped by the late fill (eight variant slots on one device lost
a batch of records now and then).  Here the null stream is kept busy by the caller (torch's default stream IS the null
stream) while sequences are uploaded, evaluators made and evaluations run: results must be those of the idle device
(merfin-histogram.C:54-91, merfin-dump.C:44-61 against the oracle)."""
import numpy as np
import pytest

from tests import synth
from tests.test_gpu_parity import assert_hist_equal, build_index, oracle_hist

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(900)]     # (threads: a hang must not take the whole GPU tier with it)


class _BusyNullStream:
    """a thread that keeps a few fp32 GEMMs queued on the null stream (torch's default stream) until it is stopped"""

    def __init__(self, torch):
        import threading
        self.torch, self.stop, self.launched, self.ok, self.error = torch, False, 0, False, None
        self.a = torch.randn(4096, 4096, device="cuda")
        self.t = threading.Thread(target=self._run, daemon=True)

    def _run(self):
        try:
            self._loop()
        except BaseException as e:                               # noqa: BLE001  (seen by __enter__ / the test, not lost in the thread)
            self.error = repr(e)

    def _loop(self):
        torch, x, evs = self.torch, self.a, []
        while not self.stop:
            for _ in range(4):
                x = x @ self.a
                x = x / x.abs().max()
            e = torch.cuda.Event()
            e.record()
            evs.append(e)
            self.launched += 4
            if len(evs) > 3:
                evs.pop(0).synchronize()                     # at most ~16 GEMMs in the queue: the stream is never idle, never far ahead
        torch.cuda.synchronize()
        self.ok = bool(torch.isfinite(x).all())

    def __enter__(self):
        import time
        self.t.start()
        t0 = time.time()
        while self.launched < 8 and self.t.is_alive() and time.time() - t0 < 120:
            time.sleep(0.001)
        assert self.launched >= 8, "the thread that keeps the null stream busy did not start: %s" % self.error
        return self

    def __exit__(self, *exc):
        self.stop = True
        self.t.join()


def test_fills_are_waited_for_while_the_null_stream_is_busy(monkeypatch):
    torch = pytest.importorskip("torch")
    import merfin_amd as m
    k, peak = 21, 17.3
    contigs, read, asm = synth.world(k=k, peak=peak, seed=4711, sizes=(90000, 30000, 4097, 20, 0))
    p, g, ka, km = oracle_hist(k, peak, contigs, read, asm)
    monkeypatch.setenv("MFX_UPLOAD_PACKED_MIN", "1")         # the packed transport (what path texts and assemblies of >= 8 MB take)
    ix = build_index(m, k, read, asm)
    ev = m.Evaluator(ix, m.KParams(peak))
    idle = ev.dump_values(m.Sequences(contigs), 0, 0, len(contigs[0]))
    with _BusyNullStream(torch) as busy:
        for rep in range(6):
            seqs = m.Sequences(contigs)                          # planes filled + uploaded under the GEMMs
            got = ev.dump_values(seqs, 0, 0, len(contigs[0]))    # one byte per base made from the planes, then the lookup kernel
            np.testing.assert_array_equal(got[0], idle[0])
            np.testing.assert_array_equal(got[1], idle[1])
            assert got[2:] == idle[2:]
            assert_hist_equal(ev.hist(m.Sequences(contigs)), g, ka, km, k)
            assert_hist_equal(ev.hist_streamed(m.Sequences.create([len(c) for c in contigs]), contigs), g, ka, km, k)
            ev2 = m.Evaluator(ix, m.KParams(peak))               # an evaluator's counters are filled at its creation
            assert_hist_equal(ev2.hist(seqs), g, ka, km, k)
    assert busy.ok and busy.launched >= 8


def test_builds_and_multi_slot_runs_while_the_null_stream_is_busy(tmp_path):
    """the same for everything else that hands buffers from one stream to another: index builds through the staging lanes (host
    arrays, a database file), the sequence-only build, replicas and N slots on one device, the sharded route -> owner loop,
    -completeness and a variant mode with device scoring -- each compared with its own result on the idle device"""
    torch = pytest.importorskip("torch")
    import merfin_amd as m
    k, peak = 21, 17.3
    contigs, read, asm = synth.world(k=k, peak=peak, seed=4712, sizes=(70000, 20000, 4097, 33))
    p, g, ka, km = oracle_hist(k, peak, contigs, read, asm)
    names, vasm, vcf, vread, vamers = synth.variant_world(k=k, peak=peak, seed=4713)
    vp = str(tmp_path / "in.vcf")
    open(vp, "w").write(vcf)
    db = str(tmp_path / "read.mfxk")
    m.db_write_flat(db, k, read[0], read[1])

    def everything(tag):
        out = {}
        ix = build_index(m, k, read, asm)                                     # host arrays through the staging lanes
        ev = m.Evaluator(ix, m.KParams(peak))
        seqs = m.Sequences(contigs)
        assert_hist_equal(ev.hist(seqs), g, ka, km, k)
        out["compl"] = ev.completeness()
        six = m.Index.for_seq(k, sum(len(c) for c in contigs) + 16)           # claim + count under the database's transfer
        six.build_for_hist(seqs, db)
        assert_hist_equal(m.Evaluator(six, m.KParams(peak)).hist(seqs), g, ka, km, k)
        n = 3                                                                 # replicas + N slots on this device
        ixs = [ix] + [ix.replicate(0) for _ in range(n - 1)]
        sqs = [seqs] + [seqs.replicate(0) for _ in range(n - 1)]
        assert_hist_equal(m.hist_multi([m.Evaluator(x, m.KParams(peak)) for x in ixs], sqs), g, ka, km, k)
        shards = []
        for r in range(n):                                                    # the sharded route -> owner loop
            sx = m.Index(k, len(read[0]) + len(asm[0]) + 16)
            sx.set_shard(r, n)
            sx.add_read(*read)
            sx.add_asm(*asm)
            shards.append(sx)
        sev = [m.Evaluator(x, m.KParams(peak)) for x in shards]
        routers = [m.Router(x, n, min(2, seqs.ntiles)) for x in shards]
        assert_hist_equal(m.hist_sharded(sev, routers, [seqs] * n), g, ka, km, k)
        vix = m.Index(k, len(vread[0]) + len(vamers[0]) + 16)                 # a variant mode, scored on the device
        vix.add_read(*vread)
        vix.add_asm(*vamers)
        o = str(tmp_path / ("polish_%s.vcf" % tag))
        m.Evaluator(vix, m.KParams(peak)).variants("polish", vp, names, vasm, o, log_path=str(tmp_path / "log"))
        out["polish"] = open(o, "rb").read()
        return out

    idle = everything("idle")
    with _BusyNullStream(torch) as busy:
        for rep in range(2):
            got = everything("busy%d" % rep)
            assert got["compl"] == idle["compl"]
            assert got["polish"] == idle["polish"] and len(got["polish"]) > 1000
    assert busy.ok


def test_slots_sharing_one_sequence_ask_for_its_bytes_at_once(monkeypatch):
    """N slots on one device share its mfx_seq (merfin -dump -devices 0,0,0,0).  A sequence uploaded packed makes its bytes per
    base on first use: four threads asking at once must all see them complete (unguarded, one thread's fill of the fresh buffer
    wiped what another had just unpacked: that slot dumped contigs without a single valid k-mer, exit code 0)"""
    import threading
    import merfin_amd as m
    k, peak = 21, 17.3
    contigs, read, asm = synth.world(k=k, peak=peak, seed=4714, sizes=(60000, 50000, 40000, 30000, 20000, 10000, 5000, 4097))
    monkeypatch.setenv("MFX_UPLOAD_PACKED_MIN", "1")
    ix = build_index(m, k, read, asm)
    ref_ev = m.Evaluator(ix, m.KParams(peak))
    ref_seqs = m.Sequences(contigs)
    ref = [ref_ev.dump_values(ref_seqs, c, 0, len(contigs[c])) for c in range(len(contigs))]
    assert sum(r[2] for r in ref) > 100000
    for rep in range(8):
        seqs = m.Sequences(contigs)                              # packed planes only: nobody has asked for the bytes yet
        n = 4
        evs = [m.Evaluator(ix, m.KParams(peak)) for _ in range(n)]
        got, errs = {}, []
        gate = threading.Barrier(n)

        def slot(d):
            try:
                gate.wait()
                for c in range(d, len(contigs), n):
                    got[c] = evs[d].dump_values(seqs, c, 0, len(contigs[c]))
            except Exception as e:                               # noqa: BLE001
                errs.append(repr(e))

        th = [threading.Thread(target=slot, args=(d,)) for d in range(n)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        assert not errs, errs
        for c in range(len(contigs)):
            np.testing.assert_array_equal(got[c][0], ref[c][0])
            np.testing.assert_array_equal(got[c][1], ref[c][1])
            assert got[c][2:] == ref[c][2:]


def test_two_threads_each_with_their_own_objects_on_one_device(tmp_path):
    """two host threads, each with its own index (built from a database file through its own staging lanes), evaluator and
    sequences on the same device, running builds, streamed and resident evaluations and dumps at the same time: every result
    is the oracle's (objects are per thread; the device, its null stream and the library's globals are shared)"""
    import threading
    import merfin_amd as m
    worlds = []
    for i, k in enumerate((21, 27)):
        peak = 17.3
        contigs, read, asm = synth.world(k=k, peak=peak, seed=4720 + i, sizes=(80000, 30000, 4097, 50))
        p, g, ka, km = oracle_hist(k, peak, contigs, read, asm)
        db = str(tmp_path / ("read%d.mfxk" % k))
        m.db_write_flat(db, k, read[0], read[1])
        worlds.append((k, peak, contigs, read, asm, g, ka, km, db))
    errs = []
    gate = threading.Barrier(2)

    def work(w):
        k, peak, contigs, read, asm, g, ka, km, db = w
        try:
            gate.wait()
            for rep in range(4):
                seqs = m.Sequences(contigs)
                six = m.Index.for_seq(k, sum(len(c) for c in contigs) + 16)
                six.build_for_hist(seqs, db)
                ev = m.Evaluator(six, m.KParams(peak))
                assert_hist_equal(ev.hist(seqs), g, ka, km, k)
                assert_hist_equal(ev.hist_streamed(m.Sequences.create([len(c) for c in contigs]), contigs), g, ka, km, k)
                full = m.Evaluator(build_index(m, k, read, asm), m.KParams(peak))
                a = ev.dump_values(seqs, 0, 0, len(contigs[0]))
                b = full.dump_values(seqs, 0, 0, len(contigs[0]))
                np.testing.assert_array_equal(a[0], b[0])
                np.testing.assert_array_equal(a[1], b[1])
                assert a[2:] == b[2:]
        except BaseException as e:                             # noqa: BLE001
            import traceback
            errs.append(traceback.format_exc()[-1500:])

    th = [threading.Thread(target=work, args=(w,)) for w in worlds]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, "\n".join(errs)
