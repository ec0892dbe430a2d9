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

pytestmark = pytest.mark.gpu


class _BusyNullStream:
    """a thread that keeps a few fp32 GEMMs queued on the null stream (torch's default stream) until it is stopped"""

    def __init__(self, torch):
        import threading
        self.torch, self.stop, self.launched = torch, False, 0
        self.a = torch.randn(4096, 4096, device="cuda")
        self.t = threading.Thread(target=self._run, daemon=True)

    def _run(self):
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
        self.t.start()
        while self.launched < 8:
            pass
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
