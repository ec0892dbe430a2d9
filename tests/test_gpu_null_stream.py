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


def _busy(torch, a, n=40):
    for _ in range(n):                                       # ~0.2-0.4 s of fp32 GEMMs queued on the null stream
        a = a @ a
        a = a / a.abs().max()
    return a


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
    a = torch.randn(6144, 6144, device="cuda")
    for rep in range(3):
        x = _busy(torch, a)
        seqs = m.Sequences(contigs)                          # planes filled + uploaded under the GEMMs
        got = ev.dump_values(seqs, 0, 0, len(contigs[0]))    # one byte per base made from the planes, then the lookup kernel
        np.testing.assert_array_equal(got[0], idle[0])
        np.testing.assert_array_equal(got[1], idle[1])
        assert got[2:] == idle[2:]
        x = _busy(torch, x)
        assert_hist_equal(ev.hist(m.Sequences(contigs)), g, ka, km, k)
        x = _busy(torch, x)
        assert_hist_equal(ev.hist_streamed(m.Sequences.create([len(c) for c in contigs]), contigs), g, ka, km, k)
        x = _busy(torch, x)
        ev2 = m.Evaluator(ix, m.KParams(peak))               # an evaluator's counters are filled at its creation
        assert_hist_equal(ev2.hist(seqs), g, ka, km, k)
        torch.cuda.synchronize()
        assert bool(torch.isfinite(x).all())
