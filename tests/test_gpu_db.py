"""k-mer database ingest (disk -> device index): flat binary, `meryl print`
text, and a meryl-layout directory written by tests/meryl_layout.py."""
import numpy as np
import pytest

from oracle import pyoracle as po
from tests import meryl_layout, synth
from tests.test_cli import _write_text_db


@pytest.mark.gpu
@pytest.mark.parametrize("k,prefix_bits", [(21, 12), (15, 8), (31, 14)])
def test_three_db_forms_load_identically(tmp_path, k, prefix_bits):
    import merfin_amd as m
    contigs, read, asm = synth.world(k=k, seed=51, sizes=(6000, 2500, 300), err_kmers=300)
    rk, rv = read
    flat, text, mdir = str(tmp_path / "r.mfxk"), str(tmp_path / "r.txt"), str(tmp_path / "r.meryl")
    m.db_write_flat(flat, k, rk, rv)
    _write_text_db(text, k, rk, rv)
    meryl_layout.write_db(mdir, k, rk, rv, prefix_bits=prefix_bits)
    for path, fmt in ((flat, "flat"), (text, "text"), (mdir, "meryl")):
        info = m.db_probe(path)
        assert info == {"k": k, "format": fmt, "n_kmers": len(rk)}
        ix = m.Index(k, len(rk) + 16)
        ix.load_db(path, 0, 3, 60)                       # -min 3 -max 60
        ek, er, ea = ix.export()
        np.testing.assert_array_equal(ek, rk)            # raw values are stored ...
        np.testing.assert_array_equal(er, rv)
        got, _ = ix.value(rk)                            # ... and filtered at query time
        np.testing.assert_array_equal(got, np.where((rv >= 3) & (rv <= 60), rv, 0))


@pytest.mark.gpu
def test_db_errors_are_reported(tmp_path):
    import merfin_amd as m
    with pytest.raises(m.MfxError) as e:
        m.db_probe(str(tmp_path / "nope"))
    assert e.value.code == -6
    bad = tmp_path / "bad.meryl"
    bad.mkdir()
    (bad / "merylIndex").write_bytes(b"\0" * 256)
    with pytest.raises(m.MfxError) as e:
        m.db_probe(str(bad))
    assert e.value.code == -7                            # refuses, never guesses
    (tmp_path / "t.txt").write_text("ACGTACGTA\t5\nACGT\t3\n")
    with pytest.raises(m.MfxError) as e:
        m.db_probe(str(tmp_path / "t.txt"))
    assert e.value.code == -7 and "differs" in str(e.value)
    k = 9
    m.db_write_flat(str(tmp_path / "k9.mfxk"), k, np.array([5, 9], dtype=np.uint64), np.array([1, 2], dtype=np.uint32))
    ix = m.Index(11, 100)
    with pytest.raises(m.MfxError) as e:
        ix.load_db(str(tmp_path / "k9.mfxk"), 0)
    assert e.value.code == -1
