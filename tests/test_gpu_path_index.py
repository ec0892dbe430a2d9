"""The PATH-ONLY index of the variant modes (mfx_vcf_prepare + mfx_index_claim_paths): varMer::score (varMer.C:76-84) asks the lookup tables
for the k-mers of the enumerated paths and nothing else, so a sequence-only index that claimed exactly those k-mers -- both databases then
only update them -- must give the records, the log and the -debug lines of the full tables (merfin-globals.C:114-163), byte for byte."""
import numpy as np
import pytest

from tests import synth


def _world(tmp_path, k, peak, seed, **kw):
    names, asm, vcf, read, amers = synth.variant_world(k=k, peak=peak, seed=seed, **kw)
    vp = str(tmp_path / "in.vcf")
    open(vp, "w").write(vcf)
    return names, asm, vp, read, amers


def _full(m, k, read, amers, peak):
    ix = m.Index(k, len(read[0]) + len(amers[0]) + 16)
    ix.add_read(*read)
    ix.add_asm(*amers)
    return ix, m.Evaluator(ix, m.KParams(peak))


def _path_index(m, k, loaded, read, amers):
    n = loaded.path_bound()
    assert n > 0
    px = m.Index.for_seq(k, n + 1024)
    assert loaded.claim_paths(px) == n
    px.add_asm(*amers)                                         # update-only: a k-mer of no path never gets a slot
    px.add_read(*read)
    return px


@pytest.mark.gpu
@pytest.mark.parametrize("mode,k,seed,comb,nosplit", [("polish", 21, 78, 15, False), ("filter", 21, 79, 15, False), ("better", 27, 80, 6, False),
                                                      ("loose", 31, 81, 15, True), ("strict", 15, 83, 4, False), ("polish", 11, 84, 15, False)])
@pytest.mark.parametrize("fused", [False, True])
def test_path_only_index_gives_the_records_of_the_full_tables(tmp_path, mode, k, seed, comb, nosplit, fused):
    """fused: mfx_vcf_prepare_path_index (what the CLI calls: the table made once the clusters are merged, every batch claimed under the
    preparation of the next) instead of prepare / path_bound / Index.for_seq / claim_paths"""
    import merfin_amd as m
    peak = 17.3
    names, asm, vp, read, amers = _world(tmp_path, k, peak, seed)
    ix, ev = _full(m, k, read, amers, peak)
    n_a = ev.variants(mode, vp, names, asm, str(tmp_path / "a.vcf"), comb=comb, nosplit=nosplit, log_path=str(tmp_path / "a.log"))
    loaded = m.LoadedVcf(vp)
    if fused:
        px = loaded.prepare_path_index(k, mode, names, asm, comb=comb, nosplit=nosplit)
        assert px is not None
        px.add_asm(*amers)
        px.add_read(*read)
    else:
        loaded.prepare(k, mode, names, asm, comb=comb, nosplit=nosplit)
        px = _path_index(m, k, loaded, read, amers)
    info = px.info()
    assert info["seq_only"] and 0 < info["distinct"] <= loaded.path_bound()       # the claimed k-mers: the paths', whatever the databases hold
    pev = m.Evaluator(px, m.KParams(peak))
    n_b = pev.variants_loaded(mode, loaded, names, asm, str(tmp_path / "b.vcf"), comb=comb, nosplit=nosplit, log_path=str(tmp_path / "b.log"))
    assert n_a == n_b and n_a > 0
    assert open(tmp_path / "a.vcf", "rb").read() == open(tmp_path / "b.vcf", "rb").read()
    assert open(tmp_path / "a.log", "rb").read() == open(tmp_path / "b.log", "rb").read()
    assert len([l for l in open(tmp_path / "b.vcf") if not l.startswith("#")]) > 20
    loaded.close()


@pytest.mark.gpu
@pytest.mark.parametrize("host_paths", [False, True])
def test_path_only_index_with_debug_and_host_enumeration(tmp_path, monkeypatch, host_paths):
    """-debug scores on the host from per-base values of host-enumerated text; MFX_VAR_DEVICE_TRAVERSE=0 enumerates every cluster on the
    host: the claims come from the packed text instead of the device's traverse, the answers are the same"""
    import merfin_amd as m
    if host_paths:
        monkeypatch.setenv("MFX_VAR_DEVICE_TRAVERSE", "0")
    k, peak = 21, 17.3
    names, asm, vp, read, amers = _world(tmp_path, k, peak, 85, burst=0.3)
    ix, ev = _full(m, k, read, amers, peak)
    dbg = None if host_paths else str(tmp_path / "a.dbg")
    n_a = ev.variants("polish", vp, names, asm, str(tmp_path / "a.vcf"), debug_path=dbg, log_path=str(tmp_path / "a.log"))
    loaded = m.LoadedVcf(vp)
    loaded.prepare(k, "polish", names, asm, debug_path=None if host_paths else str(tmp_path / "unused"))
    px = _path_index(m, k, loaded, read, amers)
    pev = m.Evaluator(px, m.KParams(peak))
    n_b = pev.variants_loaded("polish", loaded, names, asm, str(tmp_path / "b.vcf"), debug_path=None if host_paths else str(tmp_path / "b.dbg"),
                              log_path=str(tmp_path / "b.log"))
    assert n_a == n_b and n_a > 0
    for ext in ("vcf", "log") + (() if host_paths else ("dbg",)):
        assert open(tmp_path / ("a." + ext), "rb").read() == open(tmp_path / ("b." + ext), "rb").read(), ext
    loaded.close()


@pytest.mark.gpu
def test_path_only_index_claimed_batch_by_batch(tmp_path, monkeypatch):
    """one cluster per batch (MFX_VAR_BATCH_MB=0): hundreds of claims chained behind one another under the preparation"""
    import merfin_amd as m
    monkeypatch.setenv("MFX_VAR_BATCH_MB", "0")
    k, peak = 21, 17.3
    names, asm, vp, read, amers = _world(tmp_path, k, peak, 88)
    ix, ev = _full(m, k, read, amers, peak)
    n_a = ev.variants("polish", vp, names, asm, str(tmp_path / "a.vcf"), log_path=str(tmp_path / "a.log"))
    loaded = m.LoadedVcf(vp)
    px = loaded.prepare_path_index(k, "polish", names, asm)
    assert px is not None and px.info()["distinct"] > 0
    px.add_asm(*amers)
    px.add_read(*read)
    pev = m.Evaluator(px, m.KParams(peak))
    n_b = pev.variants_loaded("polish", loaded, names, asm, str(tmp_path / "b.vcf"), log_path=str(tmp_path / "b.log"))
    assert n_a == n_b and n_a > 0
    for ext in ("vcf", "log"):
        assert open(tmp_path / ("a." + ext), "rb").read() == open(tmp_path / ("b." + ext), "rb").read(), ext
    loaded.close()


@pytest.mark.gpu
def test_path_only_index_answers_its_call_set_and_nothing_else(tmp_path):
    import merfin_amd as m
    k, peak = 21, 17.3
    names, asm, vp, read, amers = _world(tmp_path, k, peak, 86)
    loaded = m.LoadedVcf(vp)
    with pytest.raises(m.MfxError):
        loaded.path_bound()                                    # not prepared
    loaded.prepare(k, "polish", names, asm)
    full = m.Index(k, 1 << 16)
    with pytest.raises(m.MfxError):
        loaded.claim_paths(full)                               # not a sequence-only index
    other_k = m.Index.for_seq(k + 2, loaded.path_bound() + 1024)
    with pytest.raises(m.MfxError):
        loaded.claim_paths(other_k)                            # prepared for another k
    px = _path_index(m, k, loaded, read, amers)
    with pytest.raises(m.MfxError):
        loaded.claim_paths(px)                                 # counts arrived: no more claims
    pev = m.Evaluator(px, m.KParams(peak))
    with pytest.raises(m.MfxError):                            # an unprepared run enumerates on its own: refused
        pev.variants("polish", vp, names, asm, str(tmp_path / "x.vcf"), log_path=str(tmp_path / "x.log"))
    twin = m.LoadedVcf(vp)
    twin.prepare(k, "polish", names, asm)
    with pytest.raises(m.MfxError):                            # the same file prepared again is ANOTHER call set
        pev.variants_loaded("polish", twin, names, asm, str(tmp_path / "y.vcf"), log_path=str(tmp_path / "y.log"))
    seqs = m.Sequences(asm)
    with pytest.raises(m.MfxError):                            # the sequence's own k-mers were not claimed as such
        pev.hist(seqs)
    with pytest.raises(m.MfxError):
        pev.dump_values(seqs, 0, 0, 1000)
    assert pev.variants_loaded("polish", loaded, names, asm, str(tmp_path / "b.vcf"), log_path=str(tmp_path / "b.log")) > 0
    twin.close()
    loaded.close()


@pytest.mark.gpu
def test_path_only_index_from_database_files_and_the_sequence(tmp_path):
    """the CLI's order: claim, then the assembly side either from -seqmers (update-only load) or counted over the uploaded sequence
    (mfx_index_count_claimed), then -readmers"""
    import merfin_amd as m
    k, peak = 21, 17.3
    names, asm, vp, read, amers = _world(tmp_path, k, peak, 87)
    ix, ev = _full(m, k, read, amers, peak)
    n_a = ev.variants("better", vp, names, asm, str(tmp_path / "a.vcf"), log_path=str(tmp_path / "a.log"))
    order = np.argsort(read[0])
    m.db_write_flat(str(tmp_path / "read.mfxk"), k, read[0][order], read[1][order])
    order = np.argsort(amers[0])
    m.db_write_flat(str(tmp_path / "asm.mfxk"), k, amers[0][order], amers[1][order])
    for tag in ("db", "seq", "staged"):
        stages = [m.DbStage(str(tmp_path / (n + ".mfxk"))) for n in ("read", "asm")] if tag == "staged" else []
        assert all(st.ok for st in stages), [st.why for st in stages]      # (sorted k-mers are written as delta-coded blocks: stageable)
        loaded = m.LoadedVcf(vp)
        loaded.prepare(k, "better", names, asm)
        px = m.Index.for_seq(k, loaded.path_bound() + 1024)
        loaded.claim_paths(px)
        if tag == "staged":                                    # the CLI's form: both databases on their way since the process started
            px.load_db_staged(stages[1], 1)
            px.load_db_staged(stages[0], 0)
            for st in stages:
                st.close()
        else:
            if tag == "db":
                px.load_db(str(tmp_path / "asm.mfxk"), 1)
            else:
                px.count_claimed(m.Sequences(asm))
            px.load_db(str(tmp_path / "read.mfxk"), 0)
        pev = m.Evaluator(px, m.KParams(peak))
        n_b = pev.variants_loaded("better", loaded, names, asm, str(tmp_path / (tag + ".vcf")), log_path=str(tmp_path / (tag + ".log")))
        assert n_a == n_b
        assert open(tmp_path / "a.vcf", "rb").read() == open(tmp_path / (tag + ".vcf"), "rb").read(), tag
        loaded.close()


@pytest.mark.gpu
@pytest.mark.parametrize("seed", list(range(300, 312)))
def test_randomized_call_sets_on_the_path_only_index_match_the_oracle(tmp_path, seed):
    """seeded sweep (mode, k odd / even, -comb, -nosplit, peak, burst density, contig sizes): the records and the log of the run on the
    path-only index against the ORACLE's restatement of vcf.C / merfin-variants.C / varMer.C on the full lookup tables"""
    import merfin_amd as m
    from oracle import pyoracle as po
    r = np.random.default_rng(seed)
    mode = ["filter", "polish", "better", "strict", "loose"][int(r.integers(0, 5))]
    k = int(r.choice([9, 12, 15, 21, 22, 27, 31]))
    comb, nosplit, peak = int(r.integers(2, 17)), bool(r.random() < 0.3), float(r.choice([9.0, 17.3, 26.0]))
    names, asm, vp, read, amers = _world(tmp_path, k, peak, seed, burst=float(r.choice([0.03, 0.08, 0.15])),
                                         sizes=tuple(int(x) for x in r.choice([300, 2500, 6000, 12000], size=int(r.integers(2, 5)))))
    n_o = po.variants_run(po.Params(k, peak), po.Lookup(k, *read), po.Lookup(k, *amers), mode, vp, names, asm, str(tmp_path / "o.vcf"), comb=comb, nosplit=nosplit,
                          log_path=str(tmp_path / "o.log"))
    loaded = m.LoadedVcf(vp)
    px = loaded.prepare_path_index(k, mode, names, asm, comb=comb, nosplit=nosplit)
    assert px is not None
    px.add_read(*read)                                         # (either order: both only update what the paths claimed)
    px.add_asm(*amers)
    pev = m.Evaluator(px, m.KParams(peak))
    n_g = pev.variants_loaded(mode, loaded, names, asm, str(tmp_path / "g.vcf"), comb=comb, nosplit=nosplit, log_path=str(tmp_path / "g.log"))
    assert n_o == n_g
    assert open(tmp_path / "g.vcf").read() == open(tmp_path / "o.vcf").read()
    special = lambda p: sorted(l for l in open(p).read().splitlines() if l.startswith("PANIC") or l.startswith("[ WARNING ]"))
    assert special(tmp_path / "g.log") == special(tmp_path / "o.log")
    loaded.close()
