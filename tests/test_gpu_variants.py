"""-filter / -polish / -better / -strict / -loose: the GPU-scored path against
the oracle's line-by-line restatement of vcf.C / merfin-variants.C / varMer.C.
Outputs are text and must be byte-identical (VCF records, -debug statistics)."""
import os

import numpy as np
import pytest

from oracle import pyoracle as po
from tests import synth

MODES = ["filter", "polish", "better", "strict", "loose"]


def _run_both(tmp_path, mode, k=21, peak=17.3, seed=synth.SEED, comb=15, nosplit=False, use_prob=False, golden_dir=None, **kw):
    import merfin_amd as m
    names, asm, vcf, read, amers = synth.variant_world(k=k, peak=peak, seed=seed, **kw)
    vp = str(tmp_path / "in.vcf")
    open(vp, "w").write(vcf)
    K = P = None
    if use_prob:
        K, P = po.load_kmetric(os.path.join(golden_dir, "example_lookup_table.txt"))
    p = po.Params(k, peak, K, P)
    R, A = po.Lookup(k, *read), po.Lookup(k, *amers)
    tag = "%s_%d" % (mode, seed)
    o_out, o_dbg, o_log = [str(tmp_path / ("o_%s.%s" % (tag, e))) for e in ("vcf", "dbg", "log")]
    g_out, g_dbg, g_log = [str(tmp_path / ("g_%s.%s" % (tag, e))) for e in ("vcf", "dbg", "log")]
    n_o = po.variants_run(p, R, A, mode, vp, names, asm, o_out, comb=comb, nosplit=nosplit, debug_path=o_dbg, log_path=o_log)
    ix = m.Index(k, len(read[0]) + len(amers[0]) + 16)
    ix.add_read(*read)
    ix.add_asm(*amers)
    ev = m.Evaluator(ix, m.KParams(peak, K, P))
    n_g = ev.variants(mode, vp, names, asm, g_out, comb=comb, nosplit=nosplit, debug_path=g_dbg, log_path=g_log)
    return (n_o, o_out, o_dbg, o_log), (n_g, g_out, g_dbg, g_log)


def _special(path):
    return sorted(l for l in open(path).read().splitlines() if l.startswith("PANIC") or l.startswith("[ WARNING ]"))


@pytest.mark.gpu
@pytest.mark.parametrize("mode", MODES)
def test_variant_modes_match_oracle(tmp_path, mode):
    (n_o, o_out, o_dbg, o_log), (n_g, g_out, g_dbg, g_log) = _run_both(tmp_path, mode, comb=10)
    assert n_o == n_g and n_o > 50
    out = open(o_out).read()
    assert open(g_out).read() == out
    body = [l for l in out.splitlines() if not l.startswith("#")]
    assert len(body) > 50                       # the mode actually selected records
    assert open(g_dbg).read() == open(o_dbg).read()
    assert os.path.getsize(o_dbg) > 20000
    assert _special(g_log) == _special(o_log)


@pytest.mark.gpu
@pytest.mark.parametrize("mode,seed,comb,nosplit,k", [("polish", 7, 4, False, 21), ("polish", 8, 12, True, 21), ("loose", 9, 6, False, 15),
                                                      ("filter", 10, 15, False, 31), ("strict", 11, 3, True, 11)])
def test_variant_options_and_k(tmp_path, mode, seed, comb, nosplit, k, golden_dir):
    (n_o, o_out, o_dbg, o_log), (n_g, g_out, g_dbg, g_log) = _run_both(tmp_path, mode, k=k, seed=seed, comb=comb, nosplit=nosplit,
                                                                      use_prob=(seed % 2 == 1), golden_dir=golden_dir, peak=26.0)
    assert n_o == n_g
    assert open(g_out).read() == open(o_out).read()
    assert open(g_dbg).read() == open(o_dbg).read()
    assert _special(g_log) == _special(o_log)
    if nosplit:
        assert any(l.startswith("PANIC : Combination") for l in _special(o_log))


@pytest.mark.gpu
def test_polish_fixes_the_assembly(tmp_path):
    """Sanity of the whole idea (not a parity claim): applying -polish's 1/1 records
    removes most of the k-mers that were missing from the reads."""
    (n_o, o_out, _, _), _ = _run_both(tmp_path, "polish", seed=21, decoys=0)
    recs = [l.split("\t") for l in open(o_out).read().splitlines() if not l.startswith("#")]
    assert sum(1 for r in recs if r[9] == "1/1") > 100


@pytest.mark.gpu
@pytest.mark.parametrize("seed", list(range(100, 100 + int(os.environ.get("MFX_RANDOM_SEEDS", "16")))))
def test_randomized_variant_runs_match_oracle(tmp_path, seed, golden_dir):
    """seeded sweep: mode, k (odd/even), -comb, -nosplit, -prob, peak, burst density -- VCF, -debug and PANIC/WARNING lines
    byte-identical to the oracle's restatement every time"""
    r = np.random.default_rng(seed)
    mode = MODES[int(r.integers(0, len(MODES)))]
    k = int(r.choice([9, 12, 15, 21, 22, 27, 31]))
    (n_o, o_out, o_dbg, o_log), (n_g, g_out, g_dbg, g_log) = _run_both(
        tmp_path, mode, k=k, seed=seed, comb=int(r.integers(2, 17)), nosplit=bool(r.random() < 0.3), use_prob=bool(r.random() < 0.5),
        golden_dir=golden_dir, peak=float(r.choice([9.0, 17.3, 26.0])), burst=float(r.choice([0.03, 0.08, 0.15])),
        sizes=tuple(int(x) for x in r.choice([300, 2500, 6000, 12000], size=int(r.integers(2, 5)))))
    assert n_o == n_g
    assert open(g_out).read() == open(o_out).read()
    assert open(g_dbg).read() == open(o_dbg).read()
    assert _special(g_log) == _special(o_log)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", list(range(200, 200 + int(os.environ.get("MFX_RANDOM_SEEDS", "10")))))
def test_randomized_vcf_records_parse_like_the_oracle(tmp_path, seed):
    """seeded sweep over what a VCF line may look like (vcf.C:93-149, vcfRecord.H:53-97): fewer than 10 columns, several
    samples, GT with phasing / missing alleles / indices past the ALT list, ALT '.', '*', symbolic, lower case, QUAL '.',
    header lines in the middle, unknown CHROM, unsorted positions, duplicate lines -- same records kept, same clusters,
    same output as the oracle"""
    import merfin_amd as m
    r = np.random.default_rng(seed)
    k, peak = 21, 17.3
    names, asm, vcf, read, amers = synth.variant_world(k=k, peak=peak, seed=seed, sizes=(9000, 3000, 500))
    lines = vcf.splitlines()
    head = [l for l in lines if l.startswith("#")]
    body = [l for l in lines if not l.startswith("#")]
    out = []
    for l in body:
        w = l.split("\t")
        roll = r.random() if len(w) >= 10 else 0.5
        if roll < 0.04:
            w = w[:int(r.integers(5, 10))]                            # fewer than 10 columns: excluded
        elif roll < 0.08:
            w.append(r.choice(["0/1", "1/1", "./."]))                  # a second sample (only the first is read)
        elif roll < 0.14:
            w[9] = str(r.choice(["1|0", "0|1", "1", ".", "1/.", "./1", "2/2", "0/0:35", "1/1:12:0.5", "0|0", "1/2"]))
        elif roll < 0.18:
            w[4] = str(r.choice([".", "*", "<DEL>", w[4].lower(), w[4] + ",*", w[3]]))
        elif roll < 0.20:
            w[5] = "."
        elif roll < 0.30:                                             # QUAL spellings: the exact fast path (digits[.digits]) and everything else strtod takes
                                                                      # (within int range: the reference prints (int)QUAL, varMer.C:486)
            w[5] = str(r.choice(["30", "12.5", "1e2", "+5", "007.50", "1234567890.12345678", ".5", "5.", "-3.2", "0.30000000000000004",
                                 "99999999999999.9", "0.05", "2.25", "1.15", "33.35", "1234567.85", "0", "00", "4.450000000000001", "1E-3", "0x10", "12abc"]))
        elif roll < 0.22:
            w[0] = "chrUnknown"
        elif roll < 0.24:
            w[3] = w[3].lower()
        out.append("\t".join(w))
        if roll > 0.97:
            out.append("\t".join(w))                                   # the same line twice
        if 0.95 < roll < 0.96:
            out.append("##a header line in the middle")
    if r.random() < 0.5:                                               # a block moved out of order
        i = int(r.integers(0, max(1, len(out) - 20)))
        out = out[:i] + out[i + 10:i + 20] + out[i:i + 10] + out[i + 20:]
    vp = str(tmp_path / "in.vcf")
    open(vp, "w").write("\n".join(head + out) + "\n")
    mode = MODES[int(r.integers(0, len(MODES)))]
    p = po.Params(k, peak)
    R, A = po.Lookup(k, *read), po.Lookup(k, *amers)
    n_o = po.variants_run(p, R, A, mode, vp, names, asm, str(tmp_path / "o.vcf"), comb=10, debug_path=str(tmp_path / "o.dbg"), log_path=str(tmp_path / "o.log"))
    ix = m.Index(k, len(read[0]) + len(amers[0]) + 16)
    ix.add_read(*read)
    ix.add_asm(*amers)
    n_g = m.Evaluator(ix, m.KParams(peak)).variants(mode, vp, names, asm, str(tmp_path / "g.vcf"), comb=10, debug_path=str(tmp_path / "g.dbg"),
                                                    log_path=str(tmp_path / "g.log"))
    assert n_g == n_o
    assert open(tmp_path / "g.vcf").read() == open(tmp_path / "o.vcf").read()
    assert open(tmp_path / "g.dbg").read() == open(tmp_path / "o.dbg").read()
    assert _special(str(tmp_path / "g.log")) == _special(str(tmp_path / "o.log"))


@pytest.mark.gpu
@pytest.mark.parametrize("mode", MODES)
def test_variants_on_a_vcf_loaded_ahead(tmp_path, mode):
    """mfx_vcf_load + mfx_variants_run_vcf (the VCF read and parsed ahead of the run: what the CLI does under the index build,
    merfin-globals.C:201-219 opens it afterwards) == mfx_variants_run, byte for byte; a handle serves one run"""
    import merfin_amd as m
    k, peak = 21, 17.3
    names, asm, vcf, read, amers = synth.variant_world(k=k, peak=peak, seed=77)
    vp = str(tmp_path / "in.vcf")
    open(vp, "w").write(vcf)
    loaded = m.LoadedVcf(vp)                                   # before any index exists: host work only
    ix = m.Index(k, len(read[0]) + len(amers[0]) + 16)
    ix.add_read(*read)
    ix.add_asm(*amers)
    ev = m.Evaluator(ix, m.KParams(peak))
    n_a = ev.variants(mode, vp, names, asm, str(tmp_path / "a.vcf"), log_path=str(tmp_path / "a.log"))
    n_b = ev.variants_loaded(mode, loaded, names, asm, str(tmp_path / "b.vcf"), log_path=str(tmp_path / "b.log"))
    assert n_a == n_b and n_a > 0
    assert open(tmp_path / "a.vcf", "rb").read() == open(tmp_path / "b.vcf", "rb").read()
    assert open(tmp_path / "a.log", "rb").read() == open(tmp_path / "b.log", "rb").read()
    with pytest.raises(m.MfxError):                            # clustering rearranged it: one run per load
        ev.variants_loaded(mode, loaded, names, asm, str(tmp_path / "c.vcf"), log_path=str(tmp_path / "c.log"))
    loaded.close()


@pytest.mark.gpu
@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("batch_mb", [None, "0"])
def test_variants_prepared_ahead(tmp_path, monkeypatch, mode, batch_mb):
    """mfx_vcf_prepare (clusters merged, combinations enumerated and packed before any index exists: what the CLI runs under its index
    build; merfin does it after load_Kmers, merfin-variants.C:131-230) + mfx_variants_run_vcf == mfx_variants_run, records and log byte
    for byte -- with the default batches and with one cluster per batch (MFX_VAR_BATCH_MB=0: the order of the log's lines across batches)"""
    import merfin_amd as m
    if batch_mb is not None:
        monkeypatch.setenv("MFX_VAR_BATCH_MB", batch_mb)
    k, peak = 21, 17.3
    names, asm, vcf, read, amers = synth.variant_world(k=k, peak=peak, seed=78)
    vp = str(tmp_path / "in.vcf")
    open(vp, "w").write(vcf)
    loaded = m.LoadedVcf(vp)
    loaded.prepare(k, mode, names, asm)                        # no index, no device
    with pytest.raises(m.MfxError):
        loaded.prepare(k, mode, names, asm)                    # once per handle
    ix = m.Index(k, len(read[0]) + len(amers[0]) + 16)
    ix.add_read(*read)
    ix.add_asm(*amers)
    ev = m.Evaluator(ix, m.KParams(peak))
    n_a = ev.variants(mode, vp, names, asm, str(tmp_path / "a.vcf"), log_path=str(tmp_path / "a.log"))
    with pytest.raises(m.MfxError):                            # prepared for -comb 15: another one is refused, the handle stays usable
        ev.variants_loaded(mode, loaded, names, asm, str(tmp_path / "x.vcf"), comb=3, log_path=str(tmp_path / "x.log"))
    n_b = ev.variants_loaded(mode, loaded, names, asm, str(tmp_path / "b.vcf"), log_path=str(tmp_path / "b.log"))
    assert n_a == n_b and n_a > 0
    assert open(tmp_path / "a.vcf", "rb").read() == open(tmp_path / "b.vcf", "rb").read()
    assert open(tmp_path / "a.log", "rb").read() == open(tmp_path / "b.log", "rb").read()
    loaded.close()


@pytest.mark.gpu
def test_variants_prepared_ahead_debug_and_odd_records(tmp_path):
    """the prepared run with -debug (scored on the host from the per-base values) and a call set with clusters beyond -comb and records of
    unknown contigs: same records, -debug lines and log as the unprepared run"""
    import merfin_amd as m
    k, peak = 21, 17.3
    names, asm, vcf, read, amers = synth.variant_world(k=k, peak=peak, seed=79, burst=0.3)
    vp = str(tmp_path / "in.vcf")
    lines = vcf.rstrip("\n").split("\n")
    body = [x for x in lines if not x.startswith("#")]
    extra = []
    for x in body[:5]:
        w = x.split("\t")
        w[0] = "chrUnknown"
        extra.append("\t".join(w))
    open(vp, "w").write("\n".join(lines + extra) + "\n")
    ix = m.Index(k, len(read[0]) + len(amers[0]) + 16)
    ix.add_read(*read)
    ix.add_asm(*amers)
    ev = m.Evaluator(ix, m.KParams(peak))
    for tag, comb in (("d", 15), ("c", 2)):
        loaded = m.LoadedVcf(vp)
        loaded.prepare(k, "polish", names, asm, comb=comb, debug_path=str(tmp_path / "unused"))
        n_a = ev.variants("polish", vp, names, asm, str(tmp_path / (tag + "a.vcf")), comb=comb, debug_path=str(tmp_path / (tag + "a.dbg")),
                          log_path=str(tmp_path / (tag + "a.log")))
        n_b = ev.variants_loaded("polish", loaded, names, asm, str(tmp_path / (tag + "b.vcf")), comb=comb, debug_path=str(tmp_path / (tag + "b.dbg")),
                                 log_path=str(tmp_path / (tag + "b.log")))
        assert n_a == n_b
        for ext in ("vcf", "dbg", "log"):
            assert open(tmp_path / (tag + "a." + ext), "rb").read() == open(tmp_path / (tag + "b." + ext), "rb").read(), (tag, ext)
        loaded.close()


@pytest.mark.gpu
@pytest.mark.parametrize("mode", MODES)
def test_variants_device_traverse_equals_host_enumeration(tmp_path, monkeypatch, mode):
    """the clusters enumerated on the device (mfx_var_traverse_kernel: merfin's traverse, merfin-variants.C:22-126, one cluster per thread,
    the default) == enumerated on the host (MFX_VAR_DEVICE_TRAVERSE=0) == the oracle: records and log byte for byte; with bursts of
    variants (clusters beyond the device's limits go to the host inside the same batch) and one cluster per batch as well"""
    import merfin_amd as m
    k, peak = 21, 17.3
    names, asm, vcf, read, amers = synth.variant_world(k=k, peak=peak, seed=81, burst=0.25)
    vp = str(tmp_path / "in.vcf")
    open(vp, "w").write(vcf)
    p = po.Params(k, peak)
    R, A = po.Lookup(k, *read), po.Lookup(k, *amers)
    n_o = po.variants_run(p, R, A, mode, vp, names, asm, str(tmp_path / "o.vcf"), log_path=str(tmp_path / "o.log"))
    ix = m.Index(k, len(read[0]) + len(amers[0]) + 16)
    ix.add_read(*read)
    ix.add_asm(*amers)
    ev = m.Evaluator(ix, m.KParams(peak))
    outs = {}
    for tag, env in (("dev", {}), ("host", {"MFX_VAR_DEVICE_TRAVERSE": "0"}), ("dev1", {"MFX_VAR_BATCH_MB": "0"}), ("chk", {"MFX_VAR_TRAVERSE_CHECK": "1"})):
        for kk in ("MFX_VAR_DEVICE_TRAVERSE", "MFX_VAR_BATCH_MB", "MFX_VAR_TRAVERSE_CHECK"):
            monkeypatch.delenv(kk, raising=False)
        for kk, vv in env.items():
            monkeypatch.setenv(kk, vv)
        n = ev.variants(mode, vp, names, asm, str(tmp_path / (tag + ".vcf")), log_path=str(tmp_path / (tag + ".log")))
        assert n == n_o
        outs[tag] = open(tmp_path / (tag + ".vcf"), "rb").read()
        assert outs[tag] == open(tmp_path / "o.vcf", "rb").read(), tag
        assert _special(str(tmp_path / (tag + ".log"))) == _special(str(tmp_path / "o.log")), tag
    assert outs["dev"] == outs["host"] == outs["dev1"]


@pytest.mark.gpu
def test_variants_record_past_its_window_fails_the_same_way_with_and_without_the_device_traverse(tmp_path, monkeypatch):
    """a record whose position lies past the end of its contig makes the reference's std::string::replace throw (merfin-variants.C:60); the
    host's enumeration reports that as an error -- and so does the run whose clusters are enumerated on the device: the kernel flags the
    cluster, the batch goes back to the host, the host throws"""
    import merfin_amd as m
    k, peak = 21, 17.3
    names, asm, vcf, read, amers = synth.variant_world(k=k, peak=peak, seed=79, burst=0.3)
    lines = vcf.rstrip("\n").split("\n")
    body = [x for x in lines if not x.startswith("#")]
    w = body[-1].split("\t")
    w[1] = str(len(asm[names.index(w[0])]) + 5)
    vp = str(tmp_path / "in.vcf")
    open(vp, "w").write("\n".join(lines + ["\t".join(w)]) + "\n")
    ix = m.Index(k, len(read[0]) + len(amers[0]) + 16)
    ix.add_read(*read)
    ix.add_asm(*amers)
    ev = m.Evaluator(ix, m.KParams(peak))
    for env in ({}, {"MFX_VAR_DEVICE_TRAVERSE": "0"}):
        monkeypatch.delenv("MFX_VAR_DEVICE_TRAVERSE", raising=False)
        for kk, vv in env.items():
            monkeypatch.setenv(kk, vv)
        with pytest.raises(m.MfxError):
            ev.variants("polish", vp, names, asm, str(tmp_path / "x.vcf"), comb=2, log_path=str(tmp_path / "x.log"))
    # ... and the evaluator is fine afterwards
    monkeypatch.delenv("MFX_VAR_DEVICE_TRAVERSE", raising=False)
    open(vp, "w").write(vcf)
    assert ev.variants("polish", vp, names, asm, str(tmp_path / "y.vcf"), log_path=str(tmp_path / "y.log")) > 0
