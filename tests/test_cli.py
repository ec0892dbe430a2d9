"""The C++ `merfin` CLI: flag surface on CPU (no GPU needed to fail validation),
and end-to-end -hist / -dump / -completeness against the oracle on the GPU."""
import gzip
import os
import subprocess

import numpy as np
import pytest

from oracle import pyoracle as po
from tests import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "merfin_amd", "bin", "merfin")


def run(args, **kw):
    return subprocess.run([EXE] + args, capture_output=True, text=True, **kw)


def test_cli_validation_messages():
    assert os.path.exists(EXE), "build the CLI with `make -C merfin_amd/cli`"
    r = run(["-hist"])
    assert r.returncode == 1
    for msg in ("No input sequences (-sequence) supplied.", "No output (-output) supplied.",
                "No haploid peak (-peak) supplied.", "No read meryl database (-readmers) supplied."):
        assert msg in r.stderr
    r = run(["-bogus", "-completeness", "-readmers", "x", "-peak", "3"])
    assert r.returncode == 1 and "Unknown option '-bogus'." in r.stderr
    assert "No sequence meryl database (-seqmers) nor sequence (-sequence) supplied." in r.stderr
    r = run([])
    assert r.returncode == 1 and "No report type" in r.stderr
    r = run(["-polish", "-sequence", "a", "-output", "o", "-readmers", "r", "-peak", "3"])
    assert r.returncode == 1 and "No variant call input (-vcf)" in r.stderr


def test_cli_prob_table_through_the_compressed_reader(tmp_path, golden_dir):
    """load_Kmetric reads -prob through compressedFileReader (merfin-globals.C:34): a .gz table parses like the plain one (the rows
    are echoed on stderr before the databases are opened, :55), a damaged .gz is an error, a missing file keeps the reference's text"""
    prob = os.path.join(golden_dir, "example_lookup_table.txt")
    gzp = str(tmp_path / "table.txt.gz")
    with open(prob, "rb") as f, gzip.open(gzp, "wb") as g:
        g.write(f.read())
    base = ["-hist", "-sequence", str(tmp_path / "none.fasta"), "-readmers", str(tmp_path / "none.meryl"), "-peak", "26", "-output", str(tmp_path / "o")]
    plain = run(base + ["-prob", prob])
    zipped = run(base + ["-prob", gzp])
    rows = lambda r: [l for l in r.stderr.splitlines() if l.startswith("Copy-number:")]
    assert len(rows(plain)) == 184 and rows(zipped) == rows(plain)
    assert "Copy-number: 1\t\tReadK: 0\tProbability: " in zipped.stderr
    bad = str(tmp_path / "bad.txt.gz")
    with open(bad, "wb") as f:
        f.write(open(gzp, "rb").read()[:200])
    r = run(base + ["-prob", bad])
    assert r.returncode == 1 and "reading the probability table (-prob)" in r.stderr
    r = run(base + ["-prob", str(tmp_path / "missing.txt.gz")])
    assert r.returncode == 1 and "Probability table (-prob) file" in r.stderr and "doesn't exist!" in r.stderr


def _write_fasta(path, contigs, width=60, gz=False):
    op = gzip.open if gz else open
    with op(path, "wb") as f:
        for i, c in enumerate(contigs):
            f.write(b">ctg%d some description\n" % i)
            for o in range(0, len(c), width):
                f.write(c[o:o + width] + b"\n")


def _write_text_db(path, k, kmers, values):
    dec = np.array(list(b"ACTG"), dtype=np.uint8)
    with open(path, "w") as f:
        for km, v in zip(kmers.tolist(), values.tolist()):
            s = bytes(dec[[(km >> (2 * (k - 1 - i))) & 3 for i in range(k)]]).decode()
            f.write("%s\t%d\n" % (s, v))


@pytest.mark.gpu
def test_cli_hist_dump_completeness_end_to_end(tmp_path, golden_dir):
    import merfin_amd as m
    k, peak = 21, 26.0
    contigs, read, asm = synth.world(k=k, peak=peak, seed=41, sizes=(25000, 6000, 4096, 300, 10))
    prob = os.path.join(golden_dir, "example_lookup_table.txt")
    K, P = po.load_kmetric(prob)
    p = po.Params(k, peak, K, P)
    R, A = po.Lookup(k, *read), po.Lookup(k, *asm)
    g, ka, km, _ = po.hist_run(p, R, A, contigs, threads=2)
    po.report_histogram(p, g, str(tmp_path / "o.hist"), str(tmp_path / "o.sum"))
    fa = str(tmp_path / "asm.fasta.gz")
    _write_fasta(fa, contigs, gz=True)
    m.db_write_flat(str(tmp_path / "read.mfxk"), k, *read)
    _write_text_db(str(tmp_path / "asm.txt"), k, *asm)

    # -hist with -seqmers given as `meryl print` text
    r = run(["-hist", "-sequence", fa, "-readmers", str(tmp_path / "read.mfxk"), "-seqmers", str(tmp_path / "asm.txt"),
             "-peak", str(peak), "-prob", prob, "-output", str(tmp_path / "g.hist")])
    assert r.returncode == 0, r.stderr
    assert (tmp_path / "g.hist").read_bytes() == (tmp_path / "o.hist").read_bytes()
    assert (tmp_path / "o.sum").read_text() in r.stderr
    cum = 0
    for i in range(len(contigs)):
        cum += int(km[i])
        if ka[i] == 0:      # histoQV(0,0) is NaN; C prints it as "-nan" or "nan"
            assert any("ctg%d\t0\t%d\t0\t%s\n" % (i, cum, t) in r.stderr for t in ("nan", "-nan"))
            continue
        qv = po.histoQV(float(km[i]), float(ka[i]), k)
        assert "ctg%d\t%d\t%d\t%d\t%.2f\n" % (i, km[i], cum, ka[i], qv) in r.stderr
    assert r.stderr.rstrip().endswith("Bye!")

    # -hist without -seqmers: the assembly k-mers are counted on the GPU (replaces `meryl count`)
    with open(prob, "rb") as f, gzip.open(str(tmp_path / "prob.gz"), "wb") as gzf:
        gzf.write(f.read())                                    # -prob through the compressed reader (merfin-globals.C:34)
    r2 = run(["-hist", "-sequence", fa, "-readmers", str(tmp_path / "read.mfxk"), "-peak", str(peak), "-prob", str(tmp_path / "prob.gz"),
              "-output", str(tmp_path / "g2.hist")])
    assert r2.returncode == 0, r2.stderr
    assert (tmp_path / "g2.hist").read_bytes() == (tmp_path / "o.hist").read_bytes()

    # -dump
    for ci, c in enumerate(contigs):
        rk, ak_, km_, _, _ = po.process_dump(p, R, A, c)
        po.output_dump(str(tmp_path / "o.dump"), "ctg%d" % ci, rk, ak_, km_, append=ci > 0)
    r3 = run(["-dump", "-sequence", fa, "-readmers", str(tmp_path / "read.mfxk"), "-seqmers", str(tmp_path / "asm.txt"),
              "-peak", str(peak), "-prob", prob, "-output", str(tmp_path / "g.dump")])
    assert r3.returncode == 0, r3.stderr
    assert (tmp_path / "g.dump").read_bytes() == (tmp_path / "o.dump").read_bytes()
    # -skipMissing: no dump file at all (merfin-dump.C:34,81-87), counts still printed
    r4 = run(["-dump", "-skipMissing", "-sequence", fa, "-readmers", str(tmp_path / "read.mfxk"), "-seqmers", str(tmp_path / "asm.txt"),
              "-peak", str(peak), "-output", str(tmp_path / "g4.dump")])
    assert r4.returncode == 0 and not (tmp_path / "g4.dump").exists()
    assert "ctg0\t" in r4.stderr

    # -completeness
    tot = und = 0.0
    for piece in range(64):
        lo, hi = piece << (2 * k - 6), (piece + 1) << (2 * k - 6)
        rs, as_ = (read[0] >= lo) & (read[0] < hi), (asm[0] >= lo) & (asm[0] < hi)
        t, u = po.completeness_piece(po.Params(k, peak), read[0][rs], read[1][rs], asm[0][as_], asm[1][as_])
        tot += t
        und += u
    r5 = run(["-completeness", "-readmers", str(tmp_path / "read.mfxk"), "-seqmers", str(tmp_path / "asm.txt"), "-peak", str(peak)])
    assert r5.returncode == 0, r5.stderr
    assert "TOTAL readK:   %15.2f\n" % tot in r5.stderr
    assert "TOTAL undrcpy:    %15.5f\n" % und in r5.stderr
    assert "COMPLETENESS:             %0.5f\n" % (1.0 - und / tot) in r5.stderr
    assert r5.stderr.count("thread ") == 64 and "thread  0 total " in r5.stderr        # merfin-completeness.C:119-120

    # -memory gate (merfin-globals.C:148-153)
    r6 = run(["-hist", "-sequence", fa, "-readmers", str(tmp_path / "read.mfxk"), "-peak", "26", "-memory", "0.0001",
              "-output", str(tmp_path / "x")])
    assert r6.returncode == 1 and "Not enough memory to load databases.  Increase -memory." in r6.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["polish", "filter", "loose"])
def test_cli_variant_modes(tmp_path, mode):
    import merfin_amd as m
    k, peak = 21, 17.3
    names, asm, vcf, read, amers = synth.variant_world(k=k, peak=peak, seed=61)
    vp = str(tmp_path / "in.vcf")
    open(vp, "w").write(vcf)
    p = po.Params(k, peak)
    R, A = po.Lookup(k, *read), po.Lookup(k, *amers)
    po.variants_run(p, R, A, mode, vp, names, asm, str(tmp_path / "o.vcf"), comb=8, debug_path=str(tmp_path / "o.dbg"))
    fa = str(tmp_path / "asm.fasta")
    with open(fa, "wb") as f:
        for n, c in zip(names, asm):
            f.write(b">" + n.encode() + b" extra words\n" + c + b"\n")
    m.db_write_flat(str(tmp_path / "read.mfxk"), k, *read)
    m.db_write_flat(str(tmp_path / "asm.mfxk"), k, *amers)
    args = ["-" + mode, "-sequence", fa, "-readmers", str(tmp_path / "read.mfxk"), "-seqmers", str(tmp_path / "asm.mfxk"),
            "-vcf", vp, "-comb", "8", "-debug", "-output", str(tmp_path / "out")]
    if mode != "filter":
        args += ["-peak", str(peak)]
    r = run(args)
    assert r.returncode == 0, r.stderr
    suffix = ".polish.vcf" if mode == "polish" else ".filter.vcf"       # merfin-variants.C:324-327
    assert (tmp_path / ("out" + suffix)).read_text() == (tmp_path / "o.vcf").read_text()
    assert gzip.open(str(tmp_path / "out.00.debug.gz"), "rt").read() == (tmp_path / "o.dbg").read_text()
    assert "Processing sequence ctg0 for variants" in r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("seqmers", [True, False])
@pytest.mark.parametrize("k", [21, 27])
def test_cli_variant_modes_on_the_path_only_index(tmp_path, seqmers, k):
    """MFX_CLI_PATH_INDEX=1 (or full tables that would not fit the device): one slot on one device builds the PATH-ONLY index (the call set
    prepared first, its paths' k-mers claimed meanwhile, both databases update-only; without -seqmers the assembly side is counted over the
    uploaded sequence): the records of the oracle and of the full tables (MFX_CLI_PATH_INDEX=0), byte for byte"""
    import merfin_amd as m
    peak = 17.3
    names, asm, vcf, read, amers = synth.variant_world(k=k, peak=peak, seed=62)
    vp = str(tmp_path / "in.vcf")
    open(vp, "w").write(vcf)
    p = po.Params(k, peak)
    R, A = po.Lookup(k, *read), po.Lookup(k, *amers)
    po.variants_run(p, R, A, "polish", vp, names, asm, str(tmp_path / "o.vcf"))
    fa = str(tmp_path / "asm.fasta")
    with open(fa, "wb") as f:
        for n, c in zip(names, asm):
            f.write(b">" + n.encode() + b"\n" + c + b"\n")
    for name, (kk, vv) in (("read", read), ("asm", amers)):           # sorted: delta-coded files, which the path-only build STAGES into device memory
        o = np.argsort(kk)
        m.db_write_flat(str(tmp_path / (name + ".mfxk")), k, kk[o], vv[o])
    args = ["-polish", "-sequence", fa, "-readmers", str(tmp_path / "read.mfxk"), "-vcf", vp, "-peak", str(peak)]
    if seqmers:
        args += ["-seqmers", str(tmp_path / "asm.mfxk")]
    r = run(args + ["-output", str(tmp_path / "path")], env=dict(os.environ, MFX_CLI_PATH_INDEX="1"))
    assert r.returncode == 0, r.stderr
    assert "Claiming the %d-mers of the variants' paths" % k in r.stderr
    r2 = run(args + ["-output", str(tmp_path / "full")], env=dict(os.environ, MFX_CLI_PATH_INDEX="0"))
    assert r2.returncode == 0, r2.stderr
    assert "variants' paths" not in r2.stderr
    r4 = run(args + ["-output", str(tmp_path / "unstaged")], env=dict(os.environ, MFX_CLI_PATH_INDEX="1", MFX_DB_STAGE="0"))      # the path-only index from the files, unstaged
    assert r4.returncode == 0, r4.stderr
    assert (tmp_path / "unstaged.polish.vcf").read_text() == (tmp_path / "o.vcf").read_text()
    r3 = run(args + ["-debug", "-output", str(tmp_path / "dbg")], env=dict(os.environ, MFX_CLI_PATH_INDEX="1"))      # -debug: host-enumerated text is what is claimed
    assert r3.returncode == 0, r3.stderr
    assert (tmp_path / "dbg.polish.vcf").read_text() == (tmp_path / "o.vcf").read_text()
    want = (tmp_path / "o.vcf").read_text()
    assert (tmp_path / "path.polish.vcf").read_text() == want
    assert (tmp_path / "full.polish.vcf").read_text() == want
    assert len([l for l in want.splitlines() if not l.startswith("#")]) > 20


@pytest.mark.gpu
def test_cli_fastq_input_and_lowercase(tmp_path):
    """-sequence may be FASTQ (merfin.C:195); quality lines must not be parsed as bases."""
    import merfin_amd as m
    k, peak = 21, 17.3
    contigs, read, asm = synth.world(k=k, peak=peak, seed=67, sizes=(9000, 2000, 30))
    p = po.Params(k, peak)
    g, ka, km, _ = po.hist_run(p, po.Lookup(k, *read), po.Lookup(k, *asm), contigs, threads=2)
    po.report_histogram(p, g, str(tmp_path / "o.hist"), None)
    fq = str(tmp_path / "asm.fastq.gz")
    with gzip.open(fq, "wb") as f:
        for i, c in enumerate(contigs):
            f.write(b"@ctg%d desc\n" % i + c + b"\n+\n" + b"I" * len(c) + b"\n")
    m.db_write_flat(str(tmp_path / "read.mfxk"), k, *read)
    r = run(["-hist", "-sequence", fq, "-readmers", str(tmp_path / "read.mfxk"), "-peak", str(peak), "-output", str(tmp_path / "g.hist")])
    assert r.returncode == 0, r.stderr
    assert (tmp_path / "g.hist").read_bytes() == (tmp_path / "o.hist").read_bytes()


@pytest.mark.gpu
def test_cli_min_max_filter_and_peak_only(tmp_path):
    """-min / -max drop read k-mers outside [min, max] at load (merfin.C:199-200 -> merylExactLookup::load);
    -dump too; no -prob: the -peak rounding rule alone (merfin-globals.C:80-89)"""
    import merfin_amd as m
    k, peak = 21, 17.3
    contigs, read, asm = synth.world(k=k, peak=peak, seed=43, sizes=(18000, 5000, 333))
    lo, hi = 4, 40
    assert ((read[1] < lo).sum() > 0) and ((read[1] > hi).sum() > 0)
    p = po.Params(k, peak)
    R, A = po.Lookup(k, read[0], read[1], lo, hi), po.Lookup(k, *asm)
    g, ka, km, _ = po.hist_run(p, R, A, contigs, threads=2)
    g0 = po.hist_run(p, po.Lookup(k, *read), A, contigs, threads=2)[0]
    assert g.kmissing > g0.kmissing                        # the filter changes the answer
    po.report_histogram(p, g, str(tmp_path / "o.hist"), str(tmp_path / "o.sum"))
    for ci, c in enumerate(contigs):
        rk, ak_, km_, _, _ = po.process_dump(p, R, A, c)
        po.output_dump(str(tmp_path / "o.dump"), "ctg%d" % ci, rk, ak_, km_, append=ci > 0)
    fa = str(tmp_path / "asm.fasta")
    _write_fasta(fa, contigs, gz=False)
    m.db_write_flat(str(tmp_path / "read.mfxk"), k, *read)
    m.db_write_flat(str(tmp_path / "asm.mfxk"), k, *asm)
    common = ["-sequence", fa, "-readmers", str(tmp_path / "read.mfxk"), "-seqmers", str(tmp_path / "asm.mfxk"),
              "-peak", str(peak), "-min", str(lo), "-max", str(hi), "-threads", "3"]
    r = run(["-hist"] + common + ["-output", str(tmp_path / "g.hist")])
    assert r.returncode == 0, r.stderr
    assert (tmp_path / "g.hist").read_bytes() == (tmp_path / "o.hist").read_bytes()
    assert (tmp_path / "o.sum").read_text() in r.stderr
    r = run(["-dump"] + common + ["-output", str(tmp_path / "g.dump")])
    assert r.returncode == 0, r.stderr
    assert (tmp_path / "g.dump").read_bytes() == (tmp_path / "o.dump").read_bytes()


@pytest.mark.gpu
def test_cli_index_cache_is_bound_to_its_inputs(tmp_path):
    """-index caches the built HBM table.  The cache must be reused only for the inputs it was built from: the
    iterative polish workflow changes -sequence between runs (asm counts are baked into the table), and -min/-max
    are applied from the table's stored filter.  A stale image is rebuilt, never silently used."""
    import merfin_amd as m
    k, peak = 21, 17.3
    contigs, read, asm = synth.world(k=k, peak=peak, seed=47, sizes=(15000, 4000))
    p = po.Params(k, peak)
    R = po.Lookup(k, *read)

    def oracle_text(ctgs, name, lo=0, hi=2**64 - 1):
        ak, av = po.count_kmers(k, ctgs)
        g = po.hist_run(p, po.Lookup(k, read[0], read[1], lo, hi), po.Lookup(k, ak, av), ctgs, threads=2)[0]
        po.report_histogram(p, g, str(tmp_path / name), str(tmp_path / (name + ".sum")))
        return (tmp_path / name).read_bytes()

    fa, img = str(tmp_path / "asm.fasta"), str(tmp_path / "cache.mfxi")
    _write_fasta(fa, contigs)
    m.db_write_flat(str(tmp_path / "read.mfxk"), k, *read)
    base = ["-hist", "-sequence", fa, "-readmers", str(tmp_path / "read.mfxk"), "-peak", str(peak), "-index", img]
    want = oracle_text(contigs, "o1.hist")
    r = run(base + ["-output", str(tmp_path / "g1.hist")])
    assert r.returncode == 0 and "Writing the index image" in r.stderr, r.stderr
    assert (tmp_path / "g1.hist").read_bytes() == want
    r = run(base + ["-output", str(tmp_path / "g2.hist")])           # same inputs: the image is used
    assert r.returncode == 0 and "Loading the index image" in r.stderr and "rebuilding" not in r.stderr, r.stderr
    assert (tmp_path / "g2.hist").read_bytes() == want
    # a "polished" assembly under the same file name
    polished = [bytes(c) for c in synth.mutate(synth.rng(5), [np.frombuffer(c, dtype=np.uint8) for c in contigs], sub_rate=3e-3)]
    _write_fasta(fa, polished)
    want2 = oracle_text(polished, "o3.hist")
    assert want2 != want
    r = run(base + ["-output", str(tmp_path / "g3.hist")])
    assert r.returncode == 0 and "rebuilding" in r.stderr and "sequence file changed" in r.stderr, r.stderr
    assert (tmp_path / "g3.hist").read_bytes() == want2
    # other -min/-max than the image's
    want3 = oracle_text(polished, "o4.hist", 4, 40)
    r = run(base + ["-min", "4", "-max", "40", "-output", str(tmp_path / "g4.hist")])
    assert r.returncode == 0 and "rebuilding" in r.stderr and "-min/-max differ" in r.stderr, r.stderr
    assert (tmp_path / "g4.hist").read_bytes() == want3


def test_cli_device_list_validation():
    r = run(["-hist", "-devices", "0-x", "-sequence", "a", "-output", "o", "-readmers", "r", "-peak", "3"])
    assert r.returncode == 1 and "Invalid device list '0-x'" in r.stderr
    r = run(["-hist", "-devices", "3-1", "-sequence", "a", "-output", "o", "-readmers", "r", "-peak", "3"])
    assert r.returncode == 1 and "Invalid device list" in r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("devices", ["0,0", "0-0,0,0"])
def test_cli_hist_on_several_devices_of_one_process(tmp_path, golden_dir, devices):
    """`merfin -hist -devices ...`: one process drives N evaluation contexts (here all on GPU 0, the box has one);
    histogram file and summary byte-identical to the committed golden fixture (= the single-device output)."""
    g = lambda n: os.path.join(golden_dir, n)
    args = ["-hist", "-sequence", g("case1.fasta"), "-readmers", g("case1.read.kmers.txt"), "-peak", "17.3",
            "-prob", g("example_lookup_table.txt"), "-output", str(tmp_path / "m.hist"), "-devices", devices]
    r = run(args)
    assert r.returncode == 0, r.stderr
    assert "-- Evaluating on %d devices." % (2 if devices == "0,0" else 3) in r.stderr
    r1 = run(args[:-2] + ["-output", str(tmp_path / "s.hist")])
    assert r1.returncode == 0, r1.stderr
    assert (tmp_path / "m.hist").read_bytes() == (tmp_path / "s.hist").read_bytes() == open(g("case1.hist"), "rb").read()
    tail = lambda s: s[s.index("K-mers not found in reads"):]
    assert tail(r.stderr) == tail(r1.stderr)


@pytest.mark.gpu
def test_cli_sequence_read_overlaps_the_index_build(tmp_path):
    """Without -seqmers the k-mer table is sized from a bound the sequence FILE gives before it is read (plain: its size;
    .gz: the ISIZE trailer), so that the read database loads while the file is read.  Same histogram with the overlap,
    without it, from .gz (where it is the default; MFX_CLI_OVERLAP forces either), and from a .gz of two members whose trailer under-reports the size (the
    table is then rebuilt with the true number)."""
    import merfin_amd as m
    k, peak = 21, 17.3
    contigs, read, asm = synth.world(k=k, peak=peak, seed=71, sizes=(30000, 90000, 4000))
    p = po.Params(k, peak)
    g, ka, km, _ = po.hist_run(p, po.Lookup(k, *read), po.Lookup(k, *asm), contigs, threads=2)
    po.report_histogram(p, g, str(tmp_path / "o.hist"), None)
    want = (tmp_path / "o.hist").read_bytes()
    m.db_write_flat(str(tmp_path / "read.mfxk"), k, *read)
    plain, gz1, gz2 = str(tmp_path / "a.fasta"), str(tmp_path / "a1.fasta.gz"), str(tmp_path / "a2.fasta.gz")
    _write_fasta(plain, contigs, gz=False)
    _write_fasta(gz1, contigs, gz=True)
    with open(gz2, "wb") as f:                               # two gzip members: ISIZE describes only the second
        f.write(gzip.compress(b">ctg0\n" + contigs[0] + b"\n"))
        f.write(gzip.compress(b">ctg1\n" + contigs[1] + b"\n>ctg2\n" + contigs[2] + b"\n"))
    base = ["-hist", "-readmers", str(tmp_path / "read.mfxk"), "-peak", str(peak)]
    # (-hist builds a sequence-only index by default, which needs the sequence first: the overlap belongs to the full
    # tables, MFX_CLI_FULL_INDEX=1; the last case is the default -hist on the two-member .gz)
    full = {"MFX_CLI_FULL_INDEX": "1"}
    for i, (fa, env, note) in enumerate([(plain, dict(full, MFX_CLI_OVERLAP="1"), False), (plain, full, False), (gz1, full, False),
                                         (gz1, dict(full, MFX_CLI_OVERLAP="0"), False), (gz2, full, True), (gz2, {}, False)]):
        out = str(tmp_path / ("g%d.hist" % i))
        r = subprocess.run([EXE] + base + ["-sequence", fa, "-output", out], capture_output=True, text=True, env=dict(os.environ, MFX_CLI_TIMING="1", **env))
        assert r.returncode == 0, r.stderr
        assert open(out, "rb").read() == want, (fa, env)
        assert ("more than its size promised" in r.stderr) == note, r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("seed", list(range(int(os.environ.get("MFX_RANDOM_SEEDS", "8")))))
def test_randomized_fasta_layouts(tmp_path, seed):
    """seeded sweep over how the same contigs can be laid out in a FASTA file -- line width (1 ... one line per record),
    CR LF, blank lines, empty records, descriptions, no newline at the end, records straddling the reader's 4 MB buffer,
    .gz -- the CLI's -hist must be the oracle's for the contigs, and the per-contig stderr lines must name them in order"""
    import merfin_amd as m
    r = np.random.default_rng(4000 + seed)
    k, peak = 21, 17.3
    sizes = tuple(int(x) for x in r.choice([0, 1, 20, 21, 400, 4096, 30000], size=int(r.integers(1, 7))))
    if seed % 4 == 0:
        sizes += (int(r.integers(4_200_000, 4_400_000)),)          # longer than the read buffer
    contigs, read, asm = synth.world(k=k, peak=peak, seed=4100 + seed, sizes=sizes)
    p = po.Params(k, peak)
    g, ka, km, _ = po.hist_run(p, po.Lookup(k, *read), po.Lookup(k, *asm), contigs, threads=2)
    po.report_histogram(p, g, str(tmp_path / "o.hist"), None)
    eol = b"\r\n" if r.random() < 0.3 else b"\n"
    body = b""
    for i, c in enumerate(contigs):
        body += b">ctg%d%s" % (i, (b" len=%d extra" % len(c)) if r.random() < 0.5 else b"") + eol
        width = int(r.choice([1, 7, 60, 61, 4096, 10**9])) if len(c) < 100000 else int(r.choice([60, 80, 10**9]))
        for o in range(0, len(c), width):
            body += c[o:o + width] + eol
        if r.random() < 0.3:
            body += eol                                               # a blank line between records
    if r.random() < 0.4 and body.endswith(eol):
        body = body[:-len(eol)]                                       # no newline at the end of the file
    gz = r.random() < 0.3
    fa = str(tmp_path / ("a.fasta.gz" if gz else "a.fasta"))
    (gzip.open if gz else open)(fa, "wb").write(body)
    m.db_write_flat(str(tmp_path / "read.mfxk"), k, *read)
    rr = run(["-hist", "-sequence", fa, "-readmers", str(tmp_path / "read.mfxk"), "-peak", str(peak), "-output", str(tmp_path / "g.hist")])
    assert rr.returncode == 0, rr.stderr
    assert (tmp_path / "g.hist").read_bytes() == (tmp_path / "o.hist").read_bytes()
    named = [l.split("\t")[0] for l in rr.stderr.splitlines() if l.count("\t") == 4 and l.startswith("ctg")]
    assert named == ["ctg%d" % i for i in range(len(contigs))]
    if not gz:
        # plain files of 1 MB and more are read by all host threads (read_fasta_parallel): the same file through that reader
        # (pieces of 4099 bytes: CR LF pairs and lines straddle them) and through the sequential one
        for env in ({"MFX_CLI_SEQ_PAR_MIN": "0", "MFX_CLI_SEQ_SLICE": "4099", "MFX_CLI_SEQ_THREADS": "7"}, {"MFX_CLI_SEQ_THREADS": "1"}):
            r2 = run(["-hist", "-sequence", fa, "-readmers", str(tmp_path / "read.mfxk"), "-peak", str(peak), "-output", str(tmp_path / "g2.hist")],
                     env=dict(os.environ, **env))
            assert r2.returncode == 0, r2.stderr
            assert (tmp_path / "g2.hist").read_bytes() == (tmp_path / "o.hist").read_bytes(), env
            assert [l.split("\t")[0] for l in r2.stderr.splitlines() if l.count("\t") == 4 and l.startswith("ctg")] == named


@pytest.mark.gpu
def test_cli_hist_falls_back_to_the_full_tables_for_a_non_canonical_database(tmp_path):
    """`merfin -hist / -dump` build a sequence-only index (one slot per canonical k-mer of -sequence).  A database that holds
    forward (non-canonical) k-mers cannot be answered from it -- value(fmer) + value(rmer) needs both strands' slots
    (merfin-globals.C:107-108) -- so the load reports it and the CLI builds the full tables: same result as the oracle."""
    import merfin_amd as m
    k, peak = 11, 5.0
    r = synth.rng(3)
    contigs = synth.as_bytes(synth.decorate(r, synth.make_truth(r, (6000, 2500))))
    fw = {}
    for c in contigs:
        for _, f, _r in po.kiter(k, c):
            fw[f] = fw.get(f, 0) + 1
    ak = np.array(sorted(fw), dtype=np.uint64)
    av = np.array([fw[x] for x in ak.tolist()], dtype=np.uint32)
    rv = (av * 5 + (ak % 3).astype(np.uint32)).astype(np.uint32)
    p = po.Params(k, peak)
    g, ka, km, _ = po.hist_run(p, po.Lookup(k, ak, rv), po.Lookup(k, ak, av), contigs, threads=2)
    po.report_histogram(p, g, str(tmp_path / "o.hist"), None)
    m.db_write_flat(str(tmp_path / "read.mfxk"), k, ak, rv)
    m.db_write_flat(str(tmp_path / "asm.mfxk"), k, ak, av)
    fa = str(tmp_path / "a.fasta")
    _write_fasta(fa, contigs, gz=False)
    for seqmers in (True, False):
        args = ["-hist", "-sequence", fa, "-readmers", str(tmp_path / "read.mfxk"), "-peak", str(peak), "-output", str(tmp_path / "g.hist")]
        if seqmers:
            args += ["-seqmers", str(tmp_path / "asm.mfxk")]
        rr = subprocess.run([EXE] + args, capture_output=True, text=True)
        assert rr.returncode == 0, rr.stderr
        assert "not canonical; building the full lookup tables" in rr.stderr
        if seqmers:                     # (without -seqmers the assembly side is the canonical count of the sequence: another world)
            assert (tmp_path / "g.hist").read_bytes() == (tmp_path / "o.hist").read_bytes()
            # -index: the fallback saves the FULL tables under the full-index digest, and the next run takes that image instead
            # of building twice again (advisor, round 3: it used to be saved under the sequence-only digest and never matched)
            img = str(tmp_path / "nc.mfxi")
            for attempt in (0, 1):
                (tmp_path / "g.hist").unlink()
                rr = subprocess.run([EXE] + args + ["-index", img], capture_output=True, text=True)
                assert rr.returncode == 0, rr.stderr
                assert ("Loading the index image" in rr.stderr) == (attempt == 1), rr.stderr
                assert ("building the full lookup tables" in rr.stderr) == (attempt == 0) and "rebuilding it" not in rr.stderr
                assert (tmp_path / "g.hist").read_bytes() == (tmp_path / "o.hist").read_bytes()


@pytest.mark.gpu
def test_cli_index_cache_of_a_sequence_only_index(tmp_path, golden_dir):
    """-index with -hist caches the sequence-only table; -completeness on the same databases must not take it (it needs every
    read k-mer) and rebuilds the full one"""
    g = lambda n: os.path.join(golden_dir, n)
    img = str(tmp_path / "c.mfxi")
    common = ["-sequence", g("case1.fasta"), "-readmers", g("case1.read.kmers.txt"), "-seqmers", g("case1.asm.kmers.txt"), "-peak", "17.3",
              "-prob", g("example_lookup_table.txt"), "-index", img]
    for attempt in (0, 1):
        r = subprocess.run([EXE, "-hist"] + common + ["-output", str(tmp_path / "h")], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        assert (tmp_path / "h").read_bytes() == open(g("case1.hist"), "rb").read()
        assert ("Loading the index image" in r.stderr) == (attempt == 1)
    ref = subprocess.run([EXE, "-completeness"] + common[:-2], capture_output=True, text=True)
    r = subprocess.run([EXE, "-completeness"] + common, capture_output=True, text=True)
    assert r.returncode == 0 and ref.returncode == 0, r.stderr
    assert "rebuilding it" in r.stderr
    tot = lambda s: [l for l in s.splitlines() if l.startswith(("TOTAL", "COMPLETENESS"))]
    assert tot(r.stderr) == tot(ref.stderr) and len(tot(r.stderr)) == 3


@pytest.mark.gpu
def test_cli_convert_then_hist_equals_hist_from_the_text(tmp_path):
    """`merfin -convert` of a `meryl print` text gives a flat database (sorted: delta-coded, decoded by the inserting kernel) that
    answers -hist and -dump exactly as the text does -- which is what the oracle says"""
    k, peak = 21, 17.3
    contigs, read, asm = synth.world(k=k, peak=peak, seed=77, sizes=(20000, 5000, 300))
    p = po.Params(k, peak)
    g, ka, km, _ = po.hist_run(p, po.Lookup(k, *read), po.Lookup(k, *asm), contigs, threads=2)
    po.report_histogram(p, g, str(tmp_path / "o.hist"), None)
    fa = str(tmp_path / "asm.fasta")
    _write_fasta(fa, contigs)
    _write_text_db(str(tmp_path / "read.txt"), k, *read)
    r = run(["-convert", str(tmp_path / "read.txt"), "-output", str(tmp_path / "read.mfxk")])
    assert r.returncode == 0 and "Wrote %d k-mers" % len(read[0]) in r.stderr, r.stderr
    outs = []
    for db in ("read.txt", "read.mfxk"):
        o = str(tmp_path / (db + ".hist"))
        r = run(["-hist", "-sequence", fa, "-readmers", str(tmp_path / db), "-peak", str(peak), "-output", o])
        assert r.returncode == 0, r.stderr
        outs.append(open(o, "rb").read())
    assert outs[0] == outs[1] == (tmp_path / "o.hist").read_bytes()


@pytest.mark.gpu
def test_cli_slots_of_one_device_with_the_packed_transport(tmp_path):
    """-dump, -hist and -polish in 4 and 8 slots on one device, the assembly uploaded PACKED (what an assembly of 8 MB or more
    takes; forced here): the slots share one mfx_seq and one null stream -- the outputs must be the single slot's, byte for
    byte, run after run (at 64 Mb the -dump of several slots came out empty in 5 of 6 runs before what a sequence makes on
    first use was made under a lock: profiles/r04_slots_soak.txt)"""
    import merfin_amd as m
    k, peak = 21, 17.3
    names, asm, vcf, read, amers = synth.variant_world(k=k, peak=peak, seed=67, sizes=(40000, 30000, 20000, 12000, 9000, 5000, 4097, 300))
    vp = str(tmp_path / "in.vcf")
    open(vp, "w").write(vcf)
    fa = str(tmp_path / "asm.fasta")
    with open(fa, "wb") as f:
        for n, c in zip(names, asm):
            f.write(b">" + n.encode() + b"\n" + c + b"\n")
    m.db_write_flat(str(tmp_path / "read.mfxk"), k, *read)
    m.db_write_flat(str(tmp_path / "asm.mfxk"), k, *amers)
    common = ["-sequence", fa, "-readmers", str(tmp_path / "read.mfxk"), "-seqmers", str(tmp_path / "asm.mfxk"), "-peak", str(peak)]
    env = dict(os.environ, MFX_UPLOAD_PACKED_MIN="1")
    for op, extra, suffix in (("-dump", [], ""), ("-hist", [], ""), ("-polish", ["-vcf", vp], ".polish.vcf")):
        ref = None
        for devices in ("0", "0,0,0,0", "0,0,0,0,0,0,0,0"):
            for rep in range(1 if devices == "0" else 4):
                out = str(tmp_path / ("o%s_%d_%d" % (op, len(devices), rep)))
                r = run([op] + common + extra + ["-devices", devices, "-output", out], env=env)
                assert r.returncode == 0, r.stderr
                data = open(out + suffix, "rb").read()
                if ref is None:
                    ref = data
                    assert len(ref) > (1000 if op != "-hist" else 10)
                assert data == ref, "%s -devices %s, repetition %d differs from the single slot" % (op, devices, rep)
