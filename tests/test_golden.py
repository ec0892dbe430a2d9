"""Committed golden fixtures (tests/golden/case1.*, made by make_golden.py):
the oracle must keep reproducing them (CPU) and the HIP path -- through the
C++ CLI and the C ABI -- must match them byte for byte (GPU)."""
import os
import subprocess

import numpy as np
import pytest

from oracle import pyoracle as po

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")
EXE = os.path.join(ROOT, "merfin_amd", "bin", "merfin")
K, PEAK, COMB = 21, 17.3, 8


def read_fasta(path):
    names, seqs = [], []
    for line in open(path, "rb"):
        if line.startswith(b">"):
            names.append(line[1:].split()[0].decode())
            seqs.append(b"")
        else:
            seqs[-1] += line.strip()
    return names, seqs


def read_kmers(path):
    code = {"A": 0, "C": 1, "T": 2, "G": 3}
    ks, vs = [], []
    for line in open(path):
        s, v = line.split("\t")
        x = 0
        for ch in s:
            x = (x << 2) | code[ch]
        ks.append(x)
        vs.append(int(v))
    return np.array(ks, dtype=np.uint64), np.array(vs, dtype=np.uint32)


def test_oracle_reproduces_golden(tmp_path):
    names, asm = read_fasta(G + "/case1.fasta")
    read, amers = read_kmers(G + "/case1.read.kmers.txt"), read_kmers(G + "/case1.asm.kmers.txt")
    # the assembly k-mer fixture is what `meryl count` of the FASTA gives
    ak, av = po.count_kmers(K, asm)
    np.testing.assert_array_equal(ak, amers[0])
    np.testing.assert_array_equal(av, amers[1])
    probK, probP = po.load_kmetric(G + "/example_lookup_table.txt")
    p = po.Params(K, PEAK, probK, probP)
    R, A = po.Lookup(K, *read), po.Lookup(K, *amers)
    g, ka, km, _ = po.hist_run(p, R, A, asm, threads=2)
    po.report_histogram(p, g, str(tmp_path / "h"), str(tmp_path / "s"))
    assert (tmp_path / "h").read_bytes() == open(G + "/case1.hist", "rb").read()
    assert (tmp_path / "s").read_bytes() == open(G + "/case1.summary", "rb").read()
    for i, (n, c) in enumerate(zip(names, asm)):
        rk, akk, kmm, _, _ = po.process_dump(p, R, A, c)
        po.output_dump(str(tmp_path / "d"), n, rk, akk, kmm, append=i > 0)
    assert (tmp_path / "d").read_bytes() == open(G + "/case1.dump", "rb").read()
    for mode in ("polish", "filter", "loose"):
        po.variants_run(p, R, A, mode, G + "/case1.vcf", names, asm, str(tmp_path / mode), comb=COMB)
        assert (tmp_path / mode).read_bytes() == open(G + "/case1.%s.vcf" % mode, "rb").read()


@pytest.mark.gpu
def test_cli_reproduces_golden(tmp_path):
    common = ["-sequence", G + "/case1.fasta", "-readmers", G + "/case1.read.kmers.txt", "-seqmers", G + "/case1.asm.kmers.txt",
              "-peak", str(PEAK), "-prob", G + "/example_lookup_table.txt"]
    r = subprocess.run([EXE, "-hist"] + common + ["-output", str(tmp_path / "h")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert (tmp_path / "h").read_bytes() == open(G + "/case1.hist", "rb").read()
    assert open(G + "/case1.summary").read() in r.stderr
    # no -seqmers: assembly k-mers counted on the GPU -> same histogram
    r = subprocess.run([EXE, "-hist", "-sequence", G + "/case1.fasta", "-readmers", G + "/case1.read.kmers.txt", "-peak", str(PEAK),
                        "-prob", G + "/example_lookup_table.txt", "-output", str(tmp_path / "h2")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert (tmp_path / "h2").read_bytes() == open(G + "/case1.hist", "rb").read()
    # -index: first run writes the image, second run loads it and skips the k-mer databases
    img = str(tmp_path / "case1.mfxi")
    for attempt in (0, 1):
        r = subprocess.run([EXE, "-hist"] + common + ["-index", img, "-output", str(tmp_path / ("hi%d" % attempt))], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        assert (tmp_path / ("hi%d" % attempt)).read_bytes() == open(G + "/case1.hist", "rb").read()
        assert ("Writing the index image" in r.stderr) == (attempt == 0) and ("Loading the index image" in r.stderr) == (attempt == 1)
    r = subprocess.run([EXE, "-dump"] + common + ["-output", str(tmp_path / "d")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert (tmp_path / "d").read_bytes() == open(G + "/case1.dump", "rb").read()
    for mode, suffix in (("polish", ".polish.vcf"), ("filter", ".filter.vcf"), ("loose", ".filter.vcf")):
        out = str(tmp_path / mode)
        r = subprocess.run([EXE, "-" + mode] + common + ["-vcf", G + "/case1.vcf", "-comb", str(COMB), "-output", out],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        assert open(out + suffix, "rb").read() == open(G + "/case1.%s.vcf" % mode, "rb").read()


@pytest.mark.gpu
@pytest.mark.parametrize("devices", ["0,0", "0,0,0,0,0"])
def test_cli_ordered_outputs_on_several_devices(tmp_path, devices):
    """-dump and the variant modes with -devices: every slot (here all on GPU 0) takes a contiguous run of contigs, the
    parts are concatenated in order -- byte-identical to the golden single-device outputs (BASELINE config 4 names 8 GPUs);
    with more slots than contigs some slots stay empty"""
    common = ["-sequence", G + "/case1.fasta", "-readmers", G + "/case1.read.kmers.txt", "-seqmers", G + "/case1.asm.kmers.txt",
              "-peak", str(PEAK), "-prob", G + "/example_lookup_table.txt", "-devices", devices]
    r = subprocess.run([EXE, "-dump"] + common + ["-output", str(tmp_path / "d")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert (tmp_path / "d").read_bytes() == open(G + "/case1.dump", "rb").read()
    assert not [f for f in os.listdir(tmp_path) if ".part" in f]
    r1 = subprocess.run([EXE, "-dump"] + common[:-2] + ["-output", str(tmp_path / "d1")], capture_output=True, text=True)
    per_contig = lambda s: [l for l in s.splitlines() if l.count("\t") == 3 and not l.startswith("--")]
    assert per_contig(r.stderr) == per_contig(r1.stderr) and len(per_contig(r.stderr)) > 1
    for mode, suffix in (("polish", ".polish.vcf"), ("filter", ".filter.vcf"), ("loose", ".filter.vcf")):
        out = str(tmp_path / mode)
        r = subprocess.run([EXE, "-" + mode] + common + ["-vcf", G + "/case1.vcf", "-comb", str(COMB), "-output", out],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        assert open(out + suffix, "rb").read() == open(G + "/case1.%s.vcf" % mode, "rb").read()
