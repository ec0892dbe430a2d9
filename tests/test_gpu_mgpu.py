"""merfin_amd.mgpu (torchrun launcher of the multi-GPU -hist) rehearsed with two
ranks on the single GPU of the test box: gloo collectives through host memory,
both ranks on device 0.  Replicated and sharded index; output must equal the
golden -hist fixture and the single-process CLI."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")


@pytest.mark.gpu
@pytest.mark.parametrize("sharded", [False, True])
@pytest.mark.parametrize("with_seqmers", [True, False])
def test_two_rank_launcher_matches_golden(tmp_path, sharded, with_seqmers):
    out = str(tmp_path / "out.hist")
    env = dict(os.environ, MFX_MGPU_BACKEND="gloo", MFX_MGPU_SHARE_GPU="1")
    port = 29800 + (os.getpid() + 7 * sharded + 3 * with_seqmers) % 1000
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), "-m", "merfin_amd.mgpu", "-sequence", G + "/case1.fasta", "-readmers", G + "/case1.read.kmers.txt",
           "-peak", "17.3", "-prob", G + "/example_lookup_table.txt", "-output", out, "-chunk-tiles", "2"]
    if with_seqmers:
        cmd += ["-seqmers", G + "/case1.asm.kmers.txt"]
    if sharded:
        cmd += ["-sharded"]
    r = subprocess.run(cmd, capture_output=True, text=True, cwd=ROOT, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    assert open(out, "rb").read() == open(G + "/case1.hist", "rb").read()
    assert open(G + "/case1.summary").read() in r.stderr
    assert "ctg0\t" in r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("sharded", [False, True])
def test_two_rank_launcher_completeness(tmp_path, sharded):
    """-completeness through the launcher prints what the single-process CLI prints (merfin-completeness.C:117-123)"""
    env = dict(os.environ, MFX_MGPU_BACKEND="gloo", MFX_MGPU_SHARE_GPU="1")
    port = 30800 + (os.getpid() + 11 * sharded) % 1000
    common = ["-completeness", "-readmers", G + "/case1.read.kmers.txt", "-seqmers", G + "/case1.asm.kmers.txt", "-peak", "17.3"]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), "-m", "merfin_amd.mgpu"] + common + (["-sharded"] if sharded else [])
    r = subprocess.run(cmd, capture_output=True, text=True, cwd=ROOT, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    cli = subprocess.run([os.path.join(ROOT, "merfin_amd", "bin", "merfin")] + common, capture_output=True, text=True, timeout=600)
    assert cli.returncode == 0, cli.stderr[-3000:]
    pick = lambda txt: [l for l in txt.splitlines() if l.startswith(("thread ", "TOTAL ", "COMPLETENESS:"))]
    want = pick(cli.stderr)
    assert len(want) == 64 + 3 and pick(r.stderr) == want


def _launch(nproc, args, port_salt):
    env = dict(os.environ, MFX_MGPU_BACKEND="gloo", MFX_MGPU_SHARE_GPU="1")
    port = 31800 + (os.getpid() + port_salt) % 1000
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
           "--master-port", str(port), "-m", "merfin_amd.mgpu"] + args
    r = subprocess.run(cmd, capture_output=True, text=True, cwd=ROOT, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    return r


COMMON = ["-sequence", G + "/case1.fasta", "-readmers", G + "/case1.read.kmers.txt", "-seqmers", G + "/case1.asm.kmers.txt",
          "-peak", "17.3", "-prob", G + "/example_lookup_table.txt"]


@pytest.mark.gpu
@pytest.mark.parametrize("nproc", [2, 3])
def test_launcher_dump_parts_concatenate_to_golden(tmp_path, nproc):
    """-dump: contiguous run of contigs per rank, rank-ordered concatenation == the in-order single-process dump"""
    out = str(tmp_path / "out.dump")
    _launch(nproc, ["-dump"] + COMMON + ["-output", out], 17 * nproc)
    assert open(out, "rb").read() == open(G + "/case1.dump", "rb").read()
    assert [f for f in os.listdir(tmp_path) if "part" in f] == []


@pytest.mark.gpu
@pytest.mark.parametrize("nproc,with_seqmers", [(2, True), (2, False), (3, False), (5, True)])
def test_launcher_parts_hist_matches_golden(tmp_path, nproc, with_seqmers):
    """-parts: every rank indexes (sequence-only) and evaluates its own run of contigs, the assembly counts come from the
    whole assembly (merfin-globals.C:182-186), the images meet in one all-reduce; 5 ranks over 4 contigs leaves one empty"""
    out = str(tmp_path / "out.hist")
    args = ["-parts", "-sequence", G + "/case1.fasta", "-readmers", G + "/case1.read.kmers.txt", "-peak", "17.3",
            "-prob", G + "/example_lookup_table.txt", "-output", out]
    if with_seqmers:
        args += ["-seqmers", G + "/case1.asm.kmers.txt"]
    r = _launch(nproc, args, 29 * nproc + with_seqmers)
    assert open(out, "rb").read() == open(G + "/case1.hist", "rb").read()
    assert open(G + "/case1.summary").read() in r.stderr
    assert "-- Part 1 of %d" % nproc in r.stderr
    want = subprocess.run([os.path.join(ROOT, "merfin_amd", "bin", "merfin"), "-hist"] + args[1:], capture_output=True, text=True, timeout=600)
    assert want.returncode == 0, want.stderr[-2000:]
    per = lambda txt: [l for l in txt.splitlines() if l.startswith("ctg")]
    assert per(want.stderr) and per(r.stderr) == per(want.stderr)


@pytest.mark.gpu
@pytest.mark.parametrize("nproc", [2, 3])
def test_launcher_parts_dump_matches_golden(tmp_path, nproc):
    out = str(tmp_path / "out.dump")
    _launch(nproc, ["-dump", "-parts"] + COMMON + ["-output", out], 31 * nproc)
    assert open(out, "rb").read() == open(G + "/case1.dump", "rb").read()


@pytest.mark.gpu
@pytest.mark.parametrize("mode,nproc", [("polish", 2), ("filter", 2), ("loose", 2), ("polish", 3)])
def test_launcher_variant_modes_match_golden(tmp_path, mode, nproc):
    """-polish/-filter/-loose over contigs split across ranks: one header, records in contig order, byte-identical"""
    out = str(tmp_path / "out")
    _launch(nproc, ["-" + mode, "-vcf", G + "/case1.vcf", "-comb", "8"] + COMMON + ["-output", out], 5 * nproc + len(mode))
    suffix = ".polish.vcf" if mode == "polish" else ".filter.vcf"
    assert open(out + suffix, "rb").read() == open(G + "/case1.%s.vcf" % mode, "rb").read()


@pytest.mark.gpu
@pytest.mark.parametrize("nproc", [2, 3])
def test_launcher_broadcast_index_matches_golden(tmp_path, nproc):
    """-broadcast-index: only rank 0 reads the k-mer databases; the others receive the built table"""
    out = str(tmp_path / "out.hist")
    r = _launch(nproc, COMMON + ["-output", out, "-broadcast-index"], 23 * nproc)
    assert open(out, "rb").read() == open(G + "/case1.hist", "rb").read()
    assert open(G + "/case1.summary").read() in r.stderr and "Broadcasting the built table" in r.stderr


@pytest.mark.gpu
def test_two_rank_launcher_keeps_bins_beyond_the_dense_image(tmp_path):
    """asmK/readK > 13107 lands in K* bins >= 65536, beyond the all-reduced dense image: the records must be gathered
    from every rank (the reference's histogram arrays are unbounded, merfin-histogram.C:74,87)"""
    import merfin_amd as m
    from oracle import pyoracle as po
    from tests.test_cli import _write_fasta
    from tests.test_gpu_parity import oracle_hist
    from tests.test_gpu_streamed_multi import _overflow_world
    k, peak, contigs, read, asm = _overflow_world(m)
    p, g, ka, km = oracle_hist(k, peak, contigs, read, asm)
    assert g.c.undrMax > 65536 and g.c.overMax > 65536
    po.report_histogram(p, g, str(tmp_path / "o.hist"), str(tmp_path / "o.sum"))
    _write_fasta(str(tmp_path / "a.fasta"), contigs)
    m.db_write_flat(str(tmp_path / "r.mfxk"), k, *read)
    m.db_write_flat(str(tmp_path / "a.mfxk"), k, *asm)
    env = dict(os.environ, MFX_MGPU_BACKEND="gloo", MFX_MGPU_SHARE_GPU="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(29950 + os.getpid() % 40), "-m", "merfin_amd.mgpu", "-sequence", str(tmp_path / "a.fasta"),
           "-readmers", str(tmp_path / "r.mfxk"), "-seqmers", str(tmp_path / "a.mfxk"), "-peak", str(peak), "-output", str(tmp_path / "g.hist")]
    r = subprocess.run(cmd, capture_output=True, text=True, cwd=ROOT, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    assert (tmp_path / "g.hist").read_bytes() == (tmp_path / "o.hist").read_bytes()
    assert (tmp_path / "o.sum").read_text() in r.stderr
