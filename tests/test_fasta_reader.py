"""The CLI's two FASTA readers (merfin_amd/cli/fasta.h; they replace dnaSeqFile::loadSequence, merfin.C:30-53) must return
the same records for every layout of a file: the sequential SeqFile::next (any input, compressed too) and
read_fasta_parallel (plain FASTA, all host threads, slices and pieces at arbitrary byte offsets).  Host code only: a small
program built from tests/native/fasta_readers_check.cpp runs both on seeded files."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def checker(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("fasta") / "fasta_readers_check")
    subprocess.run(["g++", "-O2", "-std=c++17", "-pthread", os.path.join(ROOT, "tests", "native", "fasta_readers_check.cpp"), "-o", exe], check=True)
    return exe


def layout(r, nrec):
    eol = b"\r\n" if r.random() < 0.4 else b"\n"
    body, want = b"", []
    for i in range(nrec):
        n = int(r.choice([0, 1, 20, 61, 400, 5000, 70000]))
        c = bytes(r.choice(np.frombuffer(b"ACGTacgtN", dtype=np.uint8), size=n).tobytes())
        name = b"ctg%d" % i
        desc = [b"", b" len=%d >not a header" % n, b"\tdescription with\ttabs"][int(r.integers(0, 3))]
        body += b">" + name + desc + eol
        width = int(r.choice([1, 7, 60, 61, 4096, 10**9]))
        for o in range(0, n, width):
            body += c[o:o + width] + eol
            if r.random() < 0.05:
                body += eol                                           # a blank line inside a record
        if r.random() < 0.3:
            body += eol
        want.append((name.decode(), c))
    end = r.random()
    if end < 0.3 and body.endswith(eol):
        body = body[:-len(eol)]                                       # no newline at the end of the file
    elif end < 0.4 and body.endswith(b"\n"):
        body = body[:-1] + (b"" if eol == b"\n" else b"")             # CR LF files: the file ends in a bare CR
    return body, want


@pytest.mark.parametrize("seed", range(24))
def test_parallel_reader_equals_sequential_reader(tmp_path, checker, seed):
    r = np.random.default_rng(9000 + seed)
    body, want = layout(r, int(r.integers(1, 9)))
    fa = str(tmp_path / "a.fasta")
    open(fa, "wb").write(body)
    nbases = sum(len(c) for _, c in want)
    for slice_bytes in ("0", "1", "17", "4099", "65536"):
        env = dict(os.environ, MFX_CLI_SEQ_PAR_MIN="0", MFX_CLI_SEQ_THREADS="5")
        if slice_bytes != "0":
            env["MFX_CLI_SEQ_SLICE"] = slice_bytes
        out = subprocess.run([checker, fa], capture_output=True, text=True, env=env)
        assert out.returncode == 0, (slice_bytes, out.stdout)
        got = out.stdout.split()
        assert got[:2] == ["same", str(len(want))], (slice_bytes, out.stdout)
        if not body.endswith(b"\r"):                                  # (a CR that ends the file has no LF: both readers keep it)
            assert got[2] == str(nbases), (slice_bytes, out.stdout)
    # and the sequential reader against the layout itself
    if not body.endswith(b"\r"):
        d = subprocess.run([checker, fa, "dump"], capture_output=True, text=True).stdout.splitlines()
        assert [l.split("\t")[0] for l in d] == [n for n, _ in want]
        assert [int(l.split("\t")[1]) for l in d] == [len(c) for _, c in want]


def test_inputs_the_parallel_reader_leaves_to_the_sequential_one(tmp_path, checker):
    env = dict(os.environ, MFX_CLI_SEQ_PAR_MIN="0")
    for name, body in (("q.fastq", b"@r1\nACGT\n+\nIIII\n"), ("junk.fasta", b"\n>ctg\nACGT\n"), ("e.fasta", b"")):
        p = str(tmp_path / name)
        open(p, "wb").write(body)
        assert subprocess.run([checker, p], capture_output=True, text=True, env=env).stdout.strip() == "not applicable"
    p = str(tmp_path / "one.fasta")
    open(p, "wb").write(b">ctg\nACGT\n")
    assert subprocess.run([checker, p], capture_output=True, text=True, env=dict(env, MFX_CLI_SEQ_THREADS="1")).stdout.strip() == "not applicable"
    assert subprocess.run([checker, p], capture_output=True, text=True, env=env).stdout.split() == ["same", "1", "4"]
