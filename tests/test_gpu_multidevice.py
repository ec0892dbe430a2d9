"""Every multi-device path over DISTINCT physical GPUs.  The development box has one GPU, so these tests skip there
(their device-0,0 twins run instead: tests/test_gpu_streamed_multi.py, test_gpu_sharded.py, test_golden.py, test_cli.py);
on a node with two or more GPUs they are the first real run of
  - the replicas made by the doubling tree over xGMI (mfx_index_replicate_many, mfx_seq_replicate_many) and the
    one-process N-device -hist (mfx_hist_run_multi),
  - the sharded index across devices: routed -hist with peer copies (mfx_hist_run_sharded), -dump and the variant modes
    with the value arrays added on slot 0 (mfx_dump_values_sharded, mfx_variants_run_sharded),
  - the one-process-per-GPU collective on RCCL with more than one rank (bench.py under torchrun; the sharded launcher with the
    library's all-to-all-v),
  - the CLI's -devices 0-(N-1) for every report type against the committed golden outputs."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import pyoracle as po
from tests import synth
from tests.test_gpu_parity import assert_hist_equal, oracle_hist, build_index

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")
EXE = os.path.join(ROOT, "merfin_amd", "bin", "merfin")
PEAK, COMB = 17.3, 8


def _devices():
    import merfin_amd as m
    n = m.device_count()
    if n < 2:
        pytest.skip("needs >= 2 GPUs (this box shows %d); the device-0,0 twins of these tests cover the code on one GPU" % n)
    return list(range(min(n, 8)))


@pytest.mark.parametrize("kind", ["full", "seq_only"])
def test_replicas_on_distinct_devices_and_hist_multi(kind):
    import merfin_amd as m
    devs = _devices()
    k, peak = 21, 17.3
    contigs, read, asm = synth.world(k=k, peak=peak, seed=501, sizes=(300000, 90000, 4097, 30, 0, 250000))
    p, g, ka, km = oracle_hist(k, peak, contigs, read, asm)
    seqs0 = m.Sequences(contigs, device=devs[0])
    if kind == "full":
        ix0 = build_index(m, k, read, asm)
    else:
        ix0 = m.Index.for_seq(k, sum(len(c) for c in contigs) + 16, device=devs[0])
        ix0.count_asm(seqs0)
        ix0.add_read(*read)
    ixs = [ix0] + ix0.replicate_many(devs[1:])
    sqs = [seqs0] + seqs0.replicate_many(devs[1:])
    for ix in ixs[1:]:
        assert ix.info()["distinct"] == ix0.info()["distinct"] and ix.info()["seq_only"] == (kind == "seq_only")
        rv, av = ix.value(asm[0][::7])
        np.testing.assert_array_equal(av, asm[1][::7])
    evs = [m.Evaluator(ix, m.KParams(peak)) for ix in ixs]
    res = m.hist_multi(evs, sqs)
    assert_hist_equal(res, g, ka, km, k)
    again = m.hist_multi(evs, sqs)                                # evaluator-owned streams / images are re-armed
    assert again.koverCpy == res.koverCpy and again.kmissing == res.kmissing
    # a replica alone gives the whole answer too (and unpacks its planes when asked for raw values)
    assert_hist_equal(evs[-1].hist(sqs[-1]), g, ka, km, k)
    a = evs[-1].dump_values(sqs[-1], 0, 0, len(contigs[0]))
    b = evs[0].dump_values(sqs[0], 0, 0, len(contigs[0]))
    np.testing.assert_array_equal(a[0], b[0])
    np.testing.assert_array_equal(a[1], b[1])


@pytest.mark.parametrize("k", [21, 31])
def test_sharded_index_across_devices(k, tmp_path):
    import merfin_amd as m
    devs = _devices()
    N = len(devs)
    peak = 17.3
    contigs, read, asm = synth.world(k=k, peak=peak, seed=503, sizes=(200000, 60000, 4097, 21, 0, 150000))
    p, g, ka, km = oracle_hist(k, peak, contigs, read, asm)
    shards, sqs = [], []
    for r, d in enumerate(devs):
        ix = m.Index(k, len(read[0]) + len(asm[0]) + 16, device=d)
        ix.set_shard(r, N)
        ix.add_read(*read)
        sq = m.Sequences(contigs, device=d)
        ix.count_asm(sq)
        shards.append(ix)
        sqs.append(sq)
    evs = [m.Evaluator(s, m.KParams(peak)) for s in shards]
    routers = [m.Router(s, N, 16) for s in shards]
    assert_hist_equal(m.hist_sharded(evs, routers, sqs), g, ka, km, k)
    whole = m.Evaluator(build_index(m, k, read, asm), m.KParams(peak))
    sq0 = m.Sequences(contigs)
    for c in (0, 2, 5):
        a = m.dump_values_sharded(evs, sqs, c, 0, len(contigs[c]))
        b = whole.dump_values(sq0, c, 0, len(contigs[c]))
        np.testing.assert_array_equal(a[0], b[0])
        np.testing.assert_array_equal(a[1], b[1])
        assert a[2:] == b[2:]
    # the variant modes over the shards: same VCF bytes as over the whole index
    names, vcontigs, vcf, vread, vasm = synth.variant_world(k=k, peak=peak, seed=505)
    (tmp_path / "in.vcf").write_text(vcf)
    vsh = []
    for r, d in enumerate(devs):
        ix = m.Index(k, len(vread[0]) + len(vasm[0]) + 16, device=d)
        ix.set_shard(r, N)
        ix.add_read(*vread)
        ix.add_asm(*vasm)
        vsh.append(ix)
    vevs = [m.Evaluator(s, m.KParams(peak)) for s in vsh]
    wev = m.Evaluator(build_index(m, k, vread, vasm), m.KParams(peak))
    for mode in ("polish", "filter"):
        m.variants_sharded(vevs, mode, str(tmp_path / "in.vcf"), names, vcontigs, str(tmp_path / ("s_" + mode)), comb=COMB)
        wev.variants(mode, str(tmp_path / "in.vcf"), names, vcontigs, str(tmp_path / ("w_" + mode)), comb=COMB)
        assert (tmp_path / ("s_" + mode)).read_bytes() == (tmp_path / ("w_" + mode)).read_bytes()


def test_streamed_hist_over_distinct_devices():
    """SURVEY 8(d)'s evaluate phase at N GPUs (mfx_hist_run_streamed_multi): every device receives and evaluates its own stretch of
    the assembly; equal to the oracle and, bit for bit, to the resident single-device run"""
    import merfin_amd as m
    devs = _devices()
    k, peak = 21, 17.3
    contigs, read, asm = synth.world(k=k, peak=peak, seed=509, sizes=(4200000 * len(devs) // 2, 1100000, 4097, 500, 0, 8191), err_kmers=3000)
    p, g, ka, km = oracle_hist(k, peak, contigs, read, asm)
    ix0 = build_index(m, k, read, asm)
    ixs = [ix0] + ix0.replicate_many(devs[1:])
    evs = [m.Evaluator(ix, m.KParams(peak)) for ix in ixs]
    lens = [len(c) for c in contigs]
    sqs = [m.Sequences.create(lens, device=d) for d in devs]
    res = m.hist_streamed_multi(evs, sqs, contigs)
    assert_hist_equal(res, g, ka, km, k)
    one = evs[0].hist(m.Sequences(contigs, device=devs[0]))
    assert (res.kasm, res.kmissing, res.koverCpy) == (one.kasm, one.kmissing, one.koverCpy)
    np.testing.assert_array_equal(res.undr(), one.undr())
    np.testing.assert_array_equal(res.over(), one.over())


def _torchrun(n, args, env=None, timeout=900):
    port = 29500 + os.getpid() % 400
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port)] + args
    e = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **(env or {}))
    return subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, env=e, timeout=timeout)


def test_bench_over_rccl_with_more_than_one_rank():
    """bench.py for N > 1, under torchrun and as the plain `python bench.py --gpus N` (it then starts its own ranks): the reduction must be the library's RCCL all-reduce (not the
    host-memory fallback), the reduced histogram must account for every k-mer, and rank 0 prints ONE line"""
    devs = _devices()
    n = len(devs)
    bargs = ["bench.py", "--gpus", str(n), "--steps", "3", "--warmup", "1", "--bases", "64e6", "--no-pmc", "--no-cpu-baseline"]
    r = _torchrun(n, bargs)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == n and d["config"]["hist_sum_check"] is True
    assert "RCCL via libmerfin_amd" in d["config"]["parallelism"], d["config"]["parallelism"]
    assert d["config"]["collective"] == "RCCL" and d["config"]["rccl_ranks"] == n and len(d["config"]["rank_kernel_ms"]["all"]) == n
    # ... and the plain command, no launcher around it: bench.py starts its own N ranks (README: `python bench.py --gpus N`)
    e = {k_: v for k_, v in os.environ.items() if k_ not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    rp = subprocess.run([sys.executable] + bargs, cwd=ROOT, capture_output=True, text=True, env=e, timeout=900)
    assert rp.returncode == 0, rp.stderr[-3000:]
    dp = json.loads([l for l in rp.stdout.splitlines() if l.startswith("{")][0])
    assert dp["n_gpus"] == n and dp["config"]["collective"] == "RCCL" and dp["config"]["rccl_ranks"] == n
    assert dp["config"]["kmissing"] == d["config"]["kmissing"] and dp["config"]["valid_kmers"] == d["config"]["valid_kmers"]
    # the same workload on one GPU: identical counters
    r1 = subprocess.run([sys.executable, "bench.py", "--gpus", "1", "--steps", "1", "--warmup", "0", "--bases", "64e6", "--no-pmc", "--no-cpu-baseline",
                         "--no-streamed"], cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r1.returncode == 0, r1.stderr[-3000:]
    d1 = json.loads([l for l in r1.stdout.splitlines() if l.startswith("{")][0])
    assert d1["config"]["kmissing"] == d["config"]["kmissing"] and d1["config"]["valid_kmers"] == d["config"]["valid_kmers"]
    assert abs(d1["config"]["koverCpy"] - d["config"]["koverCpy"]) <= 1e-9 * max(1.0, abs(d1["config"]["koverCpy"]))


@pytest.mark.parametrize("sharded", [False, True])
def test_launcher_one_process_per_gpu_over_rccl(sharded, tmp_path):
    """merfin_amd.mgpu under torchrun: the counts image all-reduced -- and, with -sharded, the routed k-mers exchanged --
    by the library's own RCCL collectives; histogram byte-identical to the golden single-GPU output"""
    devs = _devices()
    out = str(tmp_path / "h")
    args = ["-m", "merfin_amd.mgpu", "-sequence", G + "/case1.fasta", "-readmers", G + "/case1.read.kmers.txt", "-seqmers",
            G + "/case1.asm.kmers.txt", "-peak", str(PEAK), "-prob", G + "/example_lookup_table.txt", "-output", out]
    if sharded:
        args += ["-sharded", "-chunk-tiles", "3"]
    r = _torchrun(len(devs), args)
    assert r.returncode == 0, r.stderr[-3000:]
    assert open(out, "rb").read() == open(G + "/case1.hist", "rb").read()
    assert open(G + "/case1.summary").read() in r.stderr


def test_cli_devices_every_report_type_vs_golden(tmp_path):
    devs = _devices()
    spec = "0-%d" % (len(devs) - 1)
    common = ["-sequence", G + "/case1.fasta", "-readmers", G + "/case1.read.kmers.txt", "-seqmers", G + "/case1.asm.kmers.txt",
              "-peak", str(PEAK), "-prob", G + "/example_lookup_table.txt", "-devices", spec]
    run = lambda a: subprocess.run([EXE] + a, capture_output=True, text=True, timeout=900)
    for extra in ([], ["-sharded"]):
        r = run(["-hist"] + common + extra + ["-output", str(tmp_path / "h")])
        assert r.returncode == 0, r.stderr
        assert (tmp_path / "h").read_bytes() == open(G + "/case1.hist", "rb").read()
        assert open(G + "/case1.summary").read() in r.stderr
        r = run(["-dump"] + common + extra + ["-output", str(tmp_path / "d")])
        assert r.returncode == 0, r.stderr
        assert (tmp_path / "d").read_bytes() == open(G + "/case1.dump", "rb").read()
        for mode, suffix in (("polish", ".polish.vcf"), ("filter", ".filter.vcf"), ("loose", ".filter.vcf")):
            out = str(tmp_path / mode)
            r = run(["-" + mode] + common + extra + ["-vcf", G + "/case1.vcf", "-comb", str(COMB), "-output", out])
            assert r.returncode == 0, r.stderr
            assert open(out + suffix, "rb").read() == open(G + "/case1.%s.vcf" % mode, "rb").read()
    # -completeness: replicated (first device) and sharded print the same totals
    tot = []
    for extra in ([], ["-sharded"]):
        r = run(["-completeness", "-readmers", G + "/case1.read.kmers.txt", "-seqmers", G + "/case1.asm.kmers.txt", "-peak", str(PEAK),
                 "-devices", spec] + extra)
        assert r.returncode == 0, r.stderr
        tot.append([l for l in r.stderr.splitlines() if l.startswith(("TOTAL", "COMPLETENESS"))])
    assert tot[0] == tot[1] and len(tot[0]) == 3


@pytest.mark.parametrize("k,from_db", [(21, False), (31, True)])
def test_parts_of_an_assembly_over_distinct_devices(k, from_db):
    """config 5's -hist on the node: every GPU holds the sequence-only index of ITS contigs' k-mers (claimed there, counted over the
    whole assembly), evaluates alone, mfx_hist_run_parts adds the images -- the twin of tests/test_gpu_parts.py over distinct devices"""
    import merfin_amd as m
    from tests.test_gpu_parts import build_parts
    devs = _devices()
    peak = 17.3
    contigs, read, asm = synth.world(k=k, peak=peak, seed=640 + k, sizes=(300000, 90000, 4097, 30, 0, 250000, 120000, 7000))
    p, g, ka, km = oracle_hist(k, peak, contigs, read, asm)
    nslots = len(devs) + 1                                         # (one device carries two slots)
    ixs, seqs, ids = build_parts(m, k, contigs, read, nslots, asm_db=asm if from_db else None, device_of=lambda d: devs[d % len(devs)])
    evs = [m.Evaluator(ix, m.KParams(peak)) for ix in ixs]
    assert_hist_equal(m.hist_parts(evs, seqs, ids, len(contigs)), g, ka, km, k)


def test_cli_parts_on_distinct_devices(tmp_path):
    """`merfin -hist|-dump -devices 0-(N-1) -sharded`: a part of the sequences per GPU, the committed golden outputs"""
    devs = _devices()
    common = ["-sequence", os.path.join(G, "case1.fasta"), "-readmers", os.path.join(G, "case1.read.kmers.txt"), "-peak", str(PEAK),
              "-prob", os.path.join(G, "example_lookup_table.txt"), "-devices", ",".join(str(d) for d in devs), "-sharded"]
    r = subprocess.run([EXE, "-hist"] + common + ["-output", str(tmp_path / "p.hist")], capture_output=True, text=True)
    assert r.returncode == 0 and "one part of the sequences each" in r.stderr, r.stderr
    assert (tmp_path / "p.hist").read_bytes() == open(os.path.join(G, "case1.hist"), "rb").read()
    r = subprocess.run([EXE, "-dump"] + common + ["-output", str(tmp_path / "p.dump")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert (tmp_path / "p.dump").read_bytes() == open(os.path.join(G, "case1.dump"), "rb").read()
