"""`python bench.py --gpus N` without a launcher around it starts its own N ranks (the reference binary owns its workers,
merfin.C:366-414) and never reports a GPU count it did not use.  On the 1-GPU box the ranks rehearse on device 0
(MFX_BENCH_REHEARSE=1: RCCL refuses two ranks per device, the reduction goes through host memory and the line says so);
tests/test_gpu_multidevice.py runs the same command over RCCL where >= 2 devices exist."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(args, env=None, timeout=900):
    e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    e.update(env or {})
    return subprocess.run([sys.executable, "bench.py"] + args, cwd=ROOT, capture_output=True, text=True, env=e, timeout=timeout)


def test_plain_command_starts_its_own_ranks():
    common = ["--steps", "2", "--warmup", "1", "--bases", "48e6", "--no-pmc", "--no-cpu-baseline", "--no-e2e", "--no-k31", "--no-full-index"]
    r = _bench(["--gpus", "2"] + common, env={"MFX_BENCH_REHEARSE": "1"})
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["config"]["hist_sum_check"] is True
    assert len(d["config"]["rank_kernel_ms"]["all"]) == 2
    assert "bench.py started its own ranks" in d["config"]["launcher"]
    assert d["config"]["collective"].startswith("host memory")          # the rehearsal never claims RCCL
    # SURVEY 8(d)'s metric at N > 1: every rank streamed its share from host memory, same histogram
    assert d["value_8d"] and d["value_8d"] > 0 and d["config"]["h2d_inclusive_multi"]["equals_resident_result"] is True
    r1 = _bench(["--gpus", "1", "--no-streamed"] + common)
    assert r1.returncode == 0, r1.stderr[-3000:]
    d1 = json.loads([l for l in r1.stdout.splitlines() if l.startswith("{")][0])
    assert d1["n_gpus"] == 1 and d1["config"]["collective"] is None
    assert d1["config"]["kmissing"] == d["config"]["kmissing"] and d1["config"]["valid_kmers"] == d["config"]["valid_kmers"]
    assert abs(d1["config"]["koverCpy"] - d["config"]["koverCpy"]) <= 1e-9 * max(1.0, abs(d1["config"]["koverCpy"]))


def test_more_gpus_than_devices_is_refused():
    import merfin_amd as m
    n = m.device_count() + 1
    r = _bench(["--gpus", str(n), "--steps", "1", "--warmup", "0"])
    assert r.returncode != 0
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert d["value"] is None and d["n_gpus"] == n and "visible" in d["error"]
