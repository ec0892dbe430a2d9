"""Where the wall time of a whole `merfin -hist` process goes that the CLI's phase timers do not see: process start -> main,
the HIP runtime coming up (the device check), and tear-down after the last phase.  Inputs as bench.py's e2e leg (tools/e2e_inputs.py).
  python tools/cli_startup_timing.py [bases]        (on the GPU box)"""
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import merfin_amd as m
    from tools import synth_torch as st, e2e_inputs
    bases = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1000000000
    tmp = tempfile.mkdtemp(prefix="mfx_cst_", dir="/dev/shm" if os.path.isdir("/dev/shm") else "/tmp")
    try:
        inp = e2e_inputs.write_inputs(m, st, torch, bases, tmp, ncontigs=24, k=21, lam=26.0, log=lambda s: print("  " + s, flush=True))
        torch.cuda.empty_cache()
        exe = os.path.join(ROOT, "merfin_amd", "bin", "merfin")
        prob = os.path.join(ROOT, "tests", "golden", "example_lookup_table.txt")
        cmd = [exe, "-hist", "-sequence", inp["fasta"], "-readmers", inp["readdb"], "-peak", "26", "-prob", prob, "-output", os.path.join(tmp, "o.hist")]
        variants = ({}, {}, {}, {"MFX_CLI_WARM": "0"}, {"MFX_CLI_WARM": "0"}, {}, {"MFX_CLI_WARM": "0"})
        if os.environ.get("MFX_CST_VARIANTS") == "upload":
            variants = ({}, {"MFX_POOL_SPREAD": "0"}, {}, {"MFX_POOL_SPREAD": "0"})
        if os.environ.get("MFX_CST_VARIANTS") == "overlap":
            variants = ({}, {"MFX_BUILD_OVERLAP": "0"}, {}, {"MFX_BUILD_OVERLAP": "0"}, {"MFX_INGEST_RING_MB": "0"}, {}, {"MFX_BUILD_OVERLAP": "0"}, {"MFX_INGEST_RING_MB": "0"})
        for variant in variants:
            time.sleep(3)
            t0 = time.time()
            r = subprocess.run(cmd, stdin=subprocess.DEVNULL, capture_output=True, text=True, env=dict(os.environ, MFX_CLI_TIMING="3", MFX_UPLOAD_TIMING="1", MFX_CLI_SEQ_TIMING="1", MFX_INGEST_TIMING="1", **variant))
            t1 = time.time()
            st_ = [l for l in r.stderr.splitlines() if l.startswith("-- stamps:")]
            ph = [l for l in r.stderr.splitlines() if l.startswith("-- timing:")]
            if r.returncode or not st_:
                print("rc", r.returncode, r.stderr[-500:])
                continue
            w = st_[0].split()
            t_main, t_dev, t_end = float(w[3]), float(w[5]), float(w[7])
            print("%-24s wall %.3f s = spawn -> main %.3f + device check (HIP up) %.3f + phases %.3f + end -> exit %.3f" %
                  (variant or "(default)", t1 - t0, t_main - t0, t_dev - t_main, t_end - t_dev, t1 - t_end), flush=True)
            print("     " + (ph[0] if ph else ""), flush=True)
            for l in r.stderr.splitlines():
                if l.startswith(("-- upload", "-- packed upload", "-- read_fasta_parallel", "-- timing (index)", "-- ingest")):
                    print("     " + l, flush=True)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
