#!/usr/bin/env python3
"""Timing of the -polish path on a config-4-shaped case: an assembly with SNP
errors, a VCF proposing their corrections (+ decoys), read k-mers from the
truth genome.  Run on the GPU box:  python tools/cfg4_polish_timing.py [bases] [variants]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import merfin_amd as m
from tools import synth_torch as st

bases = int(float(sys.argv[1])) if len(sys.argv) > 1 else 64_000_000
nvar = int(float(sys.argv[2])) if len(sys.argv) > 2 else 100_000
k, lam = 21, 26.0
out = os.environ.get("MFX_TMP", "/tmp/mfx_cfg4")
os.makedirs(out, exist_ok=True)
r = np.random.default_rng(7)
ncontig = 8
sizes = st.contig_sizes(bases, ncontig)
t0 = time.time()
truth = [st.random_bases(n, st.SEED + 17 * (i + 1), "cuda") for i, n in enumerate(sizes)]
names = ["ctg%d" % i for i in range(ncontig)]
lines = ["##fileformat=VCFv4.2"] + ["##contig=<ID=%s,length=%d>" % (n, s) for n, s in zip(names, sizes)]
lines.append("#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tSAMPLE")
asm = []
ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
for ci, t in enumerate(truth):
    a = t.cpu().numpy().copy()
    n = len(a)
    nv = max(1, nvar * n // bases)
    # bursts: half of the variants come in tight groups (clusters of 2-6 within 2k)
    pos = np.unique(np.concatenate([r.integers(30, n - 30, size=nv // 2),
                                    (r.integers(30, n - 200, size=nv // 8)[:, None] + r.integers(0, 40, size=(nv // 8, 4))).ravel()]))
    tru = a[pos].copy()
    code = np.searchsorted(ACGT, tru)
    wrong = ACGT[(code + r.integers(1, 4, size=len(pos))) % 4]
    is_err = r.random(len(pos)) < 0.8                 # 80 % real assembly errors, 20 % decoys (REF is right, ALT is wrong)
    a[pos[is_err]] = wrong[is_err]
    for p, te, w, e in zip(pos.tolist(), tru.tolist(), wrong.tolist(), is_err.tolist()):
        ref, alt = (chr(w), chr(te)) if e else (chr(te), chr(w))
        lines.append("%s\t%d\t.\t%s\t%s\t30\tPASS\t.\tGT\t1/1" % (names[ci], p + 1, ref, alt))
    asm.append(a.tobytes())
vcf = out + "/in.vcf"
open(vcf, "w").write("\n".join(lines) + "\n")
print("inputs: %d bp, %d variants, generated in %.1fs" % (bases, len(lines) - ncontig - 2, time.time() - t0), flush=True)
t0 = time.time()
ix = m.Index(k, int(bases * 2.2))
st.add_reads_from_truth(ix, truth, k, lam)
seqs = m.Sequences(asm)
ix.count_asm(seqs)
print("index built in %.1fs (%d k-mers)" % (time.time() - t0, ix.info()["distinct"]), flush=True)
ev = m.Evaluator(ix, m.KParams(lam))
for mode in ("polish", "filter"):
    t0 = time.time()
    ncl = ev.variants(mode, vcf, names, asm, out + "/out.%s.vcf" % mode, log_path=out + "/log.txt")
    dt = time.time() - t0
    nrec = sum(1 for l in open(out + "/out.%s.vcf" % mode) if not l.startswith("#"))
    print("-%s: %d clusters in %.2fs = %.0f clusters/s, %d records selected" % (mode, ncl, dt, ncl / dt, nrec), flush=True)

if len(sys.argv) > 3 and sys.argv[3] == "cli":
    # the C++ CLI on one device and with 8 slots (all on this GPU: the host side is what is being looked at)
    import subprocess
    exe = os.path.join(ROOT, "merfin_amd", "bin", "merfin")
    ek, er, ea = ix.export(sort=False)
    del ev, ix
    torch.cuda.empty_cache()
    if os.environ.get("MFX_CFG4_UNSORTED"):
        # (round 4's inputs: unsorted flat files of 8-byte packed records -- 2 x 24 GB at 3 Gb, a link-bound load of 1.2-1.4 s)
        m.db_write_flat(out + "/read.mfxk", k, ek[er > 0], er[er > 0])
        m.db_write_flat(out + "/asm.mfxk", k, ek[ea > 0], ea[ea > 0])
    else:
        # the databases as `merfin -convert` makes them of a meryl database: sorted, delta-coded blocks (2.4-3 bytes per k-mer), decoded by
        # the kernel that inserts them -- the full table of the variant modes is fed from the same form as the -hist index
        from tools import e2e_inputs
        for name, vals in (("read", er), ("asm", ea)):
            sk, sv = e2e_inputs.sorted_nonzero(torch, ek, vals, k)
            m.db_write_flat(out + "/%s.mfxk" % name, k, sk, sv)
            print("%s database: %d k-mers, %.2f GB (%.2f bytes per k-mer)" % (name, len(sk), os.path.getsize(out + "/%s.mfxk" % name) / 1e9,
                                                                            os.path.getsize(out + "/%s.mfxk" % name) / max(len(sk), 1)), flush=True)
            del sk, sv
    del ek, er, ea
    with open(out + "/asm.fasta", "wb") as f:
        for nm, a in zip(names, asm):
            f.write(b">" + nm.encode() + b"\n" + a + b"\n")
    torch.cuda.empty_cache()
    if os.environ.get("MFX_CFG4_PATH_AB"):
        # the PATH-ONLY index (the default of one slot: the call set prepared first, its paths' k-mers claimed, both databases staged and
        # update-only) against the full tables (MFX_CLI_PATH_INDEX=0), alternating; the records must be the same bytes
        ref_out = None
        for rep, tok in enumerate(os.environ["MFX_CFG4_PATH_AB"].split(",")):
            pi, _, extra = tok.partition(":")                     # "0:MFX_CLI_FULL_STAGE=0": the full tables with the databases read when the build gets there
            xenv = dict(kv.split("=", 1) for kv in extra.split(":") if kv)
            time.sleep(float(os.environ.get("MFX_CFG4_SLEEP", "0")))
            t0 = time.time()
            r = subprocess.run([exe, "-polish", "-sequence", out + "/asm.fasta", "-readmers", out + "/read.mfxk", "-seqmers", out + "/asm.mfxk", "-peak", str(lam),
                                "-vcf", vcf, "-output", out + "/cli_p" + pi], capture_output=True, text=True,
                               env=dict(os.environ, MFX_CLI_TIMING="2", MFX_VAR_TIMING="1", MFX_INGEST_TIMING="1", MFX_CLI_PATH_INDEX=pi, **xenv))
            dt = time.time() - t0
            data = open(out + "/cli_p" + pi + ".polish.vcf").read() if r.returncode == 0 else ""
            if ref_out is None:
                ref_out = data
            print("merfin -polish MFX_CLI_PATH_INDEX=%s %s(%s): rc=%d wall=%.2fs same=%s %d bytes" % (pi, extra + " " if extra else "", "path-only index" if pi == "1" else "full tables", r.returncode, dt,
                                                                                                  data == ref_out, len(data)), flush=True)
            print("    " + "\n    ".join(l for l in r.stderr.splitlines() if "timing" in l or "ERROR" in l or "Memory needed" in l or "staged build" in l or "stager " in l and "sequence was in" in l
                                          or "mfx_variants]" in l and "load:" not in l))
        api = open(out + "/out.polish.vcf").read()
        print("API -polish on the full index == the CLI's records: %s" % (api == ref_out))
        sys.exit(0)
    for devs in ("0", "0,0,0,0,0,0,0,0"):
        time.sleep(float(os.environ.get("MFX_CFG4_SLEEP", "0")))
        t0 = time.time()
        r = subprocess.run([exe, "-polish", "-sequence", out + "/asm.fasta", "-readmers", out + "/read.mfxk", "-seqmers", out + "/asm.mfxk", "-peak", str(lam),
                            "-vcf", vcf, "-output", out + "/cli_" + str(len(devs)), "-devices", devs], capture_output=True, text=True,
                           env=dict(os.environ, MFX_CLI_TIMING="1"))
        dt = time.time() - t0
        print("merfin -polish -devices %s: rc=%d wall=%.2fs" % (devs, r.returncode, dt))
        print("    " + "\n    ".join(l for l in r.stderr.splitlines() if "timing" in l or "ERROR" in l))
    a_, b_ = open(out + "/cli_1.polish.vcf").read(), open(out + "/cli_15.polish.vcf").read()
    print("8 slots == 1 device:", a_ == b_, len(a_))
    try:
        print("host: %d cpus in the affinity mask, cpu.max %s" % (len(os.sched_getaffinity(0)), open("/sys/fs/cgroup/cpu.max").read().strip()))
    except Exception:
        pass
    if a_ != b_:
        A, B = a_.splitlines(), b_.splitlines()
        nd = 0
        for i, (x, y) in enumerate(zip(A, B)):
            if x != y:
                nd += 1
                if nd <= 12:
                    print("line %d:\n  1 slot : %s\n  8 slots: %s" % (i, x[:200], y[:200]))
        print("differing lines: %d of %d / %d; same multiset: %s" % (nd, len(A), len(B), sorted(A) == sorted(B)))
        import shutil
        os.makedirs(os.path.join(ROOT, "gpurun_out", "cfg4_diff"), exist_ok=True)
        open(os.path.join(ROOT, "gpurun_out", "cfg4_diff", "api.polish.head"), "w").write("".join(open(out + "/out.polish.vcf").readlines()[:20]))
        api = open(out + "/out.polish.vcf").read()
        print("API -polish == 1 slot: %s, == 8 slots: %s" % (api == a_, api == b_))
    # one device run as 1 / 2 / 4 (the default) slots sharing its table: which fills the host best
    for slots in (os.environ.get("MFX_CFG4_SLOTS", "1,2,4,1,4")).split(","):
        time.sleep(float(os.environ.get("MFX_CFG4_SLEEP", "0")))        # (the driver scrubs what the run before freed: a table allocated at once waits for it)
        t0 = time.time()
        r = subprocess.run([exe, "-polish", "-sequence", out + "/asm.fasta", "-readmers", out + "/read.mfxk", "-seqmers", out + "/asm.mfxk", "-peak", str(lam),
                            "-vcf", vcf, "-output", out + "/cli_s" + slots], capture_output=True, text=True,
                           env=dict(os.environ, MFX_CLI_TIMING="2", MFX_VARIANT_SLOTS=slots, MFX_VAR_TIMING="1", MFX_INGEST_TIMING="1"))
        dt = time.time() - t0
        print("merfin -polish MFX_VARIANT_SLOTS=%s: rc=%d wall=%.2fs same=%s" % (slots, r.returncode, dt, open(out + "/cli_s" + slots + ".polish.vcf").read() == a_))
        print("    " + "\n    ".join(l for l in r.stderr.splitlines() if "timing" in l or "ERROR" in l or "ingest" in l or "mfx_variants]" in l and "load:" not in l))

    if os.environ.get("MFX_CFG4_AHEAD_AB"):
        # stage A of the run prepared under the index build (MFX_CLI_VCF_AHEAD=2) against the VCF load alone (the default)
        for ah in ("1", "2", "1", "2", "2"):
            time.sleep(float(os.environ.get("MFX_CFG4_SLEEP", "0")))
            t0 = time.time()
            r = subprocess.run([exe, "-polish", "-sequence", out + "/asm.fasta", "-readmers", out + "/read.mfxk", "-seqmers", out + "/asm.mfxk", "-peak", str(lam),
                                "-vcf", vcf, "-output", out + "/cli_ab" + ah], capture_output=True, text=True,
                               env=dict(os.environ, MFX_CLI_TIMING="2", MFX_VARIANT_SLOTS="1", MFX_VAR_TIMING="1", MFX_CLI_VCF_AHEAD=ah))
            dt = time.time() - t0
            print("merfin -polish MFX_CLI_VCF_AHEAD=%s (%s): rc=%d wall=%.2fs same=%s" % (ah, "load + stage A ahead" if ah == "2" else "load ahead", r.returncode, dt,
                                                                                       open(out + "/cli_ab" + ah + ".polish.vcf").read() == a_))
            print("    " + "\n    ".join(l for l in r.stderr.splitlines() if "-- timing" in l or "ERROR" in l or "mfx_variants]" in l and "load:" not in l))

    if os.environ.get("MFX_CFG4_DIFF"):
        # the VCF loaded ahead (default) against MFX_CLI_VCF_AHEAD=0: the outputs, and where they differ
        outs = {}
        for ah in ("1", "0", "1"):
            time.sleep(float(os.environ.get("MFX_CFG4_SLEEP", "0")))
            r = subprocess.run([exe, "-polish", "-sequence", out + "/asm.fasta", "-readmers", out + "/read.mfxk", "-seqmers", out + "/asm.mfxk", "-peak", str(lam),
                                "-vcf", vcf, "-output", out + "/cli_ah" + ah], capture_output=True, text=True, env=dict(os.environ, MFX_CLI_VCF_AHEAD=ah))
            t = open(out + "/cli_ah" + ah + ".polish.vcf").read().splitlines()
            print("MFX_CLI_VCF_AHEAD=%s rc=%d: %d lines; == 8 slots: %s; == first ahead run: %s" % (ah, r.returncode, len(t), "\n".join(t) + "\n" == b_, outs.get("1") == t if "1" in outs else None))
            outs[ah] = t
        A, B = outs["1"], outs["0"]
        nd = 0
        for i, (x, y) in enumerate(zip(A, B)):
            if x != y:
                nd += 1
                if nd <= 8:
                    print("line %d:\n  ahead: %s\n  plain: %s" % (i, x[:160], y[:160]))
        print("differing lines: %d of %d / %d; same multiset: %s" % (nd, len(A), len(B), sorted(A) == sorted(B)))

    if os.environ.get("MFX_CFG4_SOAK"):
        # every report type in 1 / 4 / 8 slots on this device, several times: the outputs must be the single slot's, byte for byte
        # (slots on one device share its null stream: what a fill, a copy or a kernel of one slot may do to another shows here)
        reps = int(os.environ["MFX_CFG4_SOAK"])
        common = ["-sequence", out + "/asm.fasta", "-readmers", out + "/read.mfxk", "-seqmers", out + "/asm.mfxk", "-peak", str(lam)]
        ref = {}
        ops = os.environ.get("MFX_CFG4_SOAK_OPS", "-hist,-dump,-polish,-filter").split(",")
        for op, extra, suffix in (("-hist", [], ""), ("-dump", [], ""), ("-polish", ["-vcf", vcf], ".polish.vcf"), ("-filter", ["-vcf", vcf], ".filter.vcf"),
                                  ("-hist", ["-sharded"], ""), ("-dump", ["-sharded"], ""), ("-polish", ["-sharded", "-vcf", vcf], ".polish.vcf")):
            if op not in ops or ("-sharded" in extra and not os.environ.get("MFX_CFG4_SOAK_SHARDED")):
                continue
            for devs in ("0", "0,0,0,0", "0,0,0,0,0,0,0,0"):
                if devs == "0" and "-sharded" in extra:
                    continue                                        # (the reference is the unsharded single slot's, above)
                for rep in range(1 if devs == "0" else reps):
                    o = out + "/soak" + op
                    t0 = time.time()
                    r = subprocess.run([exe, op] + common + extra + ["-output", o, "-devices", devs], capture_output=True, text=True)
                    data = open(o + suffix, "rb").read() if r.returncode == 0 else b""
                    if devs == "0":
                        ref[op] = data
                    print("soak %s %s-devices %-15s rep %d: rc=%d %.2fs %d bytes same=%s" % (op, "-sharded " if "-sharded" in extra else "", devs, rep, r.returncode, time.time() - t0, len(data), data == ref[op]), flush=True)
                    if r.returncode or data != ref[op]:
                        print("    " + "\n    ".join(r.stderr.splitlines()[-25:]))
                        print("    files:", sorted(f for f in os.listdir(out) if f.startswith("soak")))
